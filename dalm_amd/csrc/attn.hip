// Scaled-dot-product attention of the towers - forward and backward, bf16, head width 128 (Llama-2-7b: BASELINE.json configs 3 / 4)
// or 64 (Falcon-7b, config 5; bge-large / BERT, every config), arbitrary boolean mask (HF's causal + left-padding mask, BERT's
// padding mask), optional attention dropout - hand-written for gfx950.
// transformers reaches torch.nn.functional.scaled_dot_product_attention through sdpa_attention_forward
// (transformers/integrations/sdpa_attention.py) or FalconAttention.forward; the reference reaches it through
// self.generator_model(...) / self.retriever_model(...) (dalm/models/rag_e2e_base_model.py:84-106) and differentiates it with
// loss.backward() (dalm/training/rag_e2e/train_rage2e.py:466).  With a mask torch dispatches its memory-efficient kernels: at cfg3
// (B 18, H 32, T 256, hd 128) 119 us forward and 395 us backward per layer - 16 ms of a 140 ms step, 2.7 % of the MFMA peak, 629 MB
// of HBM reads for a 113 MB problem (profiles/r05_attn_pmc.txt).  Here: 47 us + 152 us, traffic 1.0 x algorithmic.
//
//   P = exp(scale S + mask - lse),  S = Q K^T,  O = P V      dV = P^T dO          dP = dO V^T
//   dS = P o (dP - D),  D_i = sum_d dO[i,d] O[i,d]           dQ = scale dS K      dK = scale dS^T Q
//
// Three kernels, no atomics, every product on v_mfma_f32_32x32x16_bf16 (C tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2)
// + 4 (lane >> 5)):
//   attn_fwd_kernel       a workgroup owns 128 query rows (a wave 32 of them; Q fragments stay in registers as B operands, lane <->
//                         query row), streams 64-row K / V blocks through LDS (K row-major, V transposed), S^T tiles [key row
//                         (regs), query row (lane)]: the online softmax is per-lane arithmetic + one lane ^ 32 exchange per tile;
//                         P^T rounded to bf16 IS the B operand of O^T[d, i] += V^T[d, j] P^T[j, i].  Writes O and the log-sum-exp.
//   attn_bwd_dq_kernel    same shape, Q and dO fragments in registers, computes S^T and dP^T tiles.  The dS^T tile, rounded to
//                         bf16, IS the B operand of dQ^T[d, i] += K^T[d, j] dS^T[j, i] with the contraction index taken in the
//                         tile's register order (k-step s of a lane half h holds rows 16 s + 4 h + {0..3} and 16 s + 8 + 4 h +
//                         {0..3}); the A operand K^T is read from a transposed LDS copy with two 8-byte reads in that same order.
//                         Also writes D.
//   attn_bwd_dkdv_kernel  a workgroup owns 64 key rows (K, V fragments in registers, lane <-> key row), streams 64-row Q / dO
//                         blocks (+ transposed copies), computes S and dP tiles [query row (regs), key row (lane)]; P and dS are
//                         the B operands of dV^T[d, j] += dO^T[d, i] P[i, j] and dK^T[d, j] += Q^T[d, i] dS[i, j]; a wave
//                         accumulates half of d.
// The mask is read as BITS: attn_mask_bits_kernel packs the [B, 1, T, T] boolean mask once per step (the same mask serves every
// layer and head) into row words (bit c of word w of row i = mask[i, 32 w + c]) and column words, plus one byte per 32 x 32 tile
// that says whether anything in it is live: a lane's 32 mask bits of a tile are ONE dword, dead tiles (the causal upper
// triangle, padding) cost nothing.
// Round 6: each kernel exists in TWO forms with the same arithmetic (bit-identical outputs, tools/attn_ab.py).
// The first forms (attn_fwd / attn_bwd_dq / attn_bwd_dkdv) stage a block through registers into padded LDS tiles plus transposed
// copies, two barriers per block.  The second forms (attn_fwd2 / attn_bwd_dq2 / attn_bwd_dkdv2, the default; DALM_ATTN_FWD=1 /
// DALM_ATTN_DKDV=1 select the first) move a block HBM -> LDS by LDS-DMA into un-padded swizzled images, two stages (block n + 1
// in flight while block n is multiplied), read the "transposed" operands with ds_read_b64_tr_b16 and take the workgroup's own rows
// straight from HBM in fragment shape: one barrier per block, half the LDS time.  Row blocks are launched heavy-first.
// cfg3 layer (B 18, H 32, T 256, hd 128, causal + left padding): forward 46.8 -> 35.0 us, backward 153.8 -> 113.4 us.
// Algorithmic bytes: forward q, k, v read + o written = 4 B H T hd el (151 MB at cfg3); backward q, k, v, o, dO read + dq, dk, dv
// written = 8 B H T hd el (302 MB; the two launches together move 1.5 x that: each re-reads q, k, v, dO); algorithmic flops
// 2 / 5 GEMMs x 2 T^2 hd per head over the live tiles (dq and dk/dv each recompute S and dP: 7 are executed).
#include "common.hpp"

namespace dalm {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTRow = 2 * 64 + 8;        // bytes of a [d][64 rows] transposed LDS row (8-byte reads stay aligned and spread)
template <int HD>
struct AT {                              // head width 128 (Llama-2-7b) or 64 (Falcon-7b)
  static constexpr int LROW = 2 * HD + 16;   // bytes of a [row][HD] LDS row (16 bytes of padding: conflict-free 16-byte reads)
  static constexpr int KK = HD / 16;         // k-steps of a contraction over d
  static constexpr int ND = HD / 32;         // 32-column blocks of d
  static constexpr int CH = HD / 8;          // 16-byte chunks per row
  static constexpr int NCW = CH / 4;         // chunks a wave stages per row of a 64-row block
  static constexpr int RM = 64 * LROW;       // a 64-row block, row-major
  static constexpr int TR = HD * kTRow;      // the same block transposed
  static constexpr int N128 = CH / 2;        // 16-byte pieces a thread holds of the workgroup's own 128 rows ...
  static constexpr int N64 = CH / 4;         // ... of its own 64 rows
  static constexpr int OCC = HD == 128 ? 2 : 3;
};

struct AttnBwdParams {
  const unsigned short *q, *k, *v, *o, *d_o;
  const float* lse;
  const uint32_t *bits_rows, *bits_cols;
  const unsigned char* live;
  unsigned short *dq, *dk, *dv;
  float* delta;
  int B, H, T, W;                        // W = ceil(T / 32) mask words per row
  float scale;
  int64_t s[8][3];                       // element strides (batch, head, row) of q, k, v, o, dO, dq, dk, dv
  const unsigned long long* seed;        // attention dropout (BERT's attention_probs_dropout_prob): device seed word, NULL = off
  unsigned int salt, thresh;             // per-call salt; keep an element when its 16-bit field >= thresh = round(p 65536)
  float keep_scale;                      // 1 / (1 - p)
  const unsigned short *cos, *sin;       // backward only, may be NULL: q and k are ROTATED tensors (rotary embedding applied by
  int64_t cs_b, cs_t;                    // dalm_rope_qk); dq and dk leave as gradients of the UN-rotated ones.  [B or 1, T, hd]
  const int* cu;                         // PACKED (un-padded) layout, may be NULL: sequence b = token rows cu[b] .. cu[b + 1] - 1 of
};                                       // [n_tokens, H, hd] tensors (batch strides unused), T = the longest sequence; lse / delta /
                                         // mask words keep the padded [B, H, T] / [B, 32 W, W] layout (they are small)

// where sequence b starts (token rows) and how many rows it has
struct Seq { int64_t r0; int T; };
__device__ __forceinline__ Seq seq_of(const AttnBwdParams& p, int b) {
  if (p.cu) { const int a = p.cu[b]; return {a, p.cu[b + 1] - a}; }
  return {0, p.T};
}
// element offset of (sequence b, head h, row 0) in tensor `i` of the stride table
__device__ __forceinline__ int64_t base_off(const AttnBwdParams& p, int i, int b, int h, const Seq& sq) {
  return (p.cu ? sq.r0 * p.s[i][2] : b * p.s[i][0]) + h * p.s[i][1];
}
__device__ __forceinline__ int64_t cs_off(const AttnBwdParams& p, int b, const Seq& sq) { return p.cu ? sq.r0 * p.cs_t : b * p.cs_b; }

__device__ __forceinline__ uint4 ld16(const unsigned short* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ float dot8(const uint4& a, const uint4& b) {
  const unsigned int x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s = fmaf(__uint_as_float(x[i] << 16), __uint_as_float(y[i] << 16), s);
    s = fmaf(__uint_as_float(x[i] & 0xffff0000u), __uint_as_float(y[i] & 0xffff0000u), s);
  }
  return s;
}

// one 64-row block of a [rows][128] bf16 tensor -> registers (lane <-> row, wave w takes the 16-byte chunks 4 w .. 4 w + 3) ...
template <int HD>
__device__ __forceinline__ void block_load(const unsigned short* base, int64_t row_stride, int row0, int T, int w, int l,
                                           uint4 (&v)[AT<HD>::NCW]) {
  const int row = row0 + l;
  const bool ok = row < T;
  const unsigned short* src = base + static_cast<int64_t>(row) * row_stride;
#pragma unroll
  for (int n = 0; n < AT<HD>::NCW; ++n) v[n] = ok ? ld16(src + 8 * (AT<HD>::NCW * w + n)) : make_uint4(0u, 0u, 0u, 0u);
}
// ... -> LDS, row-major (`rm`) and, when `tr` is given, transposed ([d][row]: the 2-byte stores of a wave are one contiguous
// 128-byte run)
template <int HD>
__device__ __forceinline__ void block_store(const uint4 (&v)[AT<HD>::NCW], unsigned char* rm, unsigned char* tr, int w, int l) {
#pragma unroll
  for (int n = 0; n < AT<HD>::NCW; ++n) {
    const int c = AT<HD>::NCW * w + n;
    if (rm) *reinterpret_cast<uint4*>(rm + l * AT<HD>::LROW + 16 * c) = v[n];
    if (tr) {
      const unsigned int q[4] = {v[n].x, v[n].y, v[n].z, v[n].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        *reinterpret_cast<unsigned short*>(tr + (8 * c + 2 * e) * kTRow + 2 * l) = static_cast<unsigned short>(q[e] & 0xffffu);
        *reinterpret_cast<unsigned short*>(tr + (8 * c + 2 * e + 1) * kTRow + 2 * l) = static_cast<unsigned short>(q[e] >> 16);
      }
    }
  }
}

// the workgroup's own rows, coalesced (HD / 8 lanes per row): thread t holds chunk t % CH of rows t / CH + (256 / CH) n
template <int HD, int N>
__device__ __forceinline__ void rows_load(const unsigned short* base, int64_t row_stride, int row0, int T, int t, uint4 (&v)[N]) {
  constexpr int CH = AT<HD>::CH;
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const int row = row0 + t / CH + (256 / CH) * n;
    v[n] = row < T ? ld16(base + static_cast<int64_t>(row) * row_stride + 8 * (t % CH)) : make_uint4(0u, 0u, 0u, 0u);
  }
}
template <int HD, int N>
__device__ __forceinline__ void rows_store(const uint4 (&v)[N], unsigned char* rm, int t) {
  constexpr int CH = AT<HD>::CH;
#pragma unroll
  for (int n = 0; n < N; ++n) *reinterpret_cast<uint4*>(rm + (t / CH + (256 / CH) * n) * AT<HD>::LROW + 16 * (t % CH)) = v[n];
}
// the wave's 32 rows as B operands: lane <-> row, HD / 16 k-steps of 16
template <int HD>
__device__ __forceinline__ void rows_frags(const unsigned char* rm, int tile, int l31, int hi, bf16x8 (&f)[AT<HD>::KK]) {
#pragma unroll
  for (int kk = 0; kk < AT<HD>::KK; ++kk)
    f[kk] = *reinterpret_cast<const bf16x8*>(rm + (32 * tile + l31) * AT<HD>::LROW + 32 * kk + 16 * hi);
}

// A operand of a product whose contraction index runs over the ROWS of a 32 x 32 C tile held as the B operand:
// k-step s, lane half h: rows 16 s + 4 h + {0..3}, then 16 s + 8 + 4 h + {0..3}
__device__ __forceinline__ bf16x8 ld_tr_frag(const unsigned char* tr, int drow, int r0, int s, int hi) {
  const unsigned char* a = tr + drow * kTRow + 2 * (r0 + 16 * s + 4 * hi);
  const uint2 lo = *reinterpret_cast<const uint2*>(a), up = *reinterpret_cast<const uint2*>(a + 16);
  return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, up.x, up.y));
}

// a wave's [32 rows][128] accumulators held transposed (acc[dblk]: row d = 32 dblk + .., column = lane & 31 <-> the wave's row)
// -> LDS [rows][LROW] as bf16
template <int HD, int ND>
__device__ __forceinline__ void spill_transposed(const f32x16 (&acc)[ND], int d0, float mul, unsigned char* out, int tile, int l31, int hi) {
#pragma unroll
  for (int dblk = 0; dblk < ND; ++dblk)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint2 pk;
      pk.x = pack_bf16x2(acc[dblk][4 * q] * mul, acc[dblk][4 * q + 1] * mul);
      pk.y = pack_bf16x2(acc[dblk][4 * q + 2] * mul, acc[dblk][4 * q + 3] * mul);
      *reinterpret_cast<uint2*>(out + (32 * tile + l31) * AT<HD>::LROW + 2 * (32 * (d0 + dblk) + 8 * q + 4 * hi)) = pk;
    }
}
template <int HD, int N>
__device__ __forceinline__ void store_rows(const unsigned char* out, unsigned short* dst, int64_t row_stride, int row0, int T, int t) {
  constexpr int CH = AT<HD>::CH;
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const int row = t / CH + (256 / CH) * n, c = t % CH;
    if (row0 + row < T)
      *reinterpret_cast<uint4*>(dst + static_cast<int64_t>(row0 + row) * row_stride + 8 * c) =
          out ? *reinterpret_cast<const uint4*>(out + row * AT<HD>::LROW + 16 * c) : make_uint4(0u, 0u, 0u, 0u);
  }
}

// store_rows for a gradient of a ROTATED tensor: what leaves is the gradient of the tensor BEFORE dalm_rope_qk, i.e. that kernel's
// backward (csrc/tower.hip rope_qk_kernel, transformers' apply_rotary_pos_emb differentiated op by op) applied to the bf16 rows
// of the LDS tile, with its rounding points: lower half  o1 = rb(rb(g1 c1) + rb(g2 s2)),  upper half  o2 = rb(rb(g2 c2) - rb(g1 s1)).
template <int HD, int N>
__device__ __forceinline__ void store_rows_unrope(const unsigned char* out, unsigned short* dst, int64_t row_stride, int row0, int T,
                                                  int t, const unsigned short* cosb, const unsigned short* sinb, int64_t cs_t) {
#pragma clang fp contract(off)   // separate multiplies and add, as the eager chain's kernels
  constexpr int CH = AT<HD>::CH;
#pragma unroll
  for (int n = 0; n < N; ++n) {
    const int row = t / CH + (256 / CH) * n, c = t % CH, pc = c ^ (CH / 2);
    if (row0 + row >= T) continue;
    const uint4 own = *reinterpret_cast<const uint4*>(out + row * AT<HD>::LROW + 16 * c);
    const uint4 oth = *reinterpret_cast<const uint4*>(out + row * AT<HD>::LROW + 16 * pc);
    const uint4 cv = ld16(cosb + static_cast<int64_t>(row0 + row) * cs_t + 8 * c);
    const uint4 sv = ld16(sinb + static_cast<int64_t>(row0 + row) * cs_t + 8 * pc);
    const unsigned int g[4] = {own.x, own.y, own.z, own.w}, o[4] = {oth.x, oth.y, oth.z, oth.w};
    const unsigned int cc[4] = {cv.x, cv.y, cv.z, cv.w}, ss[4] = {sv.x, sv.y, sv.z, sv.w};
    const float sgn = c < CH / 2 ? 1.0f : -1.0f;
    unsigned int r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float gx = u ? __uint_as_float(g[e] & 0xffff0000u) : __uint_as_float(g[e] << 16);
        const float ox = u ? __uint_as_float(o[e] & 0xffff0000u) : __uint_as_float(o[e] << 16);
        const float cx = u ? __uint_as_float(cc[e] & 0xffff0000u) : __uint_as_float(cc[e] << 16);
        const float sx = u ? __uint_as_float(ss[e] & 0xffff0000u) : __uint_as_float(ss[e] << 16);
        const float a = bf16_to_f32(f32_to_bf16(gx * cx));
        const float bterm = bf16_to_f32(f32_to_bf16(ox * sx));
        v[u] = a + sgn * bterm;
      }
      r[e] = pack_bf16x2(v[0], v[1]);
    }
    *reinterpret_cast<uint4*>(dst + static_cast<int64_t>(row0 + row) * row_stride + 8 * c) = make_uint4(r[0], r[1], r[2], r[3]);
  }
}

// Attention dropout: P o M / (1 - p) in front of the P V product (torch applies it there), M regenerated by every kernel from
// (device seed word, per-call salt, element index) - never stored.  One 32-bit hash serves the PAIR of elements (i, j), (i, j + 1),
// j even (T even): 16-bit fields compared with round(p 65536).  oracle/attn_dropout.py restates it; tests pin every bit.
struct AttnDrop { unsigned int a, b, thresh; float ks; };
__device__ __forceinline__ unsigned int mix32(unsigned int x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ AttnDrop attn_drop(const AttnBwdParams& p) {
  AttnDrop d;
  d.thresh = p.seed ? p.thresh : 0u;
  d.ks = p.seed ? p.keep_scale : 1.0f;
  const unsigned long long s = p.seed ? *p.seed : 0ull;
  d.a = mix32(static_cast<unsigned int>(s) ^ (p.salt * 0x9E3779B9u));
  d.b = mix32(static_cast<unsigned int>(s >> 32) + p.salt + 0x85ebca6bu) | 1u;
  return d;
}
// both fields of the pair that holds element index c (c even): low half = element c, high half = element c + 1
__device__ __forceinline__ unsigned int drop_pair(const AttnDrop& d, unsigned int c) { return mix32(((c >> 1) ^ d.a) + d.b); }

// launch index -> (row block, head, batch).  Workgroups are dispatched in launch order and a row block's work grows with its index
// (causal mask: query block i multiplies i + 1 key blocks) or falls with it (key block j meets nblk - j query blocks): the HEAVY
// blocks of every (batch, head) are launched first, the light ones fill the tail (interleaved, the last workgroups to start were
// heavy ones and ran alone for a third of the kernel's time).  Both row blocks of one (batch, head) keep the same launch index
// mod 8 = the same XCD: the later one finds K / V (or Q / dO) in that L2 or in the Infinity Cache.
__device__ __forceinline__ bool block_coords(const AttnBwdParams& p, int nblk, bool last_block_first, int& blk, int& h, int& b) {
  const int pairs8 = (p.B * p.H + 7) & ~7;
  const int n = blockIdx.x, a = n / pairs8, pair = n - a * pairs8;
  blk = last_block_first ? nblk - 1 - a : a;
  if (pair >= p.B * p.H) return false;
  h = pair % p.H;
  b = pair / p.H;
  return true;
}

// which 32-wide sub-blocks of the other axis have a live tile against this workgroup's NT 32-row tiles: bit jj (T <= 2048)
template <int NT>
__device__ __forceinline__ unsigned long long need_mask(const AttnBwdParams& p, int b, int own32, bool own_is_row, int l) {
  unsigned int v = 0u;
  if (l < p.W) {
#pragma unroll
    for (int a = 0; a < NT; ++a)
      if (own32 + a < p.W)
        v |= own_is_row ? p.live[(static_cast<int64_t>(b) * p.W + own32 + a) * p.W + l]
                        : p.live[(static_cast<int64_t>(b) * p.W + l) * p.W + own32 + a];
  }
  return __builtin_amdgcn_ballot_w64(v != 0u);
}

template <int HD, bool DROP>
__global__ __launch_bounds__(256, DROP ? 2 : AT<HD>::OCC) void attn_bwd_dq_kernel(const AttnBwdParams p) {
  using A = AT<HD>;
  constexpr int LDS = (2 * A::RM + A::TR) > (128 * A::LROW + 512) ? (2 * A::RM + A::TR) : (128 * A::LROW + 512);
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS];
  unsigned char* Ks = lds;
  unsigned char* Vs = lds + A::RM;
  unsigned char* KT = lds + 2 * A::RM;
  float* dl_s = reinterpret_cast<float*>(lds + 128 * A::LROW);          // prologue only: [128] D of the workgroup's rows
  const int t = threadIdx.x, w = t >> 6, l = t & 63, l31 = l & 31, hi = l >> 5;
  int blk, h, b;
  if (!block_coords(p, (p.T + 127) >> 7, true, blk, h, b)) return;
  const int i0 = blk * 128;
  const int i = i0 + 32 * w + l31;
  const int64_t bh = static_cast<int64_t>(b) * p.H + h;
  const Seq sq = seq_of(p, b);
  const int T = sq.T;                                            // rows of THIS sequence (p.T: the layout's row count)
  unsigned short* dq_base = p.dq + base_off(p, 5, b, h, sq);
  const unsigned long long need = need_mask<4>(p, b, i0 >> 5, true, l);
  if (need == 0ull) {                                            // padding rows only: zero gradient, nothing to read
    store_rows<HD, A::N128>(nullptr, dq_base, p.s[5][2], i0, T, t);
    return;
  }
  const unsigned short* kbase = p.k + base_off(p, 1, b, h, sq);
  const unsigned short* vbase = p.v + base_off(p, 2, b, h, sq);

  bf16x8 Qb[A::KK], Gb[A::KK];
  float Dl;
  {
    uint4 qv[A::N128], gv[A::N128], ov[A::N128];
    rows_load<HD, A::N128>(p.q + base_off(p, 0, b, h, sq), p.s[0][2], i0, T, t, qv);
    rows_load<HD, A::N128>(p.d_o + base_off(p, 4, b, h, sq), p.s[4][2], i0, T, t, gv);
    rows_load<HD, A::N128>(p.o + base_off(p, 3, b, h, sq), p.s[3][2], i0, T, t, ov);
    rows_store<HD, A::N128>(qv, lds, t);
#pragma unroll
    for (int n = 0; n < A::N128; ++n) {                          // D = rowsum(dO o O): HD / 8 consecutive lanes hold one row
      float d = dot8(gv[n], ov[n]);
#pragma unroll
      for (int off = 1; off < A::CH; off <<= 1) d += __shfl_xor(d, off, 64);
      const int row = t / A::CH + (256 / A::CH) * n;
      if (t % A::CH == 0) {
        dl_s[row] = d;
        if (i0 + row < T) p.delta[bh * p.T + i0 + row] = d;
      }
    }
    __syncthreads();
    rows_frags<HD>(lds, w, l31, hi, Qb);
    Dl = dl_s[32 * w + l31];
    __syncthreads();
    rows_store<HD, A::N128>(gv, lds, t);
    __syncthreads();
    rows_frags<HD>(lds, w, l31, hi, Gb);
  }
  const float nl = i < T ? -p.lse[bh * p.T + i] * kLog2e : 0.f;
  const float c1 = p.scale * kLog2e;
  const int Tp = 32 * p.W;

  f32x16 acc[A::ND];
#pragma unroll
  for (int d = 0; d < A::ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  // the live 64-row blocks, one ahead: block n + 1's rows are in flight (registers) while block n is multiplied
  const int nJ = (p.T + 63) >> 6;
  unsigned int blocks = 0u;                                    // bit jb: K / V block jb has a live tile (T <= 2048: 32 blocks)
  for (int jb = 0; jb < nJ; ++jb) blocks |= (((need >> (2 * jb)) & 3ull) != 0ull ? 1u : 0u) << jb;
  uint32_t nword[2];
  uint4 kv[A::NCW], vv[A::NCW];
  auto fetch = [&](int jb) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
      nword[c] = (i < Tp && 2 * jb + c < p.W) ? p.bits_rows[(static_cast<int64_t>(b) * Tp + i) * p.W + 2 * jb + c] : 0u;
    block_load<HD>(kbase, p.s[1][2], 64 * jb, T, w, l, kv);
    block_load<HD>(vbase, p.s[2][2], 64 * jb, T, w, l, vv);
  };
  fetch(__builtin_ctz(blocks));
  const AttnDrop drop = attn_drop(p);
  const unsigned int cbase = (static_cast<unsigned int>(bh) * p.T + i) * p.T;      // element index of (row i, column 0)
  (void)drop; (void)cbase;
  while (blocks) {
    const int jb = __builtin_ctz(blocks);
    (void)jb;
    blocks &= blocks - 1u;
    uint32_t word[2] = {nword[0], nword[1]};
    __syncthreads();                                           // the previous block's fragments have been read
    block_store<HD>(kv, Ks, KT, w, l);
    block_store<HD>(vv, Vs, nullptr, w, l);
    __syncthreads();
    if (blocks) fetch(__builtin_ctz(blocks));
#pragma unroll
    for (int js = 0; js < 2; ++js) {
      if (__builtin_amdgcn_ballot_w64(word[js] != 0u) == 0ull) continue;
      f32x16 St, Pt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { St[r] = 0.f; Pt[r] = 0.f; }
      const unsigned char* ka = Ks + (32 * js + l31) * A::LROW + 16 * hi;
      const unsigned char* va = Vs + (32 * js + l31) * A::LROW + 16 * hi;
#pragma unroll
      for (int kk = 0; kk < A::KK; ++kk) {
        St = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(ka + 32 * kk), Qb[kk], St, 0, 0, 0);
        Pt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(va + 32 * kk), Gb[kk], Pt, 0, 0, 0);
      }
      unsigned int pk[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float ds[2];
        const int jl0 = (r & 3) + 8 * (r >> 2) + 4 * hi;       // even: (jl0, jl0 + 1) share a hash
        unsigned int hw = 0xffffffffu;
        if constexpr (DROP) hw = drop_pair(drop, cbase + 64 * jb + 32 * js + jl0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int jl = jl0 + u;
          const float pv = ((word[js] >> jl) & 1u) ? __builtin_amdgcn_exp2f(fmaf(St[r + u], c1, nl)) : 0.f;
          float dpe = Pt[r + u];
          if constexpr (DROP) dpe = ((u ? hw >> 16 : hw & 0xffffu) >= drop.thresh) ? dpe * drop.ks : 0.f;
          ds[u] = pv * (dpe - Dl);
        }
        pk[r >> 1] = pack_bf16x2(ds[0], ds[1]);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 dsb = __builtin_bit_cast(bf16x8, make_uint4(pk[4 * s], pk[4 * s + 1], pk[4 * s + 2], pk[4 * s + 3]));
#pragma unroll
        for (int d = 0; d < A::ND; ++d)
          acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr_frag(KT, 32 * d + l31, 32 * js, s, hi), dsb, acc[d], 0, 0, 0);
      }
    }
  }
  __syncthreads();
  spill_transposed<HD, A::ND>(acc, 0, p.scale, lds, w, l31, hi);
  __syncthreads();
  if (p.cos) store_rows_unrope<HD, A::N128>(lds, dq_base, p.s[5][2], i0, T, t, p.cos + cs_off(p, b, sq), p.sin + cs_off(p, b, sq), p.cs_t);
  else store_rows<HD, A::N128>(lds, dq_base, p.s[5][2], i0, T, t);
}

// Forward: O = softmax(scale Q K^T + mask) V and the rows' log-sum-exp (natural log), the dq kernel's structure with the
// roles turned: S^T tiles [key row (regs), query row (lane)] put a query row's scores into the registers of ONE lane pair
// (lanes l and l ^ 32), so the running maximum / sum of the online softmax are per-lane scalars and one cross-half exchange per
// tile; P^T rounded to bf16 is the B operand of O^T[d, i] += V^T[d, j] P^T[j, i] (V^T from a transposed LDS copy).
// torch's memory-efficient forward takes 115 - 123 us at cfg3 (B 18, H 32, T 256; profiles/r05_step_by_stream.txt).
// Algorithmic bytes: q, k, v read + o written = 4 B H T hd el (151 MB at cfg3).
template <int HD, bool DROP>
__global__ __launch_bounds__(256, AT<HD>::OCC) void attn_fwd_kernel(const AttnBwdParams p) {
  using A = AT<HD>;
  constexpr int LDS = (A::RM + A::TR) > 128 * A::LROW ? (A::RM + A::TR) : 128 * A::LROW;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS];
  unsigned char* Ks = lds;                                     // [64 key rows][LROW]
  unsigned char* VT = lds + A::RM;                             // [HD][kTRow]: V transposed
  const int t = threadIdx.x, w = t >> 6, l = t & 63, l31 = l & 31, hi = l >> 5;
  int blk, h, b;
  if (!block_coords(p, (p.T + 127) >> 7, true, blk, h, b)) return;
  const int i0 = blk * 128;
  const int i = i0 + 32 * w + l31;
  const int64_t bh = static_cast<int64_t>(b) * p.H + h;
  const Seq sq = seq_of(p, b);
  const int T = sq.T;
  unsigned short* o_base = p.dq + base_off(p, 5, b, h, sq);               // the output travels in the dq slot
  float* lse_out = p.delta;                                                // and the log-sum-exp in the delta slot
  const unsigned long long need = need_mask<4>(p, b, i0 >> 5, true, l);
  if (need == 0ull) {                                          // rows without a live key: zero output (as torch returns)
    store_rows<HD, A::N128>(nullptr, o_base, p.s[5][2], i0, T, t);
    if (t < 128 && i0 + t < T) lse_out[bh * p.T + i0 + t] = 0.f;
    return;
  }
  const unsigned short* kbase = p.k + base_off(p, 1, b, h, sq);
  const unsigned short* vbase = p.v + base_off(p, 2, b, h, sq);
  const int Tp = 32 * p.W;
  const int nJ = (p.T + 63) >> 6;
  unsigned int blocks = 0u;
  for (int jb = 0; jb < nJ; ++jb) blocks |= (((need >> (2 * jb)) & 3ull) != 0ull ? 1u : 0u) << jb;
  uint32_t nword[2];
  uint4 kv[A::NCW], vv[A::NCW];
  auto fetch = [&](int jb) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
      nword[c] = (i < Tp && 2 * jb + c < p.W) ? p.bits_rows[(static_cast<int64_t>(b) * Tp + i) * p.W + 2 * jb + c] : 0u;
    block_load<HD>(kbase, p.s[1][2], 64 * jb, T, w, l, kv);
    block_load<HD>(vbase, p.s[2][2], 64 * jb, T, w, l, vv);
  };
  fetch(__builtin_ctz(blocks));

  bf16x8 Qb[A::KK];
  {
    uint4 qv[A::N128];
    rows_load<HD, A::N128>(p.q + base_off(p, 0, b, h, sq), p.s[0][2], i0, T, t, qv);
    rows_store<HD, A::N128>(qv, lds, t);
    __syncthreads();
    rows_frags<HD>(lds, w, l31, hi, Qb);
  }
  const float c1 = p.scale * kLog2e;
  f32x16 acc[A::ND];
#pragma unroll
  for (int d = 0; d < A::ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;                              // running maximum (log2 domain) of the row; this lane's share of the sum

  const AttnDrop drop = attn_drop(p);
  const unsigned int cbase = (static_cast<unsigned int>(bh) * p.T + i) * p.T;
  (void)drop; (void)cbase;
  while (blocks) {
    const int jb = __builtin_ctz(blocks);
    (void)jb;
    blocks &= blocks - 1u;
    uint32_t word[2] = {nword[0], nword[1]};
    __syncthreads();                                           // the previous block's (or Q's) fragments have been read
    block_store<HD>(kv, Ks, nullptr, w, l);
    block_store<HD>(vv, nullptr, VT, w, l);                    // V: transposed copy only
    __syncthreads();
    if (blocks) fetch(__builtin_ctz(blocks));
#pragma unroll
    for (int js = 0; js < 2; ++js) {
      if (__builtin_amdgcn_ballot_w64(word[js] != 0u) == 0ull) continue;
      f32x16 St;
#pragma unroll
      for (int r = 0; r < 16; ++r) St[r] = 0.f;
      const unsigned char* ka = Ks + (32 * js + l31) * A::LROW + 16 * hi;
#pragma unroll
      for (int kk = 0; kk < A::KK; ++kk)
        St = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(ka + 32 * kk), Qb[kk], St, 0, 0, 0);
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jl = (r & 3) + 8 * (r >> 2) + 4 * hi;
        St[r] = ((word[js] >> jl) & 1u) ? St[r] * c1 : -INFINITY;
        mx = fmaxf(mx, St[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m, mx);
      const float base = mn == -INFINITY ? 0.f : mn;             // a row with nothing live so far: every p below is exp2(-inf) = 0
      const float alpha = __builtin_amdgcn_exp2f(m - base);      // m = -inf: 0
      m = mn;
      float ps = 0.f;
      unsigned int pk[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(St[r] - base), p1 = __builtin_amdgcn_exp2f(St[r + 1] - base);
        ps += p0 + p1;                                           // the softmax denominator counts dropped elements too
        if constexpr (DROP) {
          const unsigned int hw = drop_pair(drop, cbase + 64 * jb + 32 * js + (r & 3) + 8 * (r >> 2) + 4 * hi);
          pk[r >> 1] = pack_bf16x2((hw & 0xffffu) >= drop.thresh ? p0 * drop.ks : 0.f, (hw >> 16) >= drop.thresh ? p1 * drop.ks : 0.f);
        } else {
          pk[r >> 1] = pack_bf16x2(p0, p1);
        }
      }
      lsum = fmaf(lsum, alpha, ps);
#pragma unroll
      for (int d = 0; d < A::ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] *= alpha;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pb = __builtin_bit_cast(bf16x8, make_uint4(pk[4 * s], pk[4 * s + 1], pk[4 * s + 2], pk[4 * s + 3]));
#pragma unroll
        for (int d = 0; d < A::ND; ++d)
          acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr_frag(VT, 32 * d + l31, 32 * js, s, hi), pb, acc[d], 0, 0, 0);
      }
    }
  }
  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
  if (hi == 0 && i < T) lse_out[bh * p.T + i] = ltot > 0.f ? (m + __builtin_amdgcn_logf(ltot)) * kLn2 : 0.f;
  __syncthreads();
  spill_transposed<HD, A::ND>(acc, 0, inv, lds, w, l31, hi);
  __syncthreads();
  store_rows<HD, A::N128>(lds, o_base, p.s[5][2], i0, T, t);
}

template <int HD>
constexpr int dkdv_lds() { return 2 * AT<HD>::RM + 2 * AT<HD>::TR + 2 * 64 * 4; }

// 64 key rows per workgroup; wave (jt, dh) = (w >> 1, w & 1) computes the S and dP tiles of key tile jt (both waves of a tile
// do: 16 of the 24 MFMAs per tile and wave at HD = 128) and accumulates dV^T / dK^T for the HD / 2 columns d of half dh only -
// half the accumulator registers, which is what lets TWO workgroups share a CU (the one-wave-per-SIMD form spent its time
// waiting: 113 us against 97 us, tools/attn_bench.py).
// (Fetching block n + 1 while block n is multiplied, as the dq kernel does, costs this kernel 23 spilled registers: 91 -> 124 us.)
template <int HD, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(const AttnBwdParams p) {
  using A = AT<HD>;
  constexpr int NDH = A::ND / 2;                               // d blocks of this wave's half
  extern __shared__ __attribute__((aligned(16))) unsigned char dlds[];
  unsigned char* Qs = dlds;
  unsigned char* Gs = dlds + A::RM;
  unsigned char* QT = dlds + 2 * A::RM;
  unsigned char* GT = dlds + 2 * A::RM + A::TR;
  float* nl_s = reinterpret_cast<float*>(dlds + 2 * A::RM + 2 * A::TR);
  float* dl_s = nl_s + 64;
  const int t = threadIdx.x, w = t >> 6, l = t & 63, l31 = l & 31, hi = l >> 5, jt = w >> 1, dh = w & 1;
  int blk, h, b;
  if (!block_coords(p, (p.T + 63) >> 6, false, blk, h, b)) return;
  const int j0 = blk * 64;
  const int j = j0 + 32 * jt + l31;
  const int64_t bh = static_cast<int64_t>(b) * p.H + h;
  const Seq sq = seq_of(p, b);
  const int T = sq.T;
  unsigned short* dk_base = p.dk + base_off(p, 6, b, h, sq);
  unsigned short* dv_base = p.dv + base_off(p, 7, b, h, sq);
  const unsigned long long need = need_mask<2>(p, b, j0 >> 5, false, l);
  if (need == 0ull) {
    store_rows<HD, A::N64>(nullptr, dk_base, p.s[6][2], j0, T, t);
    store_rows<HD, A::N64>(nullptr, dv_base, p.s[7][2], j0, T, t);
    return;
  }
  const unsigned short* qbase = p.q + base_off(p, 0, b, h, sq);
  const unsigned short* gbase = p.d_o + base_off(p, 4, b, h, sq);

  bf16x8 Kb[A::KK], Vb[A::KK];
  {
    uint4 kv[A::N64], vv[A::N64];
    rows_load<HD, A::N64>(p.k + base_off(p, 1, b, h, sq), p.s[1][2], j0, T, t, kv);
    rows_load<HD, A::N64>(p.v + base_off(p, 2, b, h, sq), p.s[2][2], j0, T, t, vv);
    rows_store<HD, A::N64>(kv, dlds, t);
    rows_store<HD, A::N64>(vv, dlds + A::RM, t);
    __syncthreads();
    rows_frags<HD>(dlds, jt, l31, hi, Kb);
    rows_frags<HD>(dlds + A::RM, jt, l31, hi, Vb);
  }
  const float c1 = p.scale * kLog2e;
  const int Tp = 32 * p.W;
  f32x16 dVt[NDH], dKt[NDH];
#pragma unroll
  for (int d = 0; d < NDH; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dVt[d][r] = 0.f; dKt[d][r] = 0.f; }

  const AttnDrop drop = attn_drop(p);
  (void)drop;
  const int nI = (p.T + 63) >> 6;
  for (int ib = 0; ib < nI; ++ib) {
    if (((need >> (2 * ib)) & 3ull) == 0ull) continue;
    uint32_t word[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
      word[c] = (j < Tp && 2 * ib + c < p.W) ? p.bits_cols[(static_cast<int64_t>(b) * Tp + j) * p.W + 2 * ib + c] : 0u;
    uint4 qv[A::NCW], gv[A::NCW];
    block_load<HD>(qbase, p.s[0][2], 64 * ib, T, w, l, qv);
    block_load<HD>(gbase, p.s[4][2], 64 * ib, T, w, l, gv);
    float nlv = 0.f, dlv = 0.f;
    if (t < 64 && 64 * ib + t < T) {
      nlv = -p.lse[bh * p.T + 64 * ib + t] * kLog2e;
      dlv = p.delta[bh * p.T + 64 * ib + t];
    }
    __syncthreads();
    block_store<HD>(qv, Qs, QT, w, l);
    block_store<HD>(gv, Gs, GT, w, l);
    if (t < 64) { nl_s[t] = nlv; dl_s[t] = dlv; }
    __syncthreads();
#pragma unroll
    for (int is = 0; is < 2; ++is) {
      if (__builtin_amdgcn_ballot_w64(word[is] != 0u) == 0ull) continue;
      f32x16 S, dP;
#pragma unroll
      for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
      const unsigned char* qa = Qs + (32 * is + l31) * A::LROW + 16 * hi;
      const unsigned char* ga = Gs + (32 * is + l31) * A::LROW + 16 * hi;
#pragma unroll
      for (int kk = 0; kk < A::KK; ++kk) {
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(qa + 32 * kk), Kb[kk], S, 0, 0, 0);
        dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(ga + 32 * kk), Vb[kk], dP, 0, 0, 0);
      }
      unsigned int ppk[8], dpk[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 nl4 = *reinterpret_cast<const float4*>(nl_s + 32 * is + 8 * q + 4 * hi);
        const float4 dl4 = *reinterpret_cast<const float4*>(dl_s + 32 * is + 8 * q + 4 * hi);
        const float nl[4] = {nl4.x, nl4.y, nl4.z, nl4.w}, dl[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
        float pv[4], ds[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int il = 8 * q + 4 * hi + u;
          pv[u] = ((word[is] >> il) & 1u) ? __builtin_amdgcn_exp2f(fmaf(S[4 * q + u], c1, nl[u])) : 0.f;
          float dpe = dP[4 * q + u];
          if constexpr (DROP) {
            const unsigned int c = (static_cast<unsigned int>(bh) * p.T + 64 * ib + 32 * is + il) * p.T + j;
            const unsigned int hw = drop_pair(drop, c & ~1u);
            const bool keep = ((c & 1u) ? hw >> 16 : hw & 0xffffu) >= drop.thresh;
            dpe = keep ? dpe * drop.ks : 0.f;
            ds[u] = pv[u] * (dpe - dl[u]);
            pv[u] = keep ? pv[u] * drop.ks : 0.f;                // what multiplies dO in dV = (P o M / (1 - p))^T dO
          } else {
            ds[u] = pv[u] * (dpe - dl[u]);
          }
        }
        ppk[2 * q] = pack_bf16x2(pv[0], pv[1]);
        ppk[2 * q + 1] = pack_bf16x2(pv[2], pv[3]);
        dpk[2 * q] = pack_bf16x2(ds[0], ds[1]);
        dpk[2 * q + 1] = pack_bf16x2(ds[2], ds[3]);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pb = __builtin_bit_cast(bf16x8, make_uint4(ppk[4 * s], ppk[4 * s + 1], ppk[4 * s + 2], ppk[4 * s + 3]));
        const bf16x8 db = __builtin_bit_cast(bf16x8, make_uint4(dpk[4 * s], dpk[4 * s + 1], dpk[4 * s + 2], dpk[4 * s + 3]));
#pragma unroll
        for (int d = 0; d < NDH; ++d) {
          const int drow = 32 * (NDH * dh + d) + l31;
          dVt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr_frag(GT, drow, 32 * is, s, hi), pb, dVt[d], 0, 0, 0);
          dKt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr_frag(QT, drow, 32 * is, s, hi), db, dKt[d], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();
  spill_transposed<HD, NDH>(dKt, NDH * dh, p.scale, dlds, jt, l31, hi);
  spill_transposed<HD, NDH>(dVt, NDH * dh, 1.0f, dlds + A::RM, jt, l31, hi);
  __syncthreads();
  if (p.cos) store_rows_unrope<HD, A::N64>(dlds, dk_base, p.s[6][2], j0, T, t, p.cos + cs_off(p, b, sq), p.sin + cs_off(p, b, sq), p.cs_t);
  else store_rows<HD, A::N64>(dlds, dk_base, p.s[6][2], j0, T, t);
  store_rows<HD, A::N64>(dlds + A::RM, dv_base, p.s[7][2], j0, T, t);
}

// ---- LDS images filled by LDS-DMA (global_load_lds_dwordx4: the destination is a wave-uniform base + 16 bytes x lane, so an
// image cannot be padded) and read by ds_read_b128 (A operands, lane <-> row) AND ds_read_b64_tr_b16 (operands whose contraction
// index runs over the image's ROWS - what the transposed LDS copies above exist for).  A [64 rows][HD] block: row r at byte
// 2 HD r, its 16-byte chunk x at position x ^ swz(r).  swz is chosen so that both read patterns touch every bank once
// (MI355X_MICROARCH.md LDS table: b128 is served in 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...; the transpose
// read in 32-lane halves, each a 4-row x 64-byte patch): HD 128 (16 chunks, a row = all 64 banks): 4 (r & 3) + ((r >> 2) & 3);
// HD 64 (8 chunks, a row = half the banks, odd rows the other half): 4 ((r >> 1) & 1) + ((r >> 2) & 3).
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int HD>
__device__ __forceinline__ int swz(int r) {
  return HD == 128 ? 4 * (r & 3) + ((r >> 2) & 3) : 4 * ((r >> 1) & 1) + ((r >> 2) & 3);
}
// One LDS-DMA piece: 16 (or 4) bytes per lane from `sbase` (wave-uniform) + voff bytes (per lane) to LDS byte address lds_dst
// (wave-uniform) + 16 (4) lane.  Inline assembly on purpose: hipcc treats the builtin form as an LDS store it cannot tell apart
// from the images' reads and drains vmcnt in front of every ds_read that follows - the overlap this kernel exists for.  The
// kernel counts these loads itself (s_waitcnt vmcnt(0) + s_barrier before a stage is read).
__device__ __forceinline__ void glds16(const void* sbase, unsigned int voff, unsigned int lds_dst) {
  unsigned int keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void* sbase, unsigned int voff, unsigned int lds_dst) {
  unsigned int keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// rows row0 .. row0 + 63 of a [rows][HD] tensor -> image at LDS byte address `img` (rows past the sequence's end repeat its
// last row: finite values that only ever meet zero probabilities); every wave issues its quarter of the 1-KiB pieces, nothing
// passes through registers.  T row_stride < 2^31 elements (checked by the host).
template <int HD>
__device__ __forceinline__ void dma_block(const unsigned short* base, unsigned int row_stride, int row0, int T, unsigned int img, int w, int l) {
  constexpr int CH = HD / 8, RPI = 64 / CH, NI = 64 / RPI / 4;   // lanes per row, rows per piece, pieces per wave
#pragma unroll
  for (int n = 0; n < NI; ++n) {
    const int piece = NI * w + n, r = RPI * piece + l / CH, x = (l % CH) ^ swz<HD>(r);
    const unsigned int row = static_cast<unsigned int>(min(row0 + r, T - 1));
    glds16(base, 2u * (row * row_stride + 8u * x), img + 1024u * piece);
  }
}
// A operand, lane <-> row (32 tile + l31), k-step kk: the 16 bytes at d = 16 kk + 8 hi
template <int HD>
__device__ __forceinline__ bf16x8 img_frag(const unsigned char* img, int tile, int l31, int hi, int kk) {
  const int r = 32 * tile + l31;
  return *reinterpret_cast<const bf16x8*>(img + r * (2 * HD) + 16 * ((2 * kk + hi) ^ swz<HD>(r)));
}
// A operand [d (lane & 31)][k = image rows r0 + 16 s + 4 hi + {0..3}, then + 8] of d block `dblk`: what ld_tr_frag reads from a
// transposed copy, here two transpose reads of the row-major image (a 16-lane group reads a [4 rows][16 columns] patch, 4
// contiguous elements per lane; lane t of the group receives column t of the four rows)
template <int HD>
__device__ __forceinline__ bf16x8 img_tr_frag(const unsigned char* img, int dblk, int r0, int s, int l) {
  const int t = l & 15, g = l >> 4, hi = l >> 5;
  const int x = 4 * dblk + 2 * (g & 1) + ((t & 3) >> 1);
  s16x4 v[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = r0 + 16 * s + 8 * u + 4 * hi + (t >> 2);
    const unsigned char* a = img + r * (2 * HD) + 16 * (x ^ swz<HD>(r)) + 8 * (t & 1);
    v[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) s16x4*>(reinterpret_cast<uintptr_t>(a)));
  }
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 both = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w};
  return __builtin_bit_cast(bf16x8, both);
}

// Top of a streamed block: this wave's LDS-DMA pieces of the block have landed (vmcnt(0): the BUILTIN wait, which hipcc's own
// counter bookkeeping sees - after an inline-assembly wait it still believed the mask words of the block, loaded one iteration
// earlier, to be in flight and put its own vmcnt(0) in front of their first use, i.e. AFTER the next block's pieces had been
// issued: no overlap); after the barrier everybody's have, and everybody is done reading the stage the next pieces go to.  The
// empty statement pins the mask words' wait to this point as well.
__device__ __forceinline__ void stage_ready(const uint32_t (&nword)[2]) {
  __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0), expcnt / lgkmcnt untouched
  asm volatile("s_barrier" ::: "memory");
  asm volatile("" :: "v"(nword[0]), "v"(nword[1]) : "memory");
}

template <int HD>
constexpr int dkdv2_lds() { return 4 * 64 * 2 * HD + 1024; }

// dk / dv, second form: the arithmetic of attn_bwd_dkdv_kernel, instruction for instruction - same tiles, same
// operands, same order, bit-identical results - with another data path.  Q / dO blocks go HBM -> LDS by LDS-DMA into the
// un-padded swizzled images above, TWO stages: block n + 1 is in flight while block n is multiplied (the first form held a block in
// 32 registers per lane on its way to LDS and could not afford to keep them live across the products: every block's load latency
// was exposed), one barrier per block instead of two.  The transposed copies (64 two-byte LDS stores per lane and block, half of
// this kernel's LDS time) are gone: dO^T / Q^T operands are transpose reads of the row-major images.  The log-sum-exp and D of the
// block's rows travel the same way (4-byte pieces).  K / V fragments of the workgroup's own rows come straight from HBM in
// fragment shape (once per workgroup).
template <int HD, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv2_kernel(const AttnBwdParams p) {
  using A = AT<HD>;
  constexpr int NDH = A::ND / 2;
  constexpr int IMG = 64 * 2 * HD;                             // one block image
  extern __shared__ __attribute__((aligned(16))) unsigned char dlds[];
  // stage s: Q image at s 2 IMG, dO image at s 2 IMG + IMG; lse / D of the block's rows at 4 IMG + 512 s (+ 256)
  const int t = threadIdx.x, w = t >> 6, l = t & 63, l31 = l & 31, hi = l >> 5, jt = w >> 1, dh = w & 1;
  int blk, h, b;
  if (!block_coords(p, (p.T + 63) >> 6, false, blk, h, b)) return;
  const int j0 = blk * 64;
  const int j = j0 + 32 * jt + l31;
  const int64_t bh = static_cast<int64_t>(b) * p.H + h;
  const Seq sq = seq_of(p, b);
  const int T = sq.T;
  unsigned short* dk_base = p.dk + base_off(p, 6, b, h, sq);
  unsigned short* dv_base = p.dv + base_off(p, 7, b, h, sq);
  const unsigned long long need = need_mask<2>(p, b, j0 >> 5, false, l);
  if (need == 0ull || T <= 0) {
    store_rows<HD, A::N64>(nullptr, dk_base, p.s[6][2], j0, T, t);
    store_rows<HD, A::N64>(nullptr, dv_base, p.s[7][2], j0, T, t);
    return;
  }
  const unsigned short* qbase = p.q + base_off(p, 0, b, h, sq);
  const unsigned short* gbase = p.d_o + base_off(p, 4, b, h, sq);
  const int Tp = 32 * p.W;
  const int nI = (p.T + 63) >> 6;
  unsigned int blocks = 0u;                                    // bit ib: query block ib has a live tile (T <= 2048: 32 blocks)
  for (int ib = 0; ib < nI; ++ib) blocks |= (((need >> (2 * ib)) & 3ull) != 0ull ? 1u : 0u) << ib;

  uint32_t nword[2];
  const unsigned int lds0 = static_cast<unsigned int>(reinterpret_cast<uintptr_t>(dlds));   // LDS byte address of the carve-out
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const unsigned int sq_row = static_cast<unsigned int>(p.s[0][2]), sg_row = static_cast<unsigned int>(p.s[4][2]);
  const float* lse_row = p.lse + bh * p.T;
  const float* delta_row = p.delta + bh * p.T;
  auto issue = [&](int ib, int stage) {
    const unsigned int qi = lds0 + stage * (2 * IMG);
    dma_block<HD>(qbase, sq_row, 64 * ib, T, qi, wu, l);
    dma_block<HD>(gbase, sg_row, 64 * ib, T, qi + IMG, wu, l);
    if (wu == 0) {
      const unsigned int row = static_cast<unsigned int>(min(64 * ib + l, T - 1));
      glds4(lse_row, 4u * row, lds0 + 4 * IMG + 512 * stage);
      glds4(delta_row, 4u * row, lds0 + 4 * IMG + 512 * stage + 256);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
      nword[c] = (j < Tp && 2 * ib + c < p.W) ? p.bits_cols[(static_cast<int64_t>(b) * Tp + j) * p.W + 2 * ib + c] : 0u;
  };
  int cur = __builtin_ctz(blocks);
  blocks &= blocks - 1u;
  issue(cur, 0);

  bf16x8 Kb[A::KK], Vb[A::KK];
  {
    const bool ok = j < T;
    const unsigned short* kr = p.k + base_off(p, 1, b, h, sq) + static_cast<int64_t>(j) * p.s[1][2] + 8 * hi;
    const unsigned short* vr = p.v + base_off(p, 2, b, h, sq) + static_cast<int64_t>(j) * p.s[2][2] + 8 * hi;
#pragma unroll
    for (int kk = 0; kk < A::KK; ++kk) {
      Kb[kk] = __builtin_bit_cast(bf16x8, ok ? ld16(kr + 16 * kk) : make_uint4(0u, 0u, 0u, 0u));
      Vb[kk] = __builtin_bit_cast(bf16x8, ok ? ld16(vr + 16 * kk) : make_uint4(0u, 0u, 0u, 0u));
    }
  }
  const float c1 = p.scale * kLog2e;
  f32x16 dVt[NDH], dKt[NDH];
#pragma unroll
  for (int d = 0; d < NDH; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dVt[d][r] = 0.f; dKt[d][r] = 0.f; }

  const AttnDrop drop = attn_drop(p);
  (void)drop;
  int stage = 0;
  while (true) {
    stage_ready(nword);
    const uint32_t word[2] = {nword[0], nword[1]};
    const int ib = cur;
    (void)ib;
    int nxt = -1;
    if (blocks) {
      nxt = __builtin_ctz(blocks);
      blocks &= blocks - 1u;
      issue(nxt, stage ^ 1);
    }
    const unsigned char* Qi = dlds + stage * (2 * IMG);
    const unsigned char* Gi = Qi + IMG;
    const float* nl_s = reinterpret_cast<const float*>(dlds + 4 * IMG + 512 * stage);
    const float* dl_s = nl_s + 64;
#pragma unroll
    for (int is = 0; is < 2; ++is) {
      if (__builtin_amdgcn_ballot_w64(word[is] != 0u) == 0ull) continue;
      f32x16 S, dP;
#pragma unroll
      for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < A::KK; ++kk) {
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag<HD>(Qi, is, l31, hi, kk), Kb[kk], S, 0, 0, 0);
        dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag<HD>(Gi, is, l31, hi, kk), Vb[kk], dP, 0, 0, 0);
      }
      unsigned int ppk[8], dpk[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 ls4 = *reinterpret_cast<const float4*>(nl_s + 32 * is + 8 * q + 4 * hi);
        const float4 dl4 = *reinterpret_cast<const float4*>(dl_s + 32 * is + 8 * q + 4 * hi);
        const float nl[4] = {-ls4.x * kLog2e, -ls4.y * kLog2e, -ls4.z * kLog2e, -ls4.w * kLog2e}, dl[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
        float pv[4], ds[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int il = 8 * q + 4 * hi + u;
          pv[u] = ((word[is] >> il) & 1u) ? __builtin_amdgcn_exp2f(fmaf(S[4 * q + u], c1, nl[u])) : 0.f;
          float dpe = dP[4 * q + u];
          if constexpr (DROP) {
            const unsigned int c = (static_cast<unsigned int>(bh) * p.T + 64 * ib + 32 * is + il) * p.T + j;
            const unsigned int hw = drop_pair(drop, c & ~1u);
            const bool keep = ((c & 1u) ? hw >> 16 : hw & 0xffffu) >= drop.thresh;
            dpe = keep ? dpe * drop.ks : 0.f;
            ds[u] = pv[u] * (dpe - dl[u]);
            pv[u] = keep ? pv[u] * drop.ks : 0.f;                // what multiplies dO in dV = (P o M / (1 - p))^T dO
          } else {
            ds[u] = pv[u] * (dpe - dl[u]);
          }
        }
        ppk[2 * q] = pack_bf16x2(pv[0], pv[1]);
        ppk[2 * q + 1] = pack_bf16x2(pv[2], pv[3]);
        dpk[2 * q] = pack_bf16x2(ds[0], ds[1]);
        dpk[2 * q + 1] = pack_bf16x2(ds[2], ds[3]);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pb = __builtin_bit_cast(bf16x8, make_uint4(ppk[4 * s], ppk[4 * s + 1], ppk[4 * s + 2], ppk[4 * s + 3]));
        const bf16x8 db = __builtin_bit_cast(bf16x8, make_uint4(dpk[4 * s], dpk[4 * s + 1], dpk[4 * s + 2], dpk[4 * s + 3]));
#pragma unroll
        for (int d = 0; d < NDH; ++d) {
          dVt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_tr_frag<HD>(Gi, NDH * dh + d, 32 * is, s, l), pb, dVt[d], 0, 0, 0);
          dKt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_tr_frag<HD>(Qi, NDH * dh + d, 32 * is, s, l), db, dKt[d], 0, 0, 0);
        }
      }
    }
    if (nxt < 0) break;
    cur = nxt;
    stage ^= 1;
  }
  __syncthreads();
  spill_transposed<HD, NDH>(dKt, NDH * dh, p.scale, dlds, jt, l31, hi);
  spill_transposed<HD, NDH>(dVt, NDH * dh, 1.0f, dlds + A::RM, jt, l31, hi);
  __syncthreads();
  if (p.cos) store_rows_unrope<HD, A::N64>(dlds, dk_base, p.s[6][2], j0, T, t, p.cos + cs_off(p, b, sq), p.sin + cs_off(p, b, sq), p.cs_t);
  else store_rows<HD, A::N64>(dlds, dk_base, p.s[6][2], j0, T, t);
  store_rows<HD, A::N64>(dlds + A::RM, dv_base, p.s[7][2], j0, T, t);
}

// the wave's 32 rows (lane <-> row) as B operands straight from HBM in fragment shape: k-step kk = the 16 bytes at d = 16 kk + 8 hi
template <int HD>
__device__ __forceinline__ void rows_frags_global(const unsigned short* base, int64_t row_stride, int row, int T, int hi, uint4 (&f)[AT<HD>::KK]) {
  const bool ok = row < T;
  const unsigned short* src = base + static_cast<int64_t>(row) * row_stride + 8 * hi;
#pragma unroll
  for (int kk = 0; kk < AT<HD>::KK; ++kk) f[kk] = ok ? ld16(src + 16 * kk) : make_uint4(0u, 0u, 0u, 0u);
}

template <int HD>
constexpr int stream2_lds() { return 4 * 64 * 2 * HD; }      // two stages of (K image, V image)

// dq, second form: the arithmetic of attn_bwd_dq_kernel with the data path of attn_bwd_dkdv2_kernel (K / V blocks by LDS-DMA into
// two stages of swizzled images, K^T operands by transpose reads of the K image, one barrier per block).  Q / dO fragments and
// D = rowsum(dO o O) come from fragment-shaped loads (a lane pair l, l ^ 32 holds one row: even / odd 16-byte chunks); D is summed
// in the first form's order (chunk dot products, then the butterfly over chunk pairs) so that both forms write the same bits.
template <int HD, bool DROP>
__global__ __launch_bounds__(256, DROP ? 2 : AT<HD>::OCC) void attn_bwd_dq2_kernel(const AttnBwdParams p) {
  using A = AT<HD>;
  constexpr int IMG = 64 * 2 * HD;
  extern __shared__ __attribute__((aligned(16))) unsigned char dlds[];
  const int t = threadIdx.x, w = t >> 6, l = t & 63, l31 = l & 31, hi = l >> 5;
  int blk, h, b;
  if (!block_coords(p, (p.T + 127) >> 7, true, blk, h, b)) return;
  const int i0 = blk * 128;
  const int i = i0 + 32 * w + l31;
  const int64_t bh = static_cast<int64_t>(b) * p.H + h;
  const Seq sq = seq_of(p, b);
  const int T = sq.T;
  unsigned short* dq_base = p.dq + base_off(p, 5, b, h, sq);
  const unsigned long long need = need_mask<4>(p, b, i0 >> 5, true, l);
  if (need == 0ull || T <= 0) {
    store_rows<HD, A::N128>(nullptr, dq_base, p.s[5][2], i0, T, t);
    return;
  }
  const unsigned short* kbase = p.k + base_off(p, 1, b, h, sq);
  const unsigned short* vbase = p.v + base_off(p, 2, b, h, sq);
  const int Tp = 32 * p.W;
  const int nJ = (p.T + 63) >> 6;
  unsigned int blocks = 0u;
  for (int jb = 0; jb < nJ; ++jb) blocks |= (((need >> (2 * jb)) & 3ull) != 0ull ? 1u : 0u) << jb;

  uint32_t nword[2];
  const unsigned int lds0 = static_cast<unsigned int>(reinterpret_cast<uintptr_t>(dlds));
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const unsigned int sk_row = static_cast<unsigned int>(p.s[1][2]), sv_row = static_cast<unsigned int>(p.s[2][2]);
  auto issue = [&](int jb, int stage) {
    const unsigned int ki = lds0 + stage * (2 * IMG);
    dma_block<HD>(kbase, sk_row, 64 * jb, T, ki, wu, l);
    dma_block<HD>(vbase, sv_row, 64 * jb, T, ki + IMG, wu, l);
#pragma unroll
    for (int c = 0; c < 2; ++c)
      nword[c] = (i < Tp && 2 * jb + c < p.W) ? p.bits_rows[(static_cast<int64_t>(b) * Tp + i) * p.W + 2 * jb + c] : 0u;
  };
  int cur = __builtin_ctz(blocks);
  blocks &= blocks - 1u;
  issue(cur, 0);

  bf16x8 Qb[A::KK], Gb[A::KK];
  float Dl;
  {
    uint4 qv[A::KK], gv[A::KK], ov[A::KK];
    rows_frags_global<HD>(p.q + base_off(p, 0, b, h, sq), p.s[0][2], i, T, hi, qv);
    rows_frags_global<HD>(p.d_o + base_off(p, 4, b, h, sq), p.s[4][2], i, T, hi, gv);
    rows_frags_global<HD>(p.o + base_off(p, 3, b, h, sq), p.s[3][2], i, T, hi, ov);
    float pr[A::KK];
#pragma unroll
    for (int kk = 0; kk < A::KK; ++kk) {                         // chunk 2 kk + hi here, chunk 2 kk + (1 - hi) in lane l ^ 32
      const float d = dot8(gv[kk], ov[kk]);
      pr[kk] = d + __shfl_xor(d, 32, 64);
      Qb[kk] = __builtin_bit_cast(bf16x8, qv[kk]);
      Gb[kk] = __builtin_bit_cast(bf16x8, gv[kk]);
    }
#pragma unroll
    for (int off = 1; off < A::KK; off <<= 1)
#pragma unroll
      for (int kk = 0; kk < A::KK; kk += 2 * off) pr[kk] += pr[kk + off];
    Dl = pr[0];
    if (hi == 0 && i < T) p.delta[bh * p.T + i] = Dl;
  }
  const float nl = i < T ? -p.lse[bh * p.T + i] * kLog2e : 0.f;
  const float c1 = p.scale * kLog2e;

  f32x16 acc[A::ND];
#pragma unroll
  for (int d = 0; d < A::ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  const AttnDrop drop = attn_drop(p);
  const unsigned int cbase = (static_cast<unsigned int>(bh) * p.T + i) * p.T;      // element index of (row i, column 0)
  (void)drop; (void)cbase;
  int stage = 0;
  while (true) {
    stage_ready(nword);
    const uint32_t word[2] = {nword[0], nword[1]};
    const int jb = cur;
    (void)jb;
    int nxt = -1;
    if (blocks) {
      nxt = __builtin_ctz(blocks);
      blocks &= blocks - 1u;
      issue(nxt, stage ^ 1);
    }
    const unsigned char* Ki = dlds + stage * (2 * IMG);
    const unsigned char* Vi = Ki + IMG;
#pragma unroll 1
    for (int js = 0; js < 2; ++js) {
      if (__builtin_amdgcn_ballot_w64(word[js] != 0u) == 0ull) continue;
      f32x16 St, Pt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { St[r] = 0.f; Pt[r] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < A::KK; ++kk) {
        St = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag<HD>(Ki, js, l31, hi, kk), Qb[kk], St, 0, 0, 0);
        Pt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag<HD>(Vi, js, l31, hi, kk), Gb[kk], Pt, 0, 0, 0);
      }
      unsigned int pk[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float ds[2];
        const int jl0 = (r & 3) + 8 * (r >> 2) + 4 * hi;       // even: (jl0, jl0 + 1) share a hash
        unsigned int hw = 0xffffffffu;
        if constexpr (DROP) hw = drop_pair(drop, cbase + 64 * jb + 32 * js + jl0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int jl = jl0 + u;
          const float pv = ((word[js] >> jl) & 1u) ? __builtin_amdgcn_exp2f(fmaf(St[r + u], c1, nl)) : 0.f;
          float dpe = Pt[r + u];
          if constexpr (DROP) dpe = ((u ? hw >> 16 : hw & 0xffffu) >= drop.thresh) ? dpe * drop.ks : 0.f;
          ds[u] = pv * (dpe - Dl);
        }
        pk[r >> 1] = pack_bf16x2(ds[0], ds[1]);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 dsb = __builtin_bit_cast(bf16x8, make_uint4(pk[4 * s], pk[4 * s + 1], pk[4 * s + 2], pk[4 * s + 3]));
#pragma unroll
        for (int d = 0; d < A::ND; ++d)
          acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_tr_frag<HD>(Ki, d, 32 * js, s, l), dsb, acc[d], 0, 0, 0);
      }
    }
    if (nxt < 0) break;
    cur = nxt;
    stage ^= 1;
  }
  __syncthreads();
  spill_transposed<HD, A::ND>(acc, 0, p.scale, dlds, w, l31, hi);
  __syncthreads();
  if (p.cos) store_rows_unrope<HD, A::N128>(dlds, dq_base, p.s[5][2], i0, T, t, p.cos + cs_off(p, b, sq), p.sin + cs_off(p, b, sq), p.cs_t);
  else store_rows<HD, A::N128>(dlds, dq_base, p.s[5][2], i0, T, t);
}

// Forward, second form: attn_fwd_kernel's arithmetic on the same data path (K image read row-wise, V image by transpose reads).
template <int HD, bool DROP>
__global__ __launch_bounds__(256, (DROP && HD == 64) ? 2 : AT<HD>::OCC) void attn_fwd2_kernel(const AttnBwdParams p) {
  using A = AT<HD>;
  constexpr int IMG = 64 * 2 * HD;
  extern __shared__ __attribute__((aligned(16))) unsigned char dlds[];
  const int t = threadIdx.x, w = t >> 6, l = t & 63, l31 = l & 31, hi = l >> 5;
  int blk, h, b;
  if (!block_coords(p, (p.T + 127) >> 7, true, blk, h, b)) return;
  const int i0 = blk * 128;
  const int i = i0 + 32 * w + l31;
  const int64_t bh = static_cast<int64_t>(b) * p.H + h;
  const Seq sq = seq_of(p, b);
  const int T = sq.T;
  unsigned short* o_base = p.dq + base_off(p, 5, b, h, sq);               // the output travels in the dq slot
  float* lse_out = p.delta;                                                // and the log-sum-exp in the delta slot
  const unsigned long long need = need_mask<4>(p, b, i0 >> 5, true, l);
  if (need == 0ull || T <= 0) {                                          // rows without a live key: zero output (as torch returns)
    store_rows<HD, A::N128>(nullptr, o_base, p.s[5][2], i0, T, t);
    if (t < 128 && i0 + t < T) lse_out[bh * p.T + i0 + t] = 0.f;
    return;
  }
  const unsigned short* kbase = p.k + base_off(p, 1, b, h, sq);
  const unsigned short* vbase = p.v + base_off(p, 2, b, h, sq);
  const int Tp = 32 * p.W;
  const int nJ = (p.T + 63) >> 6;
  unsigned int blocks = 0u;
  for (int jb = 0; jb < nJ; ++jb) blocks |= (((need >> (2 * jb)) & 3ull) != 0ull ? 1u : 0u) << jb;

  uint32_t nword[2];
  const unsigned int lds0 = static_cast<unsigned int>(reinterpret_cast<uintptr_t>(dlds));
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const unsigned int sk_row = static_cast<unsigned int>(p.s[1][2]), sv_row = static_cast<unsigned int>(p.s[2][2]);
  auto issue = [&](int jb, int stage) {
    const unsigned int ki = lds0 + stage * (2 * IMG);
    dma_block<HD>(kbase, sk_row, 64 * jb, T, ki, wu, l);
    dma_block<HD>(vbase, sv_row, 64 * jb, T, ki + IMG, wu, l);
#pragma unroll
    for (int c = 0; c < 2; ++c)
      nword[c] = (i < Tp && 2 * jb + c < p.W) ? p.bits_rows[(static_cast<int64_t>(b) * Tp + i) * p.W + 2 * jb + c] : 0u;
  };
  int cur = __builtin_ctz(blocks);
  blocks &= blocks - 1u;
  issue(cur, 0);

  bf16x8 Qb[A::KK];
  {
    uint4 qv[A::KK];
    rows_frags_global<HD>(p.q + base_off(p, 0, b, h, sq), p.s[0][2], i, T, hi, qv);
#pragma unroll
    for (int kk = 0; kk < A::KK; ++kk) Qb[kk] = __builtin_bit_cast(bf16x8, qv[kk]);
  }
  const float c1 = p.scale * kLog2e;
  f32x16 acc[A::ND];
#pragma unroll
  for (int d = 0; d < A::ND; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;                              // running maximum (log2 domain) of the row; this lane's share of the sum

  const AttnDrop drop = attn_drop(p);
  const unsigned int cbase = (static_cast<unsigned int>(bh) * p.T + i) * p.T;
  (void)drop; (void)cbase;
  int stage = 0;
  while (true) {
    stage_ready(nword);
    const uint32_t word[2] = {nword[0], nword[1]};
    const int jb = cur;
    (void)jb;
    int nxt = -1;
    if (blocks) {
      nxt = __builtin_ctz(blocks);
      blocks &= blocks - 1u;
      issue(nxt, stage ^ 1);
    }
    const unsigned char* Ki = dlds + stage * (2 * IMG);
    const unsigned char* Vi = Ki + IMG;
#pragma unroll
    for (int js = 0; js < 2; ++js) {
      if (__builtin_amdgcn_ballot_w64(word[js] != 0u) == 0ull) continue;
      f32x16 St;
#pragma unroll
      for (int r = 0; r < 16; ++r) St[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < A::KK; ++kk)
        St = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_frag<HD>(Ki, js, l31, hi, kk), Qb[kk], St, 0, 0, 0);
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jl = (r & 3) + 8 * (r >> 2) + 4 * hi;
        St[r] = ((word[js] >> jl) & 1u) ? St[r] * c1 : -INFINITY;
        mx = fmaxf(mx, St[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m, mx);
      const float base = mn == -INFINITY ? 0.f : mn;             // a row with nothing live so far: every p below is exp2(-inf) = 0
      const float alpha = __builtin_amdgcn_exp2f(m - base);      // m = -inf: 0
      m = mn;
      float ps = 0.f;
      unsigned int pk[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(St[r] - base), p1 = __builtin_amdgcn_exp2f(St[r + 1] - base);
        ps += p0 + p1;                                           // the softmax denominator counts dropped elements too
        if constexpr (DROP) {
          const unsigned int hw = drop_pair(drop, cbase + 64 * jb + 32 * js + (r & 3) + 8 * (r >> 2) + 4 * hi);
          pk[r >> 1] = pack_bf16x2((hw & 0xffffu) >= drop.thresh ? p0 * drop.ks : 0.f, (hw >> 16) >= drop.thresh ? p1 * drop.ks : 0.f);
        } else {
          pk[r >> 1] = pack_bf16x2(p0, p1);
        }
      }
      lsum = fmaf(lsum, alpha, ps);
#pragma unroll
      for (int d = 0; d < A::ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] *= alpha;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 pb = __builtin_bit_cast(bf16x8, make_uint4(pk[4 * s], pk[4 * s + 1], pk[4 * s + 2], pk[4 * s + 3]));
#pragma unroll
        for (int d = 0; d < A::ND; ++d)
          acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_tr_frag<HD>(Vi, d, 32 * js, s, l), pb, acc[d], 0, 0, 0);
      }
    }
    if (nxt < 0) break;
    cur = nxt;
    stage ^= 1;
  }
  const float ltot = lsum + __shfl_xor(lsum, 32, 64);
  const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
  if (hi == 0 && i < T) lse_out[bh * p.T + i] = ltot > 0.f ? (m + __builtin_amdgcn_logf(ltot)) * kLn2 : 0.f;
  __syncthreads();
  spill_transposed<HD, A::ND>(acc, 0, inv, dlds, w, l31, hi);
  __syncthreads();
  store_rows<HD, A::N128>(dlds, o_base, p.s[5][2], i0, T, t);
}

// The live-tile bytes are cleared by a KERNEL in front of the mask kernels, not by hipMemsetAsync: captured into a hipGraph, the
// memset node of this small odd-sized array did not clear it on replay (tools/live_tiles_graph_check.py: bytes of the previous
// batch stayed set) - harmless for the first forms of the kernels above beyond the work spent on dead tiles (their mask words are
// zero), a fault for the second forms once a stale byte belonged to a sequence that is empty in the current batch.
__global__ __launch_bounds__(256) void attn_clear_bytes_kernel(unsigned char* __restrict__ p, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) p[i] = 0;
}

// mask [B, 1, T, T] bytes (non-zero = attend; NULL = all) and / or causal -> row words, column words, live 32 x 32 tiles
__global__ __launch_bounds__(256) void attn_mask_bits_kernel(const unsigned char* __restrict__ mask, int B, int T, int W, int64_t sb,
                                                             int64_t si, int causal, uint32_t* __restrict__ rows,
                                                             uint32_t* __restrict__ cols, unsigned char* __restrict__ live) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int Tp = 32 * W;
  if (idx >= static_cast<int64_t>(B) * Tp * W) return;
  const int w = static_cast<int>(idx % W), r = static_cast<int>((idx / W) % Tp), b = static_cast<int>(idx / (static_cast<int64_t>(W) * Tp));
  uint32_t wr = 0u, wc = 0u;
  if (r < T) {
    for (int c = 0; c < 32; ++c) {
      const int x = 32 * w + c;
      if (x >= T) break;
      bool mr = mask ? mask[b * sb + r * si + x] != 0 : true;
      bool mc = mask ? mask[b * sb + x * si + r] != 0 : true;
      if (causal) { mr = mr && x <= r; mc = mc && r <= x; }
      wr |= static_cast<uint32_t>(mr) << c;
      wc |= static_cast<uint32_t>(mc) << c;
    }
  }
  rows[idx] = wr;
  cols[idx] = wc;
  if (wr) live[(static_cast<int64_t>(b) * W + (r >> 5)) * W + w] = 1;
}

// the same words for PACKED sequences: sequence b = token rows cu[b] .. cu[b + 1] - 1, local row / column index = order inside the
// sequence; key_live [n_tokens] bytes (NULL = all): a token that may be attended (the padded layout's attention_mask at its
// position; a token that is only a QUERY - the padding position in front of a left-padded sequence, whose row predicts the first
// real token - carries 0); element (i, j) is live when j is a live key and (causal: j <= i)
__global__ __launch_bounds__(256) void attn_mask_bits_packed_kernel(const unsigned char* __restrict__ key_live,
                                                                    const int* __restrict__ cu, int B, int W, int causal,
                                                                    uint32_t* __restrict__ rows, uint32_t* __restrict__ cols,
                                                                    unsigned char* __restrict__ live) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int Tp = 32 * W;
  if (idx >= static_cast<int64_t>(B) * Tp * W) return;
  const int w = static_cast<int>(idx % W), r = static_cast<int>((idx / W) % Tp), b = static_cast<int>(idx / (static_cast<int64_t>(W) * Tp));
  const int r0 = cu[b], T = cu[b + 1] - r0;
  uint32_t wr = 0u, wc = 0u;
  if (r < T) {
    const bool kr = key_live ? key_live[r0 + r] != 0 : true;
    for (int c = 0; c < 32; ++c) {
      const int x = 32 * w + c;
      if (x >= T) break;
      const bool kx = key_live ? key_live[r0 + x] != 0 : true;
      const bool mr = kx && (!causal || x <= r);             // row r attends column x
      const bool mc = kr && (!causal || r <= x);             // row x attends column r
      wr |= static_cast<uint32_t>(mr) << c;
      wc |= static_cast<uint32_t>(mc) << c;
    }
  }
  rows[idx] = wr;
  cols[idx] = wc;
  if (wr) live[(static_cast<int64_t>(b) * W + (r >> 5)) * W + w] = 1;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline void set_dropout(AttnBwdParams& p, float dropout_p, const void* seed, uint32_t salt) {
  const bool on = dropout_p > 0.f;
  p.seed = on ? static_cast<const unsigned long long*>(seed) : nullptr;
  p.salt = salt;
  p.thresh = on ? static_cast<unsigned int>(dropout_p * 65536.0f + 0.5f) : 0u;
  p.keep_scale = on ? 1.0f / (1.0f - dropout_p) : 1.0f;
}

}  // namespace
}  // namespace dalm

using namespace dalm;

extern "C" int dalm_attn_mask_bits(const void* mask, int64_t B, int64_t T, int64_t mask_stride_b, int64_t mask_stride_row, int causal,
                                   uint32_t* bits_rows, uint32_t* bits_cols, uint8_t* live, dalm_stream_t stream) {
  DALM_REQUIRE(bits_rows && bits_cols && live, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(B > 0 && T > 0 && T <= 32768 && B <= 65535, DALM_E_SHAPE, "need 0 < T <= 32768 and 0 < B <= 65535");
  const int64_t W = (T + 31) / 32, total = B * 32 * W * W;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(attn_clear_bytes_kernel, dim3(static_cast<unsigned>((B * W * W + 255) / 256)), dim3(256), 0, s, live, B * W * W);
  hipLaunchKernelGGL(attn_mask_bits_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s,
                     static_cast<const unsigned char*>(mask), static_cast<int>(B), static_cast<int>(T), static_cast<int>(W),
                     mask_stride_b, mask_stride_row, causal, bits_rows, bits_cols, live);
  return check_launch(__func__);
}

extern "C" int dalm_attn_mask_bits_packed(const uint8_t* key_live, const int32_t* cu_seqlens, int64_t B, int64_t T, int causal,
                                          uint32_t* bits_rows, uint32_t* bits_cols, uint8_t* live, dalm_stream_t stream) {
  DALM_REQUIRE(cu_seqlens && bits_rows && bits_cols && live, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(B > 0 && T > 0 && T <= 2048 && B <= 65535, DALM_E_SHAPE, "need 0 < T <= 2048 and 0 < B <= 65535");
  const int64_t W = (T + 31) / 32, total = B * 32 * W * W;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(attn_clear_bytes_kernel, dim3(static_cast<unsigned>((B * W * W + 255) / 256)), dim3(256), 0, s, live, B * W * W);
  hipLaunchKernelGGL(attn_mask_bits_packed_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s, key_live,
                     cu_seqlens, static_cast<int>(B), static_cast<int>(W), causal, bits_rows, bits_cols, live);
  return check_launch(__func__);
}

static int dalm_attn_bwd_any(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                         const uint32_t* bits_rows, const uint32_t* bits_cols, const uint8_t* live, const int32_t* cu, int64_t B, int64_t H,
                         int64_t T, int64_t hd, float scale, const int64_t* strides, const void* cos, const void* sin,
                         int64_t cs_stride_b, int64_t cs_stride_t, float dropout_p, const void* seed, uint32_t salt, void* dq,
                         void* dk, void* dv, float* delta, dalm_stream_t stream) {
  DALM_REQUIRE(q && k && v && o && d_o && lse && bits_rows && bits_cols && live && strides && dq && dk && dv && delta, DALM_E_NULL,
               "null pointer argument");
  DALM_REQUIRE(hd == 128 || hd == 64, DALM_E_SHAPE, "head width must be 64 or 128");
  DALM_REQUIRE(B > 0 && H > 0 && T > 0 && T <= 2048 && B * H <= (1ll << 24), DALM_E_SHAPE, "need 0 < T <= 2048 and B H <= 2^24");
  const void* ptrs[8] = {q, k, v, o, d_o, dq, dk, dv};
  for (int i = 0; i < 8; ++i) {
    DALM_REQUIRE(al16(ptrs[i]), DALM_E_ALIGN, "tensors must be 16-byte aligned");
    for (int a = 0; a < 3; ++a)
      DALM_REQUIRE(strides[3 * i + a] >= 0 && strides[3 * i + a] % 8 == 0, DALM_E_ALIGN, "strides must be non-negative multiples of 8 elements");
  }
  for (int i : {0, 1, 2, 4})      // the streamed tensors' rows are addressed with 32-bit byte offsets from the sequence's first row
    DALM_REQUIRE(T * strides[3 * i + 2] < (1ll << 30), DALM_E_SHAPE, "T x row stride must stay below 2^30 elements");
  DALM_REQUIRE((cos == nullptr) == (sin == nullptr), DALM_E_NULL, "cos and sin come together");
  DALM_REQUIRE(!cos || (al16(cos) && al16(sin) && cs_stride_b >= 0 && cs_stride_b % 8 == 0 && cs_stride_t >= hd && cs_stride_t % 8 == 0),
               DALM_E_ALIGN, "cos / sin: 16-byte aligned rows of hd elements");
  DALM_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || (seed && T % 2 == 0)), DALM_E_SHAPE,
               "dropout needs 0 <= p < 1, a device seed word and an even T");
  AttnBwdParams p;
  set_dropout(p, dropout_p, seed, salt);
  p.cu = cu;
  p.cos = static_cast<const unsigned short*>(cos); p.sin = static_cast<const unsigned short*>(sin);
  p.cs_b = cs_stride_b; p.cs_t = cs_stride_t;
  p.q = static_cast<const unsigned short*>(q); p.k = static_cast<const unsigned short*>(k);
  p.v = static_cast<const unsigned short*>(v); p.o = static_cast<const unsigned short*>(o);
  p.d_o = static_cast<const unsigned short*>(d_o); p.lse = lse;
  p.bits_rows = bits_rows; p.bits_cols = bits_cols; p.live = live;
  p.dq = static_cast<unsigned short*>(dq); p.dk = static_cast<unsigned short*>(dk); p.dv = static_cast<unsigned short*>(dv);
  p.delta = delta;
  p.B = static_cast<int>(B); p.H = static_cast<int>(H); p.T = static_cast<int>(T); p.W = static_cast<int>((T + 31) / 32);
  p.scale = scale;
  for (int i = 0; i < 8; ++i)
    for (int a = 0; a < 3; ++a) p.s[i][a] = strides[3 * i + a];
  static bool lds_set = false;
  static bool first_form = false;                              // DALM_ATTN_DKDV=1: the register-staged dk / dv kernel (A/B runs)
  if (!lds_set) {
    for (const void* fn : {reinterpret_cast<const void*>(attn_bwd_dkdv_kernel<128, false>),
                           reinterpret_cast<const void*>(attn_bwd_dkdv_kernel<128, true>)})
      if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, dkdv_lds<128>()); e != hipSuccess)
        return fail(static_cast<int>(e), __func__, "could not raise the dynamic LDS limit of the dk / dv kernel");
    for (const void* fn : {reinterpret_cast<const void*>(attn_bwd_dkdv2_kernel<128, false>),
                           reinterpret_cast<const void*>(attn_bwd_dkdv2_kernel<128, true>)})
      if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, dkdv2_lds<128>()); e != hipSuccess)
        return fail(static_cast<int>(e), __func__, "could not raise the dynamic LDS limit of the dk / dv kernel");
    const char* env = getenv("DALM_ATTN_DKDV");
    first_form = env && env[0] == '1';
    lds_set = true;
  }
  const int64_t pairs8 = (B * H + 7) / 8 * 8;
  const dim3 grid_dq(static_cast<unsigned>(pairs8 * ((T + 127) / 128))), grid_dkdv(static_cast<unsigned>(pairs8 * ((T + 63) / 64)));
  hipStream_t s = as_stream(stream);
#define DALM_ATTN_BWD(HD, DROP)                                                                         \
  do {                                                                                                 \
    if (first_form) hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, DROP>), grid_dq, dim3(256), 0, s, p);  \
    else hipLaunchKernelGGL((attn_bwd_dq2_kernel<HD, DROP>), grid_dq, dim3(256), stream2_lds<HD>(), s, p);  \
    if (first_form || (HD == 128 && DROP)) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<HD, DROP>), grid_dkdv, dim3(256), dkdv_lds<HD>(), s, p);  \
    else hipLaunchKernelGGL((attn_bwd_dkdv2_kernel<HD, DROP>), grid_dkdv, dim3(256), dkdv2_lds<HD>(), s, p);  \
  } while (0)
  if (hd == 128) { if (p.seed) DALM_ATTN_BWD(128, true); else DALM_ATTN_BWD(128, false); }
  else { if (p.seed) DALM_ATTN_BWD(64, true); else DALM_ATTN_BWD(64, false); }
#undef DALM_ATTN_BWD
  return check_launch(__func__);
}

extern "C" int dalm_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                             const uint32_t* bits_rows, const uint32_t* bits_cols, const uint8_t* live, int64_t B, int64_t H,
                             int64_t T, int64_t hd, float scale, const int64_t* strides, const void* cos, const void* sin,
                             int64_t cs_stride_b, int64_t cs_stride_t, float dropout_p, const void* seed, uint32_t salt, void* dq,
                             void* dk, void* dv, float* delta, dalm_stream_t stream) {
  return dalm_attn_bwd_any(q, k, v, o, d_o, lse, bits_rows, bits_cols, live, nullptr, B, H, T, hd, scale, strides, cos, sin,
                           cs_stride_b, cs_stride_t, dropout_p, seed, salt, dq, dk, dv, delta, stream);
}

extern "C" int dalm_attn_bwd_packed(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                                    const uint32_t* bits_rows, const uint32_t* bits_cols, const uint8_t* live,
                                    const int32_t* cu_seqlens, int64_t B, int64_t H, int64_t T, int64_t hd, float scale,
                                    const int64_t* strides, const void* cos, const void* sin, int64_t cs_stride_t, float dropout_p,
                                    const void* seed, uint32_t salt, void* dq, void* dk, void* dv, float* delta,
                                    dalm_stream_t stream) {
  DALM_REQUIRE(cu_seqlens, DALM_E_NULL, "null pointer argument");
  return dalm_attn_bwd_any(q, k, v, o, d_o, lse, bits_rows, bits_cols, live, cu_seqlens, B, H, T, hd, scale, strides, cos, sin, 0,
                           cs_stride_t, dropout_p, seed, salt, dq, dk, dv, delta, stream);
}

static int dalm_attn_fwd_any(const void* q, const void* k, const void* v, const uint32_t* bits_rows, const uint8_t* live,
                            const int32_t* cu, int64_t B, int64_t H, int64_t T, int64_t hd, float scale, const int64_t* strides,
                            float dropout_p, const void* seed, uint32_t salt, void* o, float* lse, dalm_stream_t stream) {
  DALM_REQUIRE(q && k && v && bits_rows && live && strides && o && lse, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(hd == 128 || hd == 64, DALM_E_SHAPE, "head width must be 64 or 128");
  DALM_REQUIRE(B > 0 && H > 0 && T > 0 && T <= 2048 && B * H <= (1ll << 24), DALM_E_SHAPE, "need 0 < T <= 2048 and B H <= 2^24");
  const void* ptrs[4] = {q, k, v, o};
  for (int i = 0; i < 4; ++i) {
    DALM_REQUIRE(al16(ptrs[i]), DALM_E_ALIGN, "tensors must be 16-byte aligned");
    for (int a = 0; a < 3; ++a)
      DALM_REQUIRE(strides[3 * i + a] >= 0 && strides[3 * i + a] % 8 == 0, DALM_E_ALIGN, "strides must be non-negative multiples of 8 elements");
  }
  DALM_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || (seed && T % 2 == 0)), DALM_E_SHAPE,
               "dropout needs 0 <= p < 1, a device seed word and an even T");
  for (int i : {1, 2})
    DALM_REQUIRE(T * strides[3 * i + 2] < (1ll << 30), DALM_E_SHAPE, "T x row stride must stay below 2^30 elements");
  AttnBwdParams p = {};
  set_dropout(p, dropout_p, seed, salt);
  p.cu = cu;
  p.q = static_cast<const unsigned short*>(q); p.k = static_cast<const unsigned short*>(k);
  p.v = static_cast<const unsigned short*>(v);
  p.bits_rows = bits_rows; p.live = live;
  p.dq = static_cast<unsigned short*>(o);
  p.delta = lse;
  p.B = static_cast<int>(B); p.H = static_cast<int>(H); p.T = static_cast<int>(T); p.W = static_cast<int>((T + 31) / 32);
  p.scale = scale;
  for (int a = 0; a < 3; ++a) {
    p.s[0][a] = strides[a]; p.s[1][a] = strides[3 + a]; p.s[2][a] = strides[6 + a]; p.s[5][a] = strides[9 + a];
  }
  const int64_t pairs8 = (B * H + 7) / 8 * 8;
  const dim3 grid(static_cast<unsigned>(pairs8 * ((T + 127) / 128)));
  hipStream_t s = as_stream(stream);
  static const bool first_form = [] { const char* e = getenv("DALM_ATTN_FWD"); return e && e[0] == '1'; }();   // A/B runs
  if (first_form) {
    if (hd == 128) {
      if (p.seed) hipLaunchKernelGGL((attn_fwd_kernel<128, true>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((attn_fwd_kernel<128, false>), grid, dim3(256), 0, s, p);
    } else {
      if (p.seed) hipLaunchKernelGGL((attn_fwd_kernel<64, true>), grid, dim3(256), 0, s, p);
      else hipLaunchKernelGGL((attn_fwd_kernel<64, false>), grid, dim3(256), 0, s, p);
    }
  } else if (hd == 128) {
    if (p.seed) hipLaunchKernelGGL((attn_fwd2_kernel<128, true>), grid, dim3(256), stream2_lds<128>(), s, p);
    else hipLaunchKernelGGL((attn_fwd2_kernel<128, false>), grid, dim3(256), stream2_lds<128>(), s, p);
  } else {
    if (p.seed) hipLaunchKernelGGL((attn_fwd2_kernel<64, true>), grid, dim3(256), stream2_lds<64>(), s, p);
    else hipLaunchKernelGGL((attn_fwd2_kernel<64, false>), grid, dim3(256), stream2_lds<64>(), s, p);
  }
  return check_launch(__func__);
}

extern "C" int dalm_attn_fwd(const void* q, const void* k, const void* v, const uint32_t* bits_rows, const uint8_t* live, int64_t B,
                             int64_t H, int64_t T, int64_t hd, float scale, const int64_t* strides, float dropout_p, const void* seed,
                             uint32_t salt, void* o, float* lse, dalm_stream_t stream) {
  return dalm_attn_fwd_any(q, k, v, bits_rows, live, nullptr, B, H, T, hd, scale, strides, dropout_p, seed, salt, o, lse, stream);
}

extern "C" int dalm_attn_fwd_packed(const void* q, const void* k, const void* v, const uint32_t* bits_rows, const uint8_t* live,
                                    const int32_t* cu_seqlens, int64_t B, int64_t H, int64_t T, int64_t hd, float scale,
                                    const int64_t* strides, float dropout_p, const void* seed, uint32_t salt, void* o, float* lse,
                                    dalm_stream_t stream) {
  DALM_REQUIRE(cu_seqlens, DALM_E_NULL, "null pointer argument");
  return dalm_attn_fwd_any(q, k, v, bits_rows, live, cu_seqlens, B, H, T, hd, scale, strides, dropout_p, seed, salt, o, lse, stream);
}
