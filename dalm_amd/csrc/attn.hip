// Backward of scaled-dot-product attention for a decoder layer of the generator tower, bf16, head width 128 (Llama-2-7b:
// BASELINE.json configs 3 / 4), arbitrary boolean mask (HF's causal + left-padding mask) - hand-written for gfx950.
// transformers reaches torch.nn.functional.scaled_dot_product_attention through sdpa_attention_forward
// (transformers/integrations/sdpa_attention.py); the reference reaches it through self.generator_model(...)
// (dalm/models/rag_e2e_base_model.py:104-106) and differentiates it with loss.backward()
// (dalm/training/rag_e2e/train_rage2e.py:466).  With a mask torch dispatches its memory-efficient kernels; their backward at
// cfg3 (B 18, H 32, T 256, hd 128) is preprocess 15 us + dk/dv 264 us + dq 160 us = 440 us per layer, 14 ms of a 138 ms step,
// 2.7 % of the MFMA peak (profiles/r05_step_by_stream.txt).  The forward stays torch's (60 us); its log-sum-exp is this file's input.
//
//   P = exp(scale S + mask - lse),  S = Q K^T        dV = P^T dO          dP = dO V^T
//   dS = P o (dP - D),  D_i = sum_d dO[i,d] O[i,d]   dQ = scale dS K      dK = scale dS^T Q
//
// Two launches, no atomics, every product on v_mfma_f32_32x32x16_bf16 (C tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2)
// + 4 (lane >> 5)):
//   attn_bwd_dq_kernel    a workgroup owns 128 query rows (a wave 32 of them; Q and dO fragments stay in registers as B operands,
//                         lane <-> query row), streams 64-row K / V blocks through LDS, computes S^T and dP^T tiles
//                         [key row (regs), query row (lane)].  The dS^T tile, rounded to bf16, IS the B operand of
//                         dQ^T[d, i] += K^T[d, j] dS^T[j, i] with the contraction index taken in the tile's register order
//                         (k-step s of a lane half h holds rows 16 s + 4 h + {0..3} and 16 s + 8 + 4 h + {0..3}); the A operand
//                         K^T is read from a transposed LDS copy with two 8-byte reads in that same order.  Also writes D.
//   attn_bwd_dkdv_kernel  a workgroup owns 128 key rows (K, V fragments in registers, lane <-> key row), streams 64-row Q / dO
//                         blocks (+ transposed copies), computes S and dP tiles [query row (regs), key row (lane)];
//                         P and dS are the B operands of dV^T[d, j] += dO^T[d, i] P[i, j] and dK^T[d, j] += Q^T[d, i] dS[i, j].
// The mask is read as BITS: attn_mask_bits_kernel packs the [B, 1, T, T] boolean mask once per step (the same mask serves every
// layer and head) into row words (bit c of word w of row i = mask[i, 32 w + c]) and column words, plus one byte per 32 x 32 tile
// that says whether anything in it is live: a lane's 32 mask bits of a tile are ONE dword, dead tiles (the causal upper
// triangle, padding) cost nothing.
// Algorithmic bytes per launch pair: q, k, v, o, dO read + dq, dk, dv written = 8 B H T hd el (302 MB at cfg3); algorithmic
// flops 5 GEMMs x 2 T^2 hd per head over the live tiles (dq and dk/dv each recompute S and dP: 7 are executed).
#include "common.hpp"

namespace dalm {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHd = 128;
constexpr int kLRow = 2 * kHd + 16;      // bytes of a [row][128] LDS row (16 bytes of padding: conflict-free 16-byte reads)
constexpr int kTRow = 2 * 64 + 8;        // bytes of a [d][64 rows] transposed LDS row (8-byte reads stay aligned and spread)
constexpr int kTile = 64 * kLRow;        // 17408 = 128 * kTRow as well

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
};

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

// one 64-row block of a [rows][128] bf16 tensor -> LDS, row-major (`rm`) and, when `tr` is given, transposed ([d][row]).
// lane <-> row (the transposed 2-byte stores of a wave are then one contiguous 128-byte run), wave w takes 16-byte chunks 4 w .. 4 w + 3
__device__ __forceinline__ void stage_block(const unsigned short* base, int64_t row_stride, int row0, int T, unsigned char* rm,
                                            unsigned char* tr, int w, int l) {
  const int row = row0 + l;
  const bool ok = row < T;
  const unsigned short* src = base + static_cast<int64_t>(row) * row_stride;
  uint4 v[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) v[n] = ok ? ld16(src + 8 * (4 * w + n)) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int c = 4 * w + n;
    *reinterpret_cast<uint4*>(rm + l * kLRow + 16 * c) = v[n];
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

// A operand of a product whose contraction index runs over the ROWS of a 32 x 32 C tile held as the B operand:
// k-step s, lane half h: rows 16 s + 4 h + {0..3}, then 16 s + 8 + 4 h + {0..3}
__device__ __forceinline__ bf16x8 ld_tr_frag(const unsigned char* tr, int drow, int r0, int s, int hi) {
  const unsigned char* a = tr + drow * kTRow + 2 * (r0 + 16 * s + 4 * hi);
  const uint2 lo = *reinterpret_cast<const uint2*>(a), up = *reinterpret_cast<const uint2*>(a + 16);
  return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, up.x, up.y));
}

// a wave's [32 rows][128] accumulators held transposed (acc[dblk]: row d = 32 dblk + .., column = lane & 31 <-> the wave's row)
// -> LDS [128 rows][kLRow] as bf16
__device__ __forceinline__ void spill_transposed(const f32x16 (&acc)[4], float mul, unsigned char* out, int w, int l31, int hi) {
#pragma unroll
  for (int dblk = 0; dblk < 4; ++dblk)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint2 pk;
      pk.x = pack_bf16x2(acc[dblk][4 * q] * mul, acc[dblk][4 * q + 1] * mul);
      pk.y = pack_bf16x2(acc[dblk][4 * q + 2] * mul, acc[dblk][4 * q + 3] * mul);
      *reinterpret_cast<uint2*>(out + (32 * w + l31) * kLRow + 2 * (32 * dblk + 8 * q + 4 * hi)) = pk;
    }
}
__device__ __forceinline__ void store_rows(const unsigned char* out, unsigned short* dst, int64_t row_stride, int row0, int T, int t) {
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int idx = t + 256 * n, row = idx >> 4, c = idx & 15;
    if (row0 + row < T)
      *reinterpret_cast<uint4*>(dst + static_cast<int64_t>(row0 + row) * row_stride + 8 * c) =
          *reinterpret_cast<const uint4*>(out + row * kLRow + 16 * c);
  }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnBwdParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * kTile];
  unsigned char* Ks = lds;
  unsigned char* Vs = lds + kTile;
  unsigned char* KT = lds + 2 * kTile;
  const int t = threadIdx.x, w = t >> 6, l = t & 63, l31 = l & 31, hi = l >> 5;
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * 128;
  const int i = i0 + 32 * w + l31;
  const bool iok = i < p.T;
  const int64_t bh = static_cast<int64_t>(b) * p.H + h;
  const unsigned short* qrow = p.q + b * p.s[0][0] + h * p.s[0][1] + static_cast<int64_t>(i) * p.s[0][2];
  const unsigned short* orow = p.o + b * p.s[3][0] + h * p.s[3][1] + static_cast<int64_t>(i) * p.s[3][2];
  const unsigned short* grow = p.d_o + b * p.s[4][0] + h * p.s[4][1] + static_cast<int64_t>(i) * p.s[4][2];
  const unsigned short* kbase = p.k + b * p.s[1][0] + h * p.s[1][1];
  const unsigned short* vbase = p.v + b * p.s[2][0] + h * p.s[2][1];

  bf16x8 Qb[8], Gb[8];
  float Dl = 0.f;
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const int off = 16 * kk + 8 * hi;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    const uint4 a = iok ? ld16(qrow + off) : z, g = iok ? ld16(grow + off) : z, o = iok ? ld16(orow + off) : z;
    Qb[kk] = __builtin_bit_cast(bf16x8, a);
    Gb[kk] = __builtin_bit_cast(bf16x8, g);
    Dl += dot8(g, o);
  }
  Dl += __shfl_xor(Dl, 32, 64);
  const float nl = iok ? -p.lse[bh * p.T + i] * kLog2e : 0.f;
  if (iok && hi == 0) p.delta[bh * p.T + i] = Dl;
  const float c1 = p.scale * kLog2e;
  const int Tp = 32 * p.W;

  f32x16 acc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  const int nJ = (p.T + 63) >> 6, ib32 = i0 >> 5;
  for (int jb = 0; jb < nJ; ++jb) {
    int any = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        if (ib32 + a < p.W && 2 * jb + c < p.W) any |= p.live[(static_cast<int64_t>(b) * p.W + ib32 + a) * p.W + 2 * jb + c];
    if (!any) continue;                                        // uniform over the workgroup
    uint32_t word[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
      word[c] = (i < Tp && 2 * jb + c < p.W) ? p.bits_rows[(static_cast<int64_t>(b) * Tp + i) * p.W + 2 * jb + c] : 0u;
    __syncthreads();                                           // the previous block's fragments have been read
    stage_block(kbase, p.s[1][2], 64 * jb, p.T, Ks, KT, w, l);
    stage_block(vbase, p.s[2][2], 64 * jb, p.T, Vs, nullptr, w, l);
    __syncthreads();
#pragma unroll
    for (int js = 0; js < 2; ++js) {
      if (__builtin_amdgcn_ballot_w64(word[js] != 0u) == 0ull) continue;
      f32x16 St, Pt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { St[r] = 0.f; Pt[r] = 0.f; }
      const unsigned char* ka = Ks + (32 * js + l31) * kLRow + 16 * hi;
      const unsigned char* va = Vs + (32 * js + l31) * kLRow + 16 * hi;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        St = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(ka + 32 * kk), Qb[kk], St, 0, 0, 0);
        Pt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(va + 32 * kk), Gb[kk], Pt, 0, 0, 0);
      }
      unsigned int pk[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float ds[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int jl = ((r + u) & 3) + 8 * ((r + u) >> 2) + 4 * hi;
          const float pv = ((word[js] >> jl) & 1u) ? __builtin_amdgcn_exp2f(fmaf(St[r + u], c1, nl)) : 0.f;
          ds[u] = pv * (Pt[r + u] - Dl);
        }
        pk[r >> 1] = pack_bf16x2(ds[0], ds[1]);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 dsb = __builtin_bit_cast(bf16x8, make_uint4(pk[4 * s], pk[4 * s + 1], pk[4 * s + 2], pk[4 * s + 3]));
#pragma unroll
        for (int d = 0; d < 4; ++d)
          acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr_frag(KT, 32 * d + l31, 32 * js, s, hi), dsb, acc[d], 0, 0, 0);
      }
    }
  }
  __syncthreads();
  spill_transposed(acc, p.scale, lds, w, l31, hi);
  __syncthreads();
  store_rows(lds, p.dq + b * p.s[5][0] + h * p.s[5][1], p.s[5][2], i0, p.T, t);
}

constexpr int kDkdvLds = 4 * kTile + 2 * 64 * 4;

__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv_kernel(const AttnBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dlds[];
  unsigned char* Qs = dlds;
  unsigned char* Gs = dlds + kTile;
  unsigned char* QT = dlds + 2 * kTile;
  unsigned char* GT = dlds + 3 * kTile;
  float* nl_s = reinterpret_cast<float*>(dlds + 4 * kTile);
  float* dl_s = nl_s + 64;
  const int t = threadIdx.x, w = t >> 6, l = t & 63, l31 = l & 31, hi = l >> 5;
  const int b = blockIdx.z, h = blockIdx.y, j0 = blockIdx.x * 128;
  const int j = j0 + 32 * w + l31;
  const bool jok = j < p.T;
  const int64_t bh = static_cast<int64_t>(b) * p.H + h;
  const unsigned short* krow = p.k + b * p.s[1][0] + h * p.s[1][1] + static_cast<int64_t>(j) * p.s[1][2];
  const unsigned short* vrow = p.v + b * p.s[2][0] + h * p.s[2][1] + static_cast<int64_t>(j) * p.s[2][2];
  const unsigned short* qbase = p.q + b * p.s[0][0] + h * p.s[0][1];
  const unsigned short* gbase = p.d_o + b * p.s[4][0] + h * p.s[4][1];

  bf16x8 Kb[8], Vb[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const int off = 16 * kk + 8 * hi;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    Kb[kk] = __builtin_bit_cast(bf16x8, jok ? ld16(krow + off) : z);
    Vb[kk] = __builtin_bit_cast(bf16x8, jok ? ld16(vrow + off) : z);
  }
  const float c1 = p.scale * kLog2e;
  const int Tp = 32 * p.W;
  f32x16 dVt[4], dKt[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dVt[d][r] = 0.f; dKt[d][r] = 0.f; }

  const int nI = (p.T + 63) >> 6, jb32 = j0 >> 5;
  for (int ib = 0; ib < nI; ++ib) {
    int any = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int a = 0; a < 4; ++a)
        if (2 * ib + c < p.W && jb32 + a < p.W) any |= p.live[(static_cast<int64_t>(b) * p.W + 2 * ib + c) * p.W + jb32 + a];
    if (!any) continue;
    uint32_t word[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
      word[c] = (j < Tp && 2 * ib + c < p.W) ? p.bits_cols[(static_cast<int64_t>(b) * Tp + j) * p.W + 2 * ib + c] : 0u;
    __syncthreads();
    stage_block(qbase, p.s[0][2], 64 * ib, p.T, Qs, QT, w, l);
    stage_block(gbase, p.s[4][2], 64 * ib, p.T, Gs, GT, w, l);
    if (t < 64) {
      const int i = 64 * ib + t;
      nl_s[t] = i < p.T ? -p.lse[bh * p.T + i] * kLog2e : 0.f;
      dl_s[t] = i < p.T ? p.delta[bh * p.T + i] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int is = 0; is < 2; ++is) {
      if (__builtin_amdgcn_ballot_w64(word[is] != 0u) == 0ull) continue;
      f32x16 S, dP;
#pragma unroll
      for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
      const unsigned char* qa = Qs + (32 * is + l31) * kLRow + 16 * hi;
      const unsigned char* ga = Gs + (32 * is + l31) * kLRow + 16 * hi;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
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
          ds[u] = pv[u] * (dP[4 * q + u] - dl[u]);
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
        for (int d = 0; d < 4; ++d) {
          dVt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr_frag(GT, 32 * d + l31, 32 * is, s, hi), pb, dVt[d], 0, 0, 0);
          dKt[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_tr_frag(QT, 32 * d + l31, 32 * is, s, hi), db, dKt[d], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();
  spill_transposed(dKt, p.scale, dlds, w, l31, hi);
  __syncthreads();
  store_rows(dlds, p.dk + b * p.s[6][0] + h * p.s[6][1], p.s[6][2], j0, p.T, t);
  __syncthreads();
  spill_transposed(dVt, 1.0f, dlds, w, l31, hi);
  __syncthreads();
  store_rows(dlds, p.dv + b * p.s[7][0] + h * p.s[7][1], p.s[7][2], j0, p.T, t);
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

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace dalm

using namespace dalm;

extern "C" int dalm_attn_mask_bits(const void* mask, int64_t B, int64_t T, int64_t mask_stride_b, int64_t mask_stride_row, int causal,
                                   uint32_t* bits_rows, uint32_t* bits_cols, uint8_t* live, dalm_stream_t stream) {
  DALM_REQUIRE(bits_rows && bits_cols && live, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(B > 0 && T > 0 && T <= 32768 && B <= 65535, DALM_E_SHAPE, "need 0 < T <= 32768 and 0 < B <= 65535");
  DALM_REQUIRE(mask || causal, DALM_E_NULL, "neither a mask nor the causal flag: nothing to pack");
  const int64_t W = (T + 31) / 32, total = B * 32 * W * W;
  hipStream_t s = as_stream(stream);
  if (hipError_t e = hipMemsetAsync(live, 0, static_cast<size_t>(B * W * W), s); e != hipSuccess) return fail(static_cast<int>(e), __func__, hipGetErrorString(e));
  hipLaunchKernelGGL(attn_mask_bits_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, s,
                     static_cast<const unsigned char*>(mask), static_cast<int>(B), static_cast<int>(T), static_cast<int>(W),
                     mask_stride_b, mask_stride_row, causal, bits_rows, bits_cols, live);
  return check_launch(__func__);
}

extern "C" int dalm_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                             const uint32_t* bits_rows, const uint32_t* bits_cols, const uint8_t* live, int64_t B, int64_t H,
                             int64_t T, int64_t hd, float scale, const int64_t* strides, void* dq, void* dk, void* dv, float* delta,
                             dalm_stream_t stream) {
  DALM_REQUIRE(q && k && v && o && d_o && lse && bits_rows && bits_cols && live && strides && dq && dk && dv && delta, DALM_E_NULL,
               "null pointer argument");
  DALM_REQUIRE(hd == kHd, DALM_E_SHAPE, "head width must be 128");
  DALM_REQUIRE(B > 0 && H > 0 && T > 0 && T <= 32768 && B <= 65535 && H <= 65535, DALM_E_SHAPE, "need 0 < T <= 32768, 0 < B, H <= 65535");
  const void* ptrs[8] = {q, k, v, o, d_o, dq, dk, dv};
  for (int i = 0; i < 8; ++i) {
    DALM_REQUIRE(al16(ptrs[i]), DALM_E_ALIGN, "tensors must be 16-byte aligned");
    for (int a = 0; a < 3; ++a)
      DALM_REQUIRE(strides[3 * i + a] >= 0 && strides[3 * i + a] % 8 == 0, DALM_E_ALIGN, "strides must be non-negative multiples of 8 elements");
  }
  AttnBwdParams p;
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
  if (!lds_set) {
    if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kDkdvLds);
        e != hipSuccess)
      return fail(static_cast<int>(e), __func__, "could not raise the dynamic LDS limit of the dk / dv kernel");
    lds_set = true;
  }
  const dim3 grid(static_cast<unsigned>((T + 127) / 128), static_cast<unsigned>(H), static_cast<unsigned>(B));
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, s, p);
  hipLaunchKernelGGL(attn_bwd_dkdv_kernel, grid, dim3(256), kDkdvLds, s, p);
  return check_launch(__func__);
}
