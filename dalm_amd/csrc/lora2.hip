// dalm_lora2_*: the low-rank branch of LoRA-wrapped projections that SHARE their input (q_proj / v_proj of a Llama block,
// query / key / value of a BERT block), bf16 activations, gfx950.
//
// The reference asks peft for r = 8, alpha = 16, dropout 0.05 adapters (dalm/models/rag_e2e_base_model.py:61-80,145-160); per
// wrapped projection peft evaluates  out = W x + s * B(A(dropout(x))).  Round 4 (lora.hip) ran that branch as three streaming
// kernels per projection, each regenerating the dropout mask from a counter hash.  Measured there (profiles/history/r04_step_by_stream.txt,
// [4608, 4096] bf16): 0.20-0.27 of the HBM rate on the read-only kernels.  The instruction count explains it - the mask hash
// (4 multiplies + ~20 integer ops per two elements, in three kernels x two projections) and 8 dword loads per MFMA step for a
// [N, r] weight - plus a 288-workgroup grid on 256 CUs with one resident workgroup each (two rounds).  This file:
//   * the mask is computed ONCE, in the forward rowdot, with a 2-multiply hash (mask v2, oracle/lora_mask.py::keep_mask_v2),
//     and stored as one bit per element; the backward kernels read bits (1/16 of the activation bytes);
//   * projections that share x are STACKED: one pass over x yields z_q and z_v (an MFMA tile = 8 rows x {q, v}: the A operand
//     carries the row under q's mask in rows 0-7 and under v's mask in rows 8-15, the B operand A_q in columns 0-7 and A_v in
//     8-15; the two diagonal 8 x 8 blocks of the product are kept), one pass over x yields dA_q and dA_v, one pass over dx adds
//     both rank updates; independent problems (g_q, g_v) share a launch through the grid's y / z dimension;
//   * every weight operand is [r, K]-major (lora_A as stored; lora_B is kept in [r, N]-major memory by models/lora.py), so
//     every operand load is 16 bytes wide;
//   * all loads of the next batch are in flight while the current one is processed (explicit register double buffering: the
//     in-place updates could not be pipelined by the compiler), grids of 500-1500 workgroups with 2-4 resident per CU;
//   * colacc finishes in the launch (the last workgroup of a column slab to arrive adds the row splits in fixed order).
// Algorithmic bytes ([R, C] bf16 activation): rowdot R*C*2 (+ R*C/8 per mask written), rankupd 2*R*C*2 (+ R*C/8 per mask read),
// colacc R*C*2 (+ R*C/8 per mask read).
#include "lora_common.hpp"

namespace dalm {
namespace {
using namespace lora;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// ---- mask v2: chunk index c = flat element index >> 3 (8 elements);  w_0 = mix2(c ^ key.a, key.b),  w_{q+1} = xorshift32(w_q);
// element 8c + 2q is kept iff (w_q & 0xffff) >= thr, element 8c + 2q + 1 iff (w_q >> 16) >= thr.  mix2 = lowbias32 with key.b
// added between its two multiplies; xorshift32: w ^= w << 13; w ^= w >> 17; w ^= w << 5.  (One two-multiply hash per word - the
// first form - cost the forward kernel 16.5 us of integer VALU at [4608, 4096] x 2 adapters: profiles/r05_lora_rowdot_ablation.txt.)
__device__ __forceinline__ unsigned int mix2(unsigned int x, unsigned int b) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x += b; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// 8 bf16 elements (4 packed words) starting at flat index e0 (a multiple of 8): dropped elements are zeroed in place; returns the
// keep byte (bit e = element e0 + e survives).  thr_m1 = (thr16 - 1) in both halves, thr16 >= 1.
__device__ __forceinline__ unsigned int mask8(const DropKey& k, unsigned int e0, unsigned int thr_m1, uint4& v) {
  unsigned int wds[4] = {v.x, v.y, v.z, v.w};
  unsigned int packed = 0;
  const u16x2 zero = {0, 0};
  unsigned int h = mix2((e0 >> 3) ^ k.a, k.b);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; }
    // packed 16-bit ops (v_pk_sub_u16 clamp, v_pk_min_u16, v_pk_sub_u16); the min is written as an instruction because the
    // optimiser folds min(sat_sub, 1) back into two compares and two selects per word
    const u16x2 g = __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, h), __builtin_bit_cast(u16x2, thr_m1));
    unsigned int kpw;                                              // per half: 1 = keep (field >= thr), 0 = drop
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(kpw) : "v"(__builtin_bit_cast(unsigned int, g)), "v"(0x00010001u));
    const u16x2 kp = __builtin_bit_cast(u16x2, kpw);
    wds[q] &= __builtin_bit_cast(unsigned int, static_cast<u16x2>(zero - kp));   // 0xffff / 0 per half
    packed |= kpw << (2 * q);                                      // bits 2q (element 2q) and 16 + 2q (element 2q + 1)
  }
  v = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  return (packed & 0x55u) | ((packed >> 15) & 0xAAu);
}
__device__ __forceinline__ DropKey drop_key2(const unsigned long long* seed, unsigned int salt) {
  DropArgs d; d.seed = seed; d.salt = salt; d.thr16 = 0;
  return drop_key(d);
}

// f32 pair -> bf16 high parts and bf16 residuals (the MFMA B operand keeps f32 accuracy through two products)
__device__ __forceinline__ void split_pair(float a, float b, unsigned int& hi, unsigned int& lo) {
  hi = pack_bf16x2(a, b);
  lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}

// ---------------------------------------------------------------------------------------------------
// rowdot: out_t[row][j] = scale * sum_k m_t x_t[row][k] W_t[j][k]   on v_mfma_f32_16x16x32_bf16.
//   MODE 1: one problem, 16-row tiles.   MODE 3: two independent problems of one shape (blockIdx.y), no dropout needed by callers
//   MODE 2: two projections of ONE x (rank 8): 8-row tiles, see the file header.
// 256 threads; wave w takes the 32-column steps w, w + 4, ... in batches of UN with the next batch's loads in flight.
// Tried and dropped (profiles/r05_lora_rowdot_ablation.txt): a W-stationary form - workgroup = 256-column slab x 128 rows, W split
// into B fragments once, the K slabs' partial tiles added in the launch by the last slab to arrive.  Its loads and MFMAs ran at
// 4.2 TB/s (18 us for two projections), but the write-through partial tiles + ticket + last-arriver pass cost another 13-18 us
// at the end of the kernel, and an EMPTY kernel of either shape takes 6-7 us in a hipGraph: at 37.7 MB per activation the
// launch itself is half of the 0.55-of-HBM budget, so the lever that is left is more projections per launch.
// ---------------------------------------------------------------------------------------------------
template <int MODE, bool DROP>
__global__ __launch_bounds__(256) void lora2_rowdot_kernel(
    const bf16_t* __restrict__ x0, const bf16_t* __restrict__ x1, const float* __restrict__ W0, const float* __restrict__ W1,
    float* __restrict__ out0, float* __restrict__ out1, unsigned char* __restrict__ bits0, unsigned char* __restrict__ bits1,
    const unsigned long long* __restrict__ seed, unsigned int salt0, unsigned int salt1, unsigned int thr16, int R, int K,
    int rank, float scale) {
  __shared__ float red[4][16][17];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kg = lane >> 4;
  const int prob = MODE == 3 ? static_cast<int>(blockIdx.y) : 0;
  const int lp = MODE == 2 ? (i >> 3) : prob;                 // the projection this lane's operand row / column belongs to
  const int li = MODE == 2 ? (i & 7) : i;                     // row within the tile / rank index
  const int row = static_cast<int>(blockIdx.x) * (MODE == 2 ? 8 : 16) + li;
  const int rowc = min(row, R - 1);
  const bf16_t* xr = ((MODE == 3 && prob) ? x1 : x0) + static_cast<int64_t>(rowc) * K + kg * 8;
  const bool wv = li < rank;                                  // this lane's B-operand column is a real rank index
  const float* wr = (lp ? W1 : W0) + static_cast<int64_t>(min(li, rank - 1)) * K + kg * 8;
  DropKey key{0u, 1u};
  unsigned char* bp = nullptr;
  unsigned int thr_m1 = 0;
  if constexpr (DROP) {
    key = drop_key2(seed, lp ? salt1 : salt0);
    bp = (lp ? bits1 : bits0) + static_cast<int64_t>(rowc) * (K >> 3) + kg;
    thr_m1 = (thr16 - 1u) * 0x00010001u;
  }
  const unsigned int ebase = static_cast<unsigned int>(rowc) * static_cast<unsigned int>(K) + kg * 8;
  const int nsteps = K >> 5;
  constexpr int UN = 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};

  auto issue = [&](uint4 (&xa)[UN], float4 (&wa)[UN][2], int s0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = min(s0 + 4 * u, nsteps - 1) * 32;           // clamped: the extra steps are skipped in process()
      xa[u] = *reinterpret_cast<const uint4*>(xr + k);
      if (MODE == 2 || wv) {                                    // the other lanes' columns are never written out: any value does
        wa[u][0] = *reinterpret_cast<const float4*>(wr + k);
        wa[u][1] = *reinterpret_cast<const float4*>(wr + k + 4);
      }
    }
  };
  auto process = [&](uint4 (&xa)[UN], float4 (&wa)[UN][2], int s0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int st = s0 + 4 * u;
      if (st >= nsteps) continue;                               // wave-uniform
      uint4 v = xa[u];
      if constexpr (DROP) {
        const unsigned int byte = mask8(key, ebase + static_cast<unsigned int>(st) * 32u, thr_m1, v);
        if (row < R) bp[st * 4] = static_cast<unsigned char>(byte);
      }
      unsigned int hi[4], lo[4];
      split_pair(wa[u][0].x, wa[u][0].y, hi[0], lo[0]);
      split_pair(wa[u][0].z, wa[u][0].w, hi[1], lo[1]);
      split_pair(wa[u][1].x, wa[u][1].y, hi[2], lo[2]);
      split_pair(wa[u][1].z, wa[u][1].w, hi[3], lo[3]);
      const bf16x8 a8 = __builtin_bit_cast(bf16x8, v);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, __builtin_bit_cast(bf16x8, make_uint4(lo[0], lo[1], lo[2], lo[3])), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, __builtin_bit_cast(bf16x8, make_uint4(hi[0], hi[1], hi[2], hi[3])), acc, 0, 0, 0);
    }
  };

  uint4 xa0[UN], xa1[UN];
  float4 wa0[UN][2], wa1[UN][2];
  issue(xa0, wa0, wave);
  for (int s0 = wave; s0 < nsteps; s0 += 8 * UN) {
    const int s1 = s0 + 4 * UN, s2 = s0 + 8 * UN;
    if (s1 < nsteps) issue(xa1, wa1, s1);
    __builtin_amdgcn_sched_barrier(0);
    process(xa0, wa0, s0);
    if (s1 < nsteps) {
      if (s2 < nsteps) issue(xa0, wa0, s2);
      __builtin_amdgcn_sched_barrier(0);
      process(xa1, wa1, s1);
    }
  }
  // accumulator tile: lane L holds rows 4 (L / 16) + r, column L % 16
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][kg * 4 + r][i] = acc[r];
  __syncthreads();
  {
    const int m = tid >> 4, j = tid & 15;
    const float s = ((red[0][m][j] + red[1][m][j]) + red[2][m][j]) + red[3][m][j];     // fixed order
    if constexpr (MODE == 2) {
      const int pm = m >> 3;
      const int orow = static_cast<int>(blockIdx.x) * 8 + (m & 7);
      if (pm == (j >> 3) && orow < R) (pm ? out1 : out0)[static_cast<int64_t>(orow) * 8 + (j & 7)] = scale * s;
    } else {
      const int orow = static_cast<int>(blockIdx.x) * 16 + m;
      if (j < rank && orow < R) (prob ? out1 : out0)[static_cast<int64_t>(orow) * rank + j] = scale * s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// rankupd: y[row][c] += scale * sum_t m_t(row, c) sum_j z_t[row][j] W_t[j][c],  in place.  W_t is [RANK][C].
//   NT = 1: one term;  TWO_Y: two independent problems (blockIdx.z);   NT = 2: two terms on ONE y (the backward dx of stacked
//   projections).  A wave owns one row at a time over the workgroup's 512-column slab (1 KB contiguous per wave load), the thread
//   its 8 columns and the RANK x 8 slice(s) of W; z rows arrive through scalar loads.
// ---------------------------------------------------------------------------------------------------
template <int RANK>
__device__ __forceinline__ void load_w_rows(const float* __restrict__ W, int C, int c0, bool valid, float (&w)[RANK][8]) {
#pragma unroll
  for (int j = 0; j < RANK; ++j) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (valid) {
      a = *reinterpret_cast<const float4*>(W + static_cast<int64_t>(j) * C + c0);
      b = *reinterpret_cast<const float4*>(W + static_cast<int64_t>(j) * C + c0 + 4);
    }
    w[j][0] = a.x; w[j][1] = a.y; w[j][2] = a.z; w[j][3] = a.w; w[j][4] = b.x; w[j][5] = b.y; w[j][6] = b.z; w[j][7] = b.w;
  }
}

template <int RANK, int NT, bool BITS, bool TWO_Y, bool STAGE>
__global__ __launch_bounds__(256) void lora2_rankupd_kernel(
    bf16_t* __restrict__ y0, bf16_t* __restrict__ y1, const float* __restrict__ z0, const float* __restrict__ z1,
    const float* __restrict__ W0, const float* __restrict__ W1, const unsigned char* __restrict__ bits0,
    const unsigned char* __restrict__ bits1, int R, int C, int rows_per_wg, float scale) {
  static_assert(!(TWO_Y && NT == 2), "two problems carry one term each");
  constexpr int UN = NT == 2 ? 3 : 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int prob = TWO_Y ? static_cast<int>(blockIdx.z) : 0;
  const int c0 = (static_cast<int>(blockIdx.x) * 64 + lane) * 8;
  const bool valid = c0 < C;
  const int r_lo = static_cast<int>(blockIdx.y) * rows_per_wg, r_hi = min(R, r_lo + rows_per_wg);
  bf16_t* y = prob ? y1 : y0;
  const float* zt[NT];
  const unsigned char* bt[NT];
  float w[NT][RANK][8];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int q = TWO_Y ? prob : t;
    zt[t] = q ? z1 : z0;
    bt[t] = q ? bits1 : bits0;
    load_w_rows<RANK>(q ? W1 : W0, C, c0, valid, w[t]);
  }
  const int cb = c0 >> 3, bpr = C >> 3;
  // STAGE (C % 128 == 0): the workgroup's mask bytes - 64 contiguous bytes per row and term - come in through LDS with 16-byte
  // loads; one byte load per lane, row and term (the first form) cost the dx pass as much as the activation itself
  extern __shared__ __attribute__((aligned(16))) unsigned char sbits[];      // [NT][rows_per_wg][64]
  if constexpr (BITS && STAGE) {
    for (int idx = tid; idx < NT * rows_per_wg * 4; idx += 256) {
      const int piece = idx & 3, rw = (idx >> 2) % rows_per_wg, t = (idx >> 2) / rows_per_wg;
      const int row = r_lo + rw, col_byte = static_cast<int>(blockIdx.x) * 64 + piece * 16;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (row < r_hi && col_byte < bpr) v = *reinterpret_cast<const uint4*>(bt[t] + static_cast<int64_t>(row) * bpr + col_byte);
      *reinterpret_cast<uint4*>(sbits + (t * rows_per_wg + rw) * 64 + piece * 16) = v;
    }
    __syncthreads();
  }

  auto issue = [&](uint4 (&raw)[UN], unsigned int (&by)[NT][UN], int b0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int row = min(b0 + 4 * u, R - 1);
      raw[u] = valid ? *reinterpret_cast<const uint4*>(y + static_cast<int64_t>(row) * C + c0) : make_uint4(0u, 0u, 0u, 0u);
      if constexpr (BITS) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if constexpr (STAGE) by[t][u] = sbits[(t * rows_per_wg + min(b0 + 4 * u, r_hi - 1) - r_lo) * 64 + lane];
          else by[t][u] = valid ? bt[t][static_cast<int64_t>(row) * bpr + cb] : 0u;
        }
      }
    }
  };
  auto process = [&](uint4 (&raw)[UN], unsigned int (&by)[NT][UN], int b0) __attribute__((always_inline)) {
    float zr[UN][NT][RANK];
#pragma unroll
    for (int u = 0; u < UN; ++u) {                                 // the batch's z rows first: scalar loads, one wait
      const int row = min(b0 + 4 * u, R - 1);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < RANK; ++j) zr[u][t][j] = zt[t][static_cast<int64_t>(row) * RANK + j];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int row = b0 + 4 * u;
      if (row >= r_hi) continue;                                   // wave-uniform
      float yv[8];
      Chunk8<bf16_t>::decode(raw[u], yv);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < RANK; ++j) acc = fmaf(zr[u][t][j], w[t][j][e], acc);
          const float upd = fmaf(scale, acc, yv[e]);
          if constexpr (BITS) yv[e] = (by[t][u] >> e) & 1u ? upd : yv[e];
          else yv[e] = upd;
        }
      }
      if (valid) Chunk8<bf16_t>::store(y + static_cast<int64_t>(row) * C + c0, yv);
    }
  };

  uint4 raw0[UN], raw1[UN];
  unsigned int by0[NT][UN], by1[NT][UN];
  const int first = r_lo + wave;
  if (first >= r_hi) return;                                       // no barrier below this point
  issue(raw0, by0, first);
  for (int b0 = first; b0 < r_hi; b0 += 8 * UN) {
    const int b1 = b0 + 4 * UN, b2 = b0 + 8 * UN;
    if (b1 < r_hi) issue(raw1, by1, b1);
    __builtin_amdgcn_sched_barrier(0);
    process(raw0, by0, b0);
    if (b1 < r_hi) {
      if (b2 < r_hi) issue(raw0, by0, b2);
      __builtin_amdgcn_sched_barrier(0);
      process(raw1, by1, b1);
    }
  }
}

// PER per-lane sums -> added over the 8 lanes that differ in lane bits 3..5; lane L keeps the PER/8 sums starting at
// PER/8 * (L >> 3).  Each stage exchanges HALF of what is left (lane pair (l, l ^ bit): one keeps the lower half of the values,
// the other the upper half): PER/2 + PER/4 + PER/8 exchanges instead of 3 PER butterfly shuffles through the LDS crossbar.
template <int PER>
__device__ __forceinline__ void fold_rowlanes(float (&a)[PER], int lane) {
#pragma unroll
  for (int v = 0; v < PER / 2; ++v) {        // bit 5: v_permlane32_swap: vdst' = (x[0:31], y[0:31]), src' = (x[32:63], y[32:63])
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[v]), __float_as_uint(a[v + PER / 2]), false, false);
    a[v] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
#pragma unroll
  for (int v = 0; v < PER / 4; ++v) {        // bit 4: v_permlane16_swap: odd 16-lane rows of vdst <-> even rows of src
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[v]), __float_as_uint(a[v + PER / 4]), false, false);
    a[v] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  const bool up = (lane & 8) != 0;
#pragma unroll
  for (int v = 0; v < PER / 8; ++v) {        // bit 3: DPP row_ror:8 (lane l <- lane l ^ 8 of its 16-lane row)
    const float keep = up ? a[v + PER / 8] : a[v], send = up ? a[v] : a[v + PER / 8];
    a[v] = keep + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(send), 0x128, 0xf, 0xf, false));
  }
}

// ---------------------------------------------------------------------------------------------------
// colacc: out_t[j][c] = scale * sum_row m_t(row, c) x[row][c] z_t[row][j]      (out is [RANK][C])
//   NT = 1: one problem;  TWO_X: two independent problems (blockIdx.z);  NT = 2: two (z, mask) pairs over ONE x.
// grid (C / 64 slabs, row splits[, 2]); 256 threads = 8 column chunks x 32 row lanes; the split's z rows are staged in LDS; every
// thread keeps NT x 8 x RANK sums.  The splits' partial sums leave through agent-scope 8-byte stores; the last workgroup of a
// slab to arrive (one ticket per slab and problem) adds them in split order and writes the result: one launch.
// ---------------------------------------------------------------------------------------------------
template <int RANK, int NT, bool BITS, bool TWO_X, bool STAGE>
__global__ __launch_bounds__(256, 2) void lora2_colacc_kernel(
    const bf16_t* __restrict__ x0, const bf16_t* __restrict__ x1, const float* __restrict__ z0, const float* __restrict__ z1,
    const unsigned char* __restrict__ bits0, const unsigned char* __restrict__ bits1, float* __restrict__ out0,
    float* __restrict__ out1, unsigned long long* __restrict__ part /* [2][S][C * RANK / 2] pairs of floats */,
    unsigned int* __restrict__ tickets /* [slabs * (TWO_X ? 2 : 1)], zero on entry, left zero */, int R, int C,
    int rows_per_split, int S, float scale, int bits_off /* bytes: where the mask stage starts in LDS */) {
  static_assert(!(TWO_X && NT == 2), "two problems carry one term each");
  constexpr int UN = (NT * RANK > 8) ? 2 : 4;                      // rows in flight per thread and buffer (the sums take NT*8*RANK registers)
  extern __shared__ __attribute__((aligned(16))) float smem[];   // z stage [NT][rows_per_split][RANK], later red[4][8][NT*8*RANK]
  __shared__ unsigned int slot;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = tid & 7, rl = tid >> 3;
  const int prob = TWO_X ? static_cast<int>(blockIdx.z) : 0;
  const int c0 = static_cast<int>(blockIdx.x) * 64 + chunk * 8;
  const bool valid = c0 < C;
  const int r_lo = static_cast<int>(blockIdx.y) * rows_per_split, r_hi = min(R, r_lo + rows_per_split);
  const int nrow = r_hi - r_lo;
  const bf16_t* x = prob ? x1 : x0;
  const unsigned char* bt[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int q = TWO_X ? prob : t;
    bt[t] = q ? bits1 : bits0;
    const float* zsrc = (q ? z1 : z0) + static_cast<int64_t>(r_lo) * RANK;
    float* zdst = smem + t * rows_per_split * RANK;
    for (int i4 = tid; i4 < rows_per_split * RANK / 4; i4 += 256) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i4 * 4 < nrow * RANK) v = *reinterpret_cast<const float4*>(zsrc + i4 * 4);
      *reinterpret_cast<float4*>(zdst + i4 * 4) = v;
    }
  }
  const int cb = c0 >> 3, bpr = C >> 3;
  // STAGE (C % 64 == 0): the slab's 8 mask bytes per row and term come in through LDS, one 8-byte load per row and term
  unsigned char* sbits = reinterpret_cast<unsigned char*>(smem) + bits_off;      // [NT][rows_per_split][8]
  if constexpr (BITS && STAGE) {
    for (int idx = tid; idx < NT * rows_per_split; idx += 256) {
      const int t = idx / rows_per_split, rw = idx % rows_per_split, row = r_lo + rw;
      uint2 v = make_uint2(0u, 0u);
      if (row < r_hi) v = *reinterpret_cast<const uint2*>(bt[t] + static_cast<int64_t>(row) * bpr + static_cast<int>(blockIdx.x) * 8);
      *reinterpret_cast<uint2*>(sbits + static_cast<int64_t>(idx) * 8) = v;
    }
  }
  __syncthreads();

  float acc[NT][8][RANK];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int j = 0; j < RANK; ++j) acc[t][e][j] = 0.f;

  auto issue = [&](uint4 (&raw)[UN], unsigned int (&by)[NT][UN], int b0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int row = b0 + 32 * u;
      const bool ok = valid && row < r_hi;
      const int rr = min(row, R - 1);
      raw[u] = ok ? *reinterpret_cast<const uint4*>(x + static_cast<int64_t>(rr) * C + c0) : make_uint4(0u, 0u, 0u, 0u);
      if constexpr (BITS) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if constexpr (STAGE) by[t][u] = sbits[(t * rows_per_split + min(row - r_lo, rows_per_split - 1)) * 8 + chunk];
          else by[t][u] = ok ? bt[t][static_cast<int64_t>(rr) * bpr + cb] : 0u;
        }
      }
    }
  };
  auto process = [&](uint4 (&raw)[UN], unsigned int (&by)[NT][UN], int b0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int lr = min(b0 + 32 * u - r_lo, rows_per_split - 1);   // rows past the split: raw is zero, any staged z will do
      float xv[8];
      Chunk8<bf16_t>::decode(raw[u], xv);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float zr[RANK];
        const float* zp = smem + (t * rows_per_split + lr) * RANK;
#pragma unroll
        for (int j4 = 0; j4 < RANK; j4 += 4) {
          const float4 a = *reinterpret_cast<const float4*>(zp + j4);
          zr[j4] = a.x; zr[j4 + 1] = a.y; zr[j4 + 2] = a.z; zr[j4 + 3] = a.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float xe = xv[e];
          if constexpr (BITS) xe = (by[t][u] >> e) & 1u ? xe : 0.f;
#pragma unroll
          for (int j = 0; j < RANK; ++j) acc[t][e][j] = fmaf(xe, zr[j], acc[t][e][j]);
        }
      }
    }
  };

  {
    uint4 raw0[UN], raw1[UN];
    unsigned int by0[NT][UN], by1[NT][UN];
    const int first = r_lo + rl;
    issue(raw0, by0, first);
    for (int b0 = first; b0 < r_hi; b0 += 64 * UN) {
      const int b1 = b0 + 32 * UN, b2 = b0 + 64 * UN;
      issue(raw1, by1, b1);                                          // rows past the split load nothing (ok == false)
      __builtin_amdgcn_sched_barrier(0);
      process(raw0, by0, b0);
      issue(raw0, by0, b2);
      __builtin_amdgcn_sched_barrier(0);
      if (b1 < r_hi) process(raw1, by1, b1);
    }
  }

  // the 8 row lanes of a wave that share a column chunk (lane bits 3..5) are added with a halving exchange (permlane32_swap,
  // permlane16_swap, DPP row_ror:8 - no LDS): afterwards lane L holds the sums v in [PER/8 * (L >> 3), PER/8 * (L >> 3) + PER/8)
  // of its chunk (v = (t * 8 + e) * RANK + j).  Then the 4 waves through LDS, fixed order.
  constexpr int PER = NT * 8 * RANK;                                 // sums per column chunk
  float (&av)[PER] = reinterpret_cast<float (&)[PER]>(acc);
  fold_rowlanes<PER>(av, lane);
  __syncthreads();                                                   // the z stage is read out: smem becomes red[4][8][PER]
  {
    float* dst = smem + (wave * 8 + chunk) * PER + (PER / 8) * (lane >> 3);
#pragma unroll
    for (int k = 0; k < PER / 8; k += 4) *reinterpret_cast<float4*>(dst + k) = make_float4(av[k], av[k + 1], av[k + 2], av[k + 3]);
  }
  __syncthreads();
  // value index v in [0, NT * 64 * RANK): term t = v / (64 RANK), column cl = (v / RANK) % 64, rank j = v % RANK; handled in pairs
  constexpr int NV = NT * 64 * RANK;
  const int64_t half = static_cast<int64_t>(C) * RANK / 2;         // pairs per (term, split)
  const bool direct = S == 1;
  for (int v = tid * 2; v < NV; v += 512) {
    const int t = v / (64 * RANK), cl = (v / RANK) & 63, j = v % RANK;
    const int ch = cl >> 3, e = cl & 7;
    float s2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int off = (t * 8 + e) * RANK + j + h;
      s2[h] = (smem[(0 * 8 + ch) * PER + off] + smem[(1 * 8 + ch) * PER + off]) +
              (smem[(2 * 8 + ch) * PER + off] + smem[(3 * 8 + ch) * PER + off]);
    }
    const int c = static_cast<int>(blockIdx.x) * 64 + cl;
    if (c >= C) continue;
    const int q = TWO_X ? prob : t;
    if (direct) {
      float* o = q ? out1 : out0;
      o[static_cast<int64_t>(j) * C + c] = scale * s2[0];
      o[static_cast<int64_t>(j + 1) * C + c] = scale * s2[1];
    } else {
      unsigned long long* dst = part + (static_cast<int64_t>(q) * S + blockIdx.y) * half + (static_cast<int64_t>(c) * RANK + j) / 2;
      st_pair(dst, s2[0], s2[1]);
    }
  }
  if (direct) return;
  unsigned int* tk = tickets + (TWO_X ? prob * gridDim.x : 0) + blockIdx.x;
  if (draw_ticket(tk, &slot) != static_cast<unsigned int>(S - 1)) return;
  if (tid == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int v = tid * 2; v < NV; v += 512) {
    const int t = v / (64 * RANK), cl = (v / RANK) & 63, j = v % RANK;
    const int c = static_cast<int>(blockIdx.x) * 64 + cl;
    if (c >= C) continue;
    const int q = TWO_X ? prob : t;
    const unsigned long long* src = part + static_cast<int64_t>(q) * S * half + (static_cast<int64_t>(c) * RANK + j) / 2;
    float a = 0.f, b = 0.f;
    for (int s0 = 0; s0 < S; s0 += 8) {                            // 8 loads in flight; added in split order (fixed)
      float va[8], vb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) ld_pair(src + static_cast<int64_t>(min(s0 + u, S - 1)) * half, va[u], vb[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a += (s0 + u < S) ? va[u] : 0.f;
        b += (s0 + u < S) ? vb[u] : 0.f;
      }
    }
    float* o = q ? out1 : out0;
    o[static_cast<int64_t>(j) * C + c] = scale * a;
    o[static_cast<int64_t>(j + 1) * C + c] = scale * b;
  }
}

// ---- launch geometry ----
struct ColaccGeom { int rows_per_split, S, slabs; size_t lds; int bits_off; };
inline ColaccGeom colacc_geom(int64_t R, int64_t C, int rank, int nt, int nprob) {
  ColaccGeom g;
  g.slabs = static_cast<int>((C + 63) / 64);
  int64_t S = (512 + static_cast<int64_t>(g.slabs) * nprob - 1) / (static_cast<int64_t>(g.slabs) * nprob);
  const int64_t max_s = (R + 63) / 64;
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  int64_t rows = (R + S - 1) / S;
  const int64_t cap = 16384 / (static_cast<int64_t>(nt) * rank);      // z stage <= 64 KB
  if (rows > cap) rows = cap;
  rows = (rows + 31) / 32 * 32;
  g.rows_per_split = static_cast<int>(rows);
  g.S = static_cast<int>((R + rows - 1) / rows);
  const size_t stage = static_cast<size_t>(nt) * rows * rank * sizeof(float);
  const size_t red = static_cast<size_t>(4) * 8 * nt * 8 * rank * sizeof(float);
  g.lds = stage > red ? stage : red;
  g.bits_off = static_cast<int>(g.lds);
  g.lds += static_cast<size_t>(nt) * rows * 8;                          // the mask stage (used when the call carries masks)
  return g;
}

}  // namespace
}  // namespace dalm

using namespace dalm;

#define DALM_LORA2_SHAPE(R, C, rank)                                                                                   \
  DALM_REQUIRE((rank) == 8 || (rank) == 16, DALM_E_SHAPE, "rank must be 8 or 16");                                     \
  DALM_REQUIRE((R) > 0 && (C) > 0 && (C) % 8 == 0 && (R) <= 0x7fffffffll && (C) <= 0x7fffffffll, DALM_E_SHAPE,         \
               "need rows > 0 and a positive column count that is a multiple of 8");                                   \
  DALM_REQUIRE(mode >= 1 && mode <= 3, DALM_E_SHAPE, "mode must be 1 (one problem), 2 (two terms, shared activation) or 3 (two problems)"); \
  DALM_REQUIRE(mode != 2 || (rank) == 8, DALM_E_SHAPE, "stacked projections need rank 8")

extern "C" int dalm_lora2_rowdot(const void* x0, const void* x1, const float* W0, const float* W1, float* out0, float* out1,
                                 void* bits0, void* bits1, int64_t R, int64_t K, int rank, float scale, float p,
                                 const void* seed, uint32_t salt0, uint32_t salt1, int mode, dalm_stream_t stream) {
  DALM_LORA2_SHAPE(R, K, rank);
  DALM_REQUIRE(K % 32 == 0, DALM_E_SHAPE, "the contraction length must be a multiple of 32");
  DALM_REQUIRE(p >= 0.f && p < 1.f, DALM_E_SHAPE, "dropout probability must be in [0, 1)");
  const bool two = mode != 1;
  DALM_REQUIRE(x0 && W0 && out0 && (!two || (W1 && out1)) && (mode != 3 || x1), DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(al16(x0) && al16(W0) && (!two || al16(W1)) && (mode != 3 || al16(x1)), DALM_E_ALIGN, "x / W must be 16-byte aligned");
  const bool drop = p > 0.f;
  DALM_REQUIRE(!drop || (seed && bits0 && (!two || bits1)), DALM_E_NULL, "dropout needs the seed word and the mask buffers");
  const unsigned int thr16 = static_cast<unsigned int>(p * 65536.0f + 0.5f);
  DALM_REQUIRE(!drop || thr16 >= 1, DALM_E_SHAPE, "dropout probability below 2^-17 rounds to no dropout: pass 0");
  const bf16_t* xa0 = static_cast<const bf16_t*>(x0);
  const bf16_t* xa1 = static_cast<const bf16_t*>(mode == 3 ? x1 : x0);
  const float* Wa1 = two ? W1 : W0;
  float* oa1 = two ? out1 : out0;
  unsigned char* ba0 = static_cast<unsigned char*>(bits0);
  unsigned char* ba1 = static_cast<unsigned char*>(two ? bits1 : bits0);
  const unsigned long long* sd = static_cast<const unsigned long long*>(seed);
  const int Ri = static_cast<int>(R), Ki = static_cast<int>(K);
  hipStream_t s = as_stream(stream);
  const unsigned tiles16 = static_cast<unsigned>((R + 15) / 16), tiles8 = static_cast<unsigned>((R + 7) / 8);
#define DALM_RD2(MODE, DR, GRID) hipLaunchKernelGGL((lora2_rowdot_kernel<MODE, DR>), GRID, dim3(256), 0, s, xa0, xa1, W0, Wa1, \
    out0, oa1, ba0, ba1, sd, salt0, salt1, thr16, Ri, Ki, rank, scale)
  if (mode == 1) { if (drop) DALM_RD2(1, true, dim3(tiles16)); else DALM_RD2(1, false, dim3(tiles16)); }
  else if (mode == 2) { if (drop) DALM_RD2(2, true, dim3(tiles8)); else DALM_RD2(2, false, dim3(tiles8)); }
  else { if (drop) DALM_RD2(3, true, dim3(tiles16, 2)); else DALM_RD2(3, false, dim3(tiles16, 2)); }
#undef DALM_RD2
  return check_launch(__func__);
}

extern "C" int dalm_lora2_rankupd(void* y0, void* y1, const float* z0, const float* z1, const float* W0, const float* W1,
                                  const void* bits0, const void* bits1, int64_t R, int64_t C, int rank, float scale, int mode,
                                  dalm_stream_t stream) {
  DALM_LORA2_SHAPE(R, C, rank);
  const bool two = mode != 1;
  DALM_REQUIRE(y0 && z0 && W0 && (!two || (z1 && W1)) && (mode != 3 || y1), DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(al16(y0) && al16(W0) && (!two || al16(W1)) && (mode != 3 || al16(y1)), DALM_E_ALIGN, "y / W must be 16-byte aligned");
  const bool bits = bits0 != nullptr;
  DALM_REQUIRE(!two || ((bits1 != nullptr) == bits), DALM_E_NULL, "either both terms carry a mask or neither");
  bf16_t* ya0 = static_cast<bf16_t*>(y0);
  bf16_t* ya1 = static_cast<bf16_t*>(mode == 3 ? y1 : y0);
  const float* za1 = two ? z1 : z0;
  const float* Wa1 = two ? W1 : W0;
  const unsigned char* ba0 = static_cast<const unsigned char*>(bits0);
  const unsigned char* ba1 = static_cast<const unsigned char*>(two ? bits1 : bits0);
  const int Ri = static_cast<int>(R), Ci = static_cast<int>(C);
  const int64_t slabs = (C + 511) / 512, nprob = mode == 3 ? 2 : 1;
  // one resident round of workgroups (2 per CU with two terms' W slices in registers, 3 with one): every workgroup pays its W
  // loads and mask stage once (1536 workgroups of 24 rows measured no faster than round 4's kernel)
  int64_t S = (mode == 2 ? 512 : 768) / (slabs * nprob);
  const int64_t max_s = (R + 3) / 4, min_s = (R + 255) / 256;       // at most 256 rows per workgroup (the LDS mask stage)
  if (S > max_s) S = max_s;
  if (S < min_s) S = min_s;
  if (S < 1) S = 1;
  int64_t rows = ((R + S - 1) / S + 3) / 4 * 4;
  const int rows_per_wg = static_cast<int>(rows);
  S = (R + rows - 1) / rows;
  DALM_REQUIRE(S <= 65535, DALM_E_SHAPE, "too many rows for one launch");
  const dim3 grid(static_cast<unsigned>(slabs), static_cast<unsigned>(S), static_cast<unsigned>(nprob));
  hipStream_t s = as_stream(stream);
  const bool stage = bits && C % 128 == 0;
  const size_t lds = stage ? static_cast<size_t>(mode == 2 ? 2 : 1) * rows_per_wg * 64 : 0;
#define DALM_RU2(RK, NT, BT, TY, ST) hipLaunchKernelGGL((lora2_rankupd_kernel<RK, NT, BT, TY, ST>), grid, dim3(256), lds, s, ya0, ya1, \
    z0, za1, W0, Wa1, ba0, ba1, Ri, Ci, rows_per_wg, scale)
#define DALM_RU2_B(RK, NT, TY) do { if (stage) DALM_RU2(RK, NT, true, TY, true); else if (bits) DALM_RU2(RK, NT, true, TY, false); \
                                    else DALM_RU2(RK, NT, false, TY, false); } while (0)
  if (mode == 2) DALM_RU2_B(8, 2, false);
  else if (mode == 3) { if (rank == 8) DALM_RU2_B(8, 1, true); else DALM_RU2_B(16, 1, true); }
  else { if (rank == 8) DALM_RU2_B(8, 1, false); else DALM_RU2_B(16, 1, false); }
#undef DALM_RU2_B
#undef DALM_RU2
  return check_launch(__func__);
}

extern "C" size_t dalm_lora2_colacc_workspace_bytes(int64_t R, int64_t C, int rank, int mode) {
  if (R <= 0 || C <= 0 || rank <= 0 || mode < 1 || mode > 3) return 0;
  const ColaccGeom g = colacc_geom(R, C, rank, mode == 2 ? 2 : 1, mode == 3 ? 2 : 1);
  return static_cast<size_t>(2) * g.S * C * rank * sizeof(float);
}
extern "C" size_t dalm_lora2_colacc_ticket_words(int64_t C, int mode) {
  if (C <= 0) return 0;
  return static_cast<size_t>((C + 63) / 64) * (mode == 3 ? 2 : 1);
}

extern "C" int dalm_lora2_colacc(const void* x0, const void* x1, const float* z0, const float* z1, const void* bits0,
                                 const void* bits1, float* out0, float* out1, int64_t R, int64_t C, int rank, float scale,
                                 int mode, void* ws, size_t ws_bytes, uint32_t* tickets, dalm_stream_t stream) {
  DALM_LORA2_SHAPE(R, C, rank);
  const bool two = mode != 1;
  DALM_REQUIRE(x0 && z0 && out0 && ws && tickets && (!two || (z1 && out1)) && (mode != 3 || x1), DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(al16(x0) && al16(z0) && (!two || al16(z1)) && (mode != 3 || al16(x1)) && (reinterpret_cast<uintptr_t>(ws) & 7) == 0,
               DALM_E_ALIGN, "x / z must be 16-byte aligned, the workspace 8-byte aligned");
  DALM_REQUIRE(ws_bytes >= dalm_lora2_colacc_workspace_bytes(R, C, rank, mode), DALM_E_WORKSPACE, "workspace too small");
  const bool bits = bits0 != nullptr;
  DALM_REQUIRE(!two || ((bits1 != nullptr) == bits), DALM_E_NULL, "either both terms carry a mask or neither");
  const ColaccGeom g = colacc_geom(R, C, rank, mode == 2 ? 2 : 1, mode == 3 ? 2 : 1);
  DALM_REQUIRE(g.S <= 65535, DALM_E_SHAPE, "too many rows for one launch");
  const bf16_t* xa0 = static_cast<const bf16_t*>(x0);
  const bf16_t* xa1 = static_cast<const bf16_t*>(mode == 3 ? x1 : x0);
  const float* za1 = two ? z1 : z0;
  const unsigned char* ba0 = static_cast<const unsigned char*>(bits0);
  const unsigned char* ba1 = static_cast<const unsigned char*>(two ? bits1 : bits0);
  float* oa1 = two ? out1 : out0;
  unsigned long long* part = static_cast<unsigned long long*>(ws);
  const int Ri = static_cast<int>(R), Ci = static_cast<int>(C);
  const dim3 grid(static_cast<unsigned>(g.slabs), static_cast<unsigned>(g.S), mode == 3 ? 2u : 1u);
  hipStream_t s = as_stream(stream);
  const bool stage = bits && C % 64 == 0;
#define DALM_CA2(RK, NT, BT, TX, ST) hipLaunchKernelGGL((lora2_colacc_kernel<RK, NT, BT, TX, ST>), grid, dim3(256), g.lds, s, xa0, xa1, \
    z0, za1, ba0, ba1, out0, oa1, part, tickets, Ri, Ci, g.rows_per_split, g.S, scale, g.bits_off)
#define DALM_CA2_B(RK, NT, TX) do { if (stage) DALM_CA2(RK, NT, true, TX, true); else if (bits) DALM_CA2(RK, NT, true, TX, false); \
                                    else DALM_CA2(RK, NT, false, TX, false); } while (0)
  if (mode == 2) DALM_CA2_B(8, 2, false);
  else if (mode == 3) { if (rank == 8) DALM_CA2_B(8, 1, true); else DALM_CA2_B(16, 1, true); }
  else { if (rank == 8) DALM_CA2_B(8, 1, false); else DALM_CA2_B(16, 1, false); }
#undef DALM_CA2_B
#undef DALM_CA2
  return check_launch(__func__);
}
