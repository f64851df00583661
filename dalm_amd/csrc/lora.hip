// dalm_lora_*: the low-rank branch of a LoRA-wrapped Linear as streaming kernels (gfx950).
//
// The reference asks peft for r = 8, alpha = 16, dropout 0.05 adapters on q_proj / v_proj (key / query / value for BERT
// retrievers): dalm/models/rag_e2e_base_model.py:145-160.  peft evaluates  out = W x + s * B(A(dropout(x)))  as eager ops; in
// the cfg3 step each wrapped projection cost 82 us forward (dropout kernel, two weight casts, two skinny GEMMs, an add) and
// 132 us backward (a full-size scale, four skinny GEMMs, two casts, the dropout backward, a gradient add) around a 128 us
// base GEMM - 13.7 ms of a 160 ms step (profiles/history/r04_step_by_stream.txt).  With r = 8 every one of those tensors is either
// [rows, 8] or streams the [rows, K] activation once, so the branch is three HBM-bound kernels:
//   rowdot :  z[row, j]  = scale * sum_k m x[row, k] W(j, k)            forward z = dropout(x) A^T / (1-p);  backward dz = s g B
//   rankupd:  y[row, c] += scale * m * sum_j z[row, j] W(j, c)          forward out += s z B^T;  backward dx += m (dz A) / (1-p)
//   colacc :  o(j, c)    = scale * sum_row m x[row, c] z[row, j]         backward dB = s g^T z,  dA = dz^T (m x) / (1-p)
// A, B, z, dz, dA, dB stay in f32 (no autocast casts); x, g, out, dx are the tensors' own dtype (bf16 or f32).
//
// Dropout: m is a keep mask that is never stored.  It is a counter-based hash of (seed word read from device memory, salt,
// flat element index of the [rows, K] activation), recomputed identically by the three kernels that need it.  The seed word
// lives in device memory so that a hipGraph replay of the step sees a new mask whenever the step advanced it.
// Algorithmic bytes per call: rowdot R*K*el, rankupd 2*R*C*el, colacc R*C*el (+ O(R*r) and O(K*r) terms).
#include "lora_common.hpp"

namespace dalm {
namespace {
using namespace lora;

// keep bits (bit e = element e0 + e survives) for the 8 elements starting at flat index e0 (a multiple of 8)
__device__ __forceinline__ unsigned int keep8(const DropKey& k, unsigned int e0, unsigned int thr16) {
  unsigned int bits = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned int h = lowbias32(lowbias32(((e0 >> 1) + i) ^ k.a) + k.b);
    bits |= ((h & 0xffffu) >= thr16 ? 1u : 0u) << (2 * i);
    bits |= ((h >> 16) >= thr16 ? 1u : 0u) << (2 * i + 1);
  }
  return bits;
}

// v[RANK] per lane -> lane L < RANK holds the wave-wide sum of v[slot(L)], slot = the bit reversal of L over log2(RANK) bits
template <int RANK>
__device__ __forceinline__ float fold_rank(float (&v)[RANK], int lane) {
#pragma unroll
  for (int half = RANK / 2, s = 1; half >= 1; half >>= 1, s <<= 1) {
    const bool up = (lane & s) != 0;
#pragma unroll
    for (int i = 0; i < half; ++i) {
      const float send = up ? v[i] : v[i + half];
      const float keep = up ? v[i + half] : v[i];
      v[i] = keep + __shfl_xor(send, s, 64);
    }
  }
  float r = v[0];
#pragma unroll
  for (int s = RANK; s < 64; s <<= 1) r += __shfl_xor(r, s, 64);
  return r;
}
template <int RANK> __device__ __forceinline__ int fold_slot(int lane) {   // which j lane `lane` < RANK ends up holding
  int j = 0;
#pragma unroll
  for (int b = 0, n = RANK; n > 1; n >>= 1, ++b) j |= ((lane >> b) & 1) * (n >> 1);
  return j;
}

// rowdot / rankupd: rows in flight per thread (their raw chunks, 4-8 registers each, share the register file with the thread's
// RANK x 8 slice of W); a workgroup walks `rows_per_wg` rows in batches of this size
template <typename T, int RANK> struct BatchFor { static constexpr int value = (RANK == 8 ? 8 : 4) / (sizeof(T) == 4 ? 2 : 1); };
constexpr int kSpan = 512 * 8;   // columns one pass of a 512-thread workgroup covers

// W(j, k) for the 8 columns k0..k0+7: KMAJOR = true reads W[j][K] (lora_A), false reads W[K][RANK] (lora_B)
template <int RANK, bool KMAJOR>
__device__ __forceinline__ void load_w(const float* __restrict__ W, int K, int k0, bool valid, float (&w)[RANK][8]) {
#pragma unroll
  for (int j = 0; j < RANK; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) w[j][e] = 0.f;
  if (!valid) return;
  if constexpr (KMAJOR) {
#pragma unroll
    for (int j = 0; j < RANK; ++j) {
      const float4 a = *reinterpret_cast<const float4*>(W + static_cast<int64_t>(j) * K + k0);
      const float4 b = *reinterpret_cast<const float4*>(W + static_cast<int64_t>(j) * K + k0 + 4);
      w[j][0] = a.x; w[j][1] = a.y; w[j][2] = a.z; w[j][3] = a.w; w[j][4] = b.x; w[j][5] = b.y; w[j][6] = b.z; w[j][7] = b.w;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int j4 = 0; j4 < RANK; j4 += 4) {
        const float4 a = *reinterpret_cast<const float4*>(W + static_cast<int64_t>(k0 + e) * RANK + j4);
        w[j4][e] = a.x; w[j4 + 1][e] = a.y; w[j4 + 2][e] = a.z; w[j4 + 3][e] = a.w;
      }
  }
}

// ---------------------------------------------------------------------------------------------------
// rowdot: out[row][j] = scale * sum_k m x[row][k] W(j, k).  512 threads; thread t owns columns t*8 + 4096*i and keeps its
// slice of W in registers (loaded once when K <= 4096) while the workgroup walks its rows, BATCH rows in flight.
// ---------------------------------------------------------------------------------------------------
template <typename T, int RANK, bool DROP, bool KMAJOR>
__global__ __launch_bounds__(512) void lora_rowdot_kernel(const T* __restrict__ x, const float* __restrict__ W,
                                                          float* __restrict__ out, int R, int K, int rows_per_wg, float scale,
                                                          DropArgs drop) {
  constexpr int BATCH = BatchFor<T, RANK>::value;
  __shared__ float red[8][BATCH][RANK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row_lo = blockIdx.x * rows_per_wg, row_hi = min(R, row_lo + rows_per_wg);
  const int nspan = (K + kSpan - 1) / kSpan;
  DropKey key{0u, 1u};
  if constexpr (DROP) key = drop_key(drop);
  float w[RANK][8];
  if (nspan == 1) load_w<RANK, KMAJOR>(W, K, tid * 8, tid * 8 < K, w);
  for (int b0 = row_lo; b0 < row_hi; b0 += BATCH) {
    float racc[BATCH];
#pragma unroll
    for (int r = 0; r < BATCH; ++r) racc[r] = 0.f;
    for (int sp = 0; sp < nspan; ++sp) {
      const int k0 = sp * kSpan + tid * 8;
      const bool valid = k0 < K;
      if (nspan > 1) load_w<RANK, KMAJOR>(W, K, k0, valid, w);
      typename Chunk8<T>::Raw raw[BATCH];
#pragma unroll
      for (int r = 0; r < BATCH; ++r) {
        const int row = min(b0 + r, R - 1);
        raw[r] = valid ? Chunk8<T>::load_raw(x + static_cast<int64_t>(row) * K + k0) : Chunk8<T>::zero_raw();
      }
#pragma unroll
      for (int r = 0; r < BATCH; ++r) {
        float xv[8];
        Chunk8<T>::decode(raw[r], xv);
        if constexpr (DROP) {
          const int row = min(b0 + r, R - 1);
          const unsigned int bits = keep8(key, static_cast<unsigned int>(row) * static_cast<unsigned int>(K) + k0, drop.thr16);
#pragma unroll
          for (int e = 0; e < 8; ++e) xv[e] = (bits >> e) & 1u ? xv[e] : 0.f;
        }
        float part[RANK];
#pragma unroll
        for (int j = 0; j < RANK; ++j) {
          float sacc = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) sacc = fmaf(xv[e], w[j][e], sacc);
          part[j] = sacc;
        }
        racc[r] += fold_rank<RANK>(part, lane);
      }
    }
    if (lane < RANK) {
      const int j = fold_slot<RANK>(lane);
#pragma unroll
      for (int r = 0; r < BATCH; ++r) red[wave][r][j] = racc[r];
    }
    __syncthreads();
    if (tid < BATCH * RANK) {
      const int r = tid / RANK, j = tid % RANK;
      float sacc = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) sacc += red[w8][r][j];   // fixed order
      if (b0 + r < row_hi) out[static_cast<int64_t>(b0 + r) * RANK + j] = scale * sacc;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// rowdot for bf16 activations on the matrix cores (v_mfma_f32_16x16x32_bf16): the VALU kernel above spends ~20 instructions per
// element (8 FMAs, the unpack, the mask hash, the cross-lane fold) with one 8-wave workgroup per CU - 36-42 us for the 37.7 MB
// of a cfg3 projection input.  Here a workgroup owns 16 rows; wave w takes the 32-column steps w, w+8, ...: lane L supplies
// x[row L%16][8 columns of group L/16] as the A operand (a 16-byte load; 4 lanes cover 64 contiguous bytes of a row) and
// W(j = L%16, the same 8 columns) as the B operand, split into bf16 high and low parts (two MFMAs) so that W keeps f32
// accuracy; the fold over columns happens in the accumulator.  The 8 waves' 16x16 partial tiles are added through LDS in
// wave order.  Needs K % 32 == 0.
// ---------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_bf16_pair(float a, float b, unsigned int& hi, unsigned int& lo) {
  hi = pack_bf16x2(a, b);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  lo = pack_bf16x2(ra, rb);
}

template <int RANK, bool DROP, bool KMAJOR>
__global__ __launch_bounds__(512) void lora_rowdot_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ W,
                                                               float* __restrict__ out, int R, int K, float scale,
                                                               DropArgs drop) {
  __shared__ float red[8][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kg = lane >> 4;
  const int row = min(static_cast<int>(blockIdx.x) * 16 + i, R - 1);
  const bf16_t* xr = x + static_cast<int64_t>(row) * K + kg * 8;
  const bool wv = i < RANK;                    // this lane's B-operand column is a real rank index
  DropKey key{0u, 1u};
  if constexpr (DROP) key = drop_key(drop);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nsteps = K >> 5;
  constexpr int UN = 4;
  for (int s0 = wave; s0 < nsteps; s0 += 8 * UN) {
    uint4 xa[UN];
    float wf[UN][8];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int st = min(s0 + 8 * u, nsteps - 1);     // clamped: the extra steps are skipped below
      const int k = st * 32;
      xa[u] = *reinterpret_cast<const uint4*>(xr + k);
      if (wv) {
        const int kk = k + kg * 8;
        if constexpr (KMAJOR) {
          const float4 a = *reinterpret_cast<const float4*>(W + static_cast<int64_t>(i) * K + kk);
          const float4 b = *reinterpret_cast<const float4*>(W + static_cast<int64_t>(i) * K + kk + 4);
          wf[u][0] = a.x; wf[u][1] = a.y; wf[u][2] = a.z; wf[u][3] = a.w;
          wf[u][4] = b.x; wf[u][5] = b.y; wf[u][6] = b.z; wf[u][7] = b.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) wf[u][e] = W[static_cast<int64_t>(kk + e) * RANK + i];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) wf[u][e] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (s0 + 8 * u >= nsteps) continue;      // wave-uniform
      uint4 v = xa[u];
      if constexpr (DROP) {
        const unsigned int e0 = static_cast<unsigned int>(row) * static_cast<unsigned int>(K) + (s0 + 8 * u) * 32 + kg * 8;
        const unsigned int bits = keep8(key, e0, drop.thr16);
        unsigned int wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned int m = ((bits >> (2 * q)) & 1u ? 0x0000ffffu : 0u) | ((bits >> (2 * q + 1)) & 1u ? 0xffff0000u : 0u);
          wds[q] &= m;
        }
        v = make_uint4(wds[0], wds[1], wds[2], wds[3]);
      }
      unsigned int hi[4], lo[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) split_bf16_pair(wf[u][2 * q], wf[u][2 * q + 1], hi[q], lo[q]);
      const uint4 bh = make_uint4(hi[0], hi[1], hi[2], hi[3]), bl = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      const bf16x8 a8 = __builtin_bit_cast(bf16x8, v);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, __builtin_bit_cast(bf16x8, bl), acc, 0, 0, 0);   // small part first
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
    }
  }
  // accumulator tile: lane L holds rows 4 (L / 16) + r, column L % 16
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][kg * 4 + r][i] = acc[r];
  __syncthreads();
  if (tid < 16 * RANK) {
    const int m = tid / RANK, j = tid % RANK;
    float sacc = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) sacc += red[w8][m][j];   // fixed order
    const int orow = static_cast<int>(blockIdx.x) * 16 + m;
    if (orow < R) out[static_cast<int64_t>(orow) * RANK + j] = scale * sacc;
  }
}

// ---------------------------------------------------------------------------------------------------
// rankupd: y[row][c] += scale * m * sum_j z[row][j] W(j, c), in place.  Same thread <-> column mapping as rowdot.
// CMAJOR = true reads W[C][RANK] (lora_B: forward), false reads W[RANK][C] (lora_A: backward, with the dropout mask).
// ---------------------------------------------------------------------------------------------------
template <typename T, int RANK, bool DROP, bool CMAJOR>
__global__ __launch_bounds__(512) void lora_rankupd_kernel(T* __restrict__ y, const float* __restrict__ z,
                                                           const float* __restrict__ W, int R, int C, int rows_per_wg,
                                                           float scale, DropArgs drop) {
  // Narrow tensors (C < 4096: the 1024-wide BERT projections) would leave 3/4 of the 512 threads without a column chunk:
  // there the threads form G = 512 / (C/8) row groups that walk interleaved rows (measured before this: 37.9 us for the
  // 2 x 39 MB of a [19200, 1024] bf16 update).
  constexpr int BATCH = BatchFor<T, RANK>::value;
  const int tid = threadIdx.x;
  const int row_lo = blockIdx.x * rows_per_wg, row_hi = min(R, row_lo + rows_per_wg);
  const int chunks = (C + 7) >> 3;
  const int cpr = min(512, chunks);                     // threads per row
  const int G = 512 / cpr;                              // row groups per workgroup
  const int chunk = tid % cpr, rg = tid / cpr;
  if (rg >= G) return;                                  // 512 % cpr leftover threads (no barrier in this kernel)
  const int span = cpr * 8;
  const int nspan = (C + span - 1) / span;
  DropKey key{0u, 1u};
  if constexpr (DROP) key = drop_key(drop);
  float w[RANK][8];
  if (nspan == 1) load_w<RANK, !CMAJOR>(W, C, chunk * 8, chunk * 8 < C, w);
  for (int b0 = row_lo + rg; b0 < row_hi; b0 += BATCH * G) {
    for (int sp = 0; sp < nspan; ++sp) {
      const int c0 = sp * span + chunk * 8;
      if (c0 >= C) continue;
      if (nspan > 1) load_w<RANK, !CMAJOR>(W, C, c0, true, w);
      typename Chunk8<T>::Raw raw[BATCH];
#pragma unroll
      for (int r = 0; r < BATCH; ++r) raw[r] = Chunk8<T>::load_raw(y + static_cast<int64_t>(min(b0 + r * G, R - 1)) * C + c0);
#pragma unroll
      for (int r = 0; r < BATCH; ++r) {
        const int row = b0 + r * G;
        if (row >= row_hi) continue;
        float yv[8];
        Chunk8<T>::decode(raw[r], yv);
        float zr[RANK];
#pragma unroll
        for (int j4 = 0; j4 < RANK; j4 += 4) {
          const float4 a = *reinterpret_cast<const float4*>(z + static_cast<int64_t>(row) * RANK + j4);
          zr[j4] = a.x; zr[j4 + 1] = a.y; zr[j4 + 2] = a.z; zr[j4 + 3] = a.w;
        }
        unsigned int bits = 0xffu;
        if constexpr (DROP) bits = keep8(key, static_cast<unsigned int>(row) * static_cast<unsigned int>(C) + c0, drop.thr16);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float u = 0.f;
#pragma unroll
          for (int j = 0; j < RANK; ++j) u = fmaf(zr[j], w[j][e], u);
          yv[e] = (bits >> e) & 1u ? fmaf(scale, u, yv[e]) : yv[e];
        }
        Chunk8<T>::store(y + static_cast<int64_t>(row) * C + c0, yv);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// colacc: part[split][c][j] = sum over the split's rows of m x[row][c] z[row][j].  grid (C/64 slabs, row splits), 256 threads =
// 8 column chunks x 32 row lanes; every thread keeps an 8 x RANK block of sums in registers.  lora_colacc_reduce_kernel adds the
// splits in fixed order, scales and writes [C][RANK] or [RANK][C].
// ---------------------------------------------------------------------------------------------------
template <typename T, int RANK, bool DROP>
__global__ __launch_bounds__(256) void lora_colacc_kernel(const T* __restrict__ x, const float* __restrict__ z,
                                                          float* __restrict__ part, int R, int C, int rows_per_split,
                                                          DropArgs drop) {
  __shared__ float red[4][8][8 * RANK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = tid & 7, rl = tid >> 3;
  const int c0 = blockIdx.x * 64 + chunk * 8;
  const bool valid = c0 < C;
  const int r_lo = blockIdx.y * rows_per_split, r_hi = min(R, r_lo + rows_per_split);
  DropKey key{0u, 1u};
  if constexpr (DROP) key = drop_key(drop);
  float acc[8][RANK];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j < RANK; ++j) acc[e][j] = 0.f;
  constexpr int UN = 4;
  for (int rb = r_lo + rl; rb < r_hi; rb += 32 * UN) {
    float xv[UN][8], zr[UN][RANK];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int row = rb + 32 * u;
      const bool ok = valid && row < r_hi;
      const int rr = min(row, R - 1);
      if (ok) Chunk8<T>::load(x + static_cast<int64_t>(rr) * C + c0, xv[u]);
#pragma unroll
      for (int j4 = 0; j4 < RANK; j4 += 4) {
        const float4 a = *reinterpret_cast<const float4*>(z + static_cast<int64_t>(rr) * RANK + j4);
        zr[u][j4] = a.x; zr[u][j4 + 1] = a.y; zr[u][j4 + 2] = a.z; zr[u][j4 + 3] = a.w;
      }
      if (!ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[u][e] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if constexpr (DROP) {
        const int rr = min(rb + 32 * u, R - 1);
        const unsigned int bits = keep8(key, static_cast<unsigned int>(rr) * static_cast<unsigned int>(C) + c0, drop.thr16);
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[u][e] = (bits >> e) & 1u ? xv[u][e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int j = 0; j < RANK; ++j) acc[e][j] = fmaf(xv[u][e], zr[u][j], acc[e][j]);
    }
  }
  // the 8 row lanes of a wave that share a column chunk (lane bits 3..5), then the 4 waves through LDS: fixed order
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j < RANK; ++j) {
      float v = acc[e][j];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      acc[e][j] = v;
    }
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int j = 0; j < RANK; ++j) red[wave][lane][e * RANK + j] = acc[e][j];
  }
  __syncthreads();
  for (int i = tid; i < 8 * 8 * RANK; i += 256) {
    const int ch = i / (8 * RANK), rest = i % (8 * RANK);
    const float s = (red[0][ch][rest] + red[1][ch][rest]) + (red[2][ch][rest] + red[3][ch][rest]);
    const int c = blockIdx.x * 64 + ch * 8 + rest / RANK;
    if (c < C) part[(static_cast<int64_t>(blockIdx.y) * C + c) * RANK + rest % RANK] = s;
  }
}

template <int RANK>
__global__ __launch_bounds__(256) void lora_colacc_reduce_kernel(const float* __restrict__ part, int splits, int C, float scale,
                                                                 int out_jmajor, float* __restrict__ out) {
  const int64_t n = static_cast<int64_t>(C) * RANK;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int q0 = 0; q0 < splits; q0 += 8) {          // 8 loads in flight; added in split order (fixed)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[static_cast<int64_t>(min(q0 + u, splits - 1)) * n + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (q0 + u < splits) ? v[u] : 0.f;
  }
  const int c = static_cast<int>(i / RANK), j = static_cast<int>(i % RANK);
  out[out_jmajor ? static_cast<int64_t>(j) * C + c : i] = scale * s;
}

// colacc: rows per split.  About 512 workgroups (slabs x splits) keep the chip busy: the generator's 64 slabs of a 4096-wide
// activation take 8 splits of 576 rows at cfg3; a 1024-wide BERT activation (16 slabs) takes 32 splits of its 2304 rows
// instead of 4 (64 workgroups measured 24 us for 4.7 MB)
inline int colacc_rows_per_split(int64_t R, int64_t C) {
  const int64_t slabs = (C + 63) / 64;
  int64_t splits = (512 + slabs - 1) / slabs;
  const int64_t max_splits = (R + 63) / 64;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int64_t rows = (R + splits - 1) / splits;
  rows = (rows + 31) / 32 * 32;
  return static_cast<int>(rows);
}
// rowdot / rankupd: one 512-thread workgroup per CU at a time (its W slice and rows in flight fill the register file), so the
// row count per workgroup is chosen to make the grid a whole number of 256-workgroup rounds (R = 4608: 256 x 18 rows)
inline int lora_rows_per_wg(int64_t R, int64_t row_groups = 1) {     // row_groups: rankupd on narrow tensors (see the kernel)
  const int64_t rounds = (R + 256 * 24 * row_groups - 1) / (256 * 24 * row_groups);
  const int64_t n_wg = 256 * rounds;
  return static_cast<int>((R + n_wg - 1) / n_wg);
}

}  // namespace
}  // namespace dalm

using namespace dalm;

#define DALM_LORA_COMMON_CHECKS(R, C)                                                                              \
  DALM_REQUIRE(dtype == DALM_F32 || dtype == DALM_BF16, DALM_E_DTYPE, "dtype must be DALM_F32 or DALM_BF16");       \
  DALM_REQUIRE(rank == 8 || rank == 16, DALM_E_SHAPE, "rank must be 8 or 16");                                      \
  DALM_REQUIRE((R) > 0 && (C) > 0 && (C) % 8 == 0 && (R) <= 0x7fffffffll && (C) <= 0x7fffffffll, DALM_E_SHAPE,      \
               "need rows > 0 and a positive column count that is a multiple of 8");                                \
  DALM_REQUIRE(p >= 0.f && p < 1.f, DALM_E_SHAPE, "dropout probability must be in [0, 1)")

extern "C" int dalm_lora_rowdot(const void* x, int dtype, const float* W, int w_kmajor, int64_t R, int64_t K, int rank,
                                float scale, float p, const void* seed, uint32_t salt, float* out, dalm_stream_t stream) {
  DALM_REQUIRE(x && W && out, DALM_E_NULL, "null pointer argument");
  DALM_LORA_COMMON_CHECKS(R, K);
  DALM_REQUIRE(al16(x) && al16(W) && al16(out), DALM_E_ALIGN, "x / W / out must be 16-byte aligned");
  const DropArgs d = drop_args(p, seed, salt);
  hipStream_t s = as_stream(stream);
  static const bool valu_only = [] { const char* e = getenv("DALM_LORA_ROWDOT_VALU"); return e && atoi(e) != 0; }();
  if (dtype == DALM_BF16 && K % 32 == 0 && !valu_only) {
    const dim3 mgrid(static_cast<unsigned>((R + 15) / 16));
#define DALM_ROWDOT_M(RK, DR, KM) \
    hipLaunchKernelGGL((lora_rowdot_mfma_kernel<RK, DR, KM>), mgrid, dim3(512), 0, s, static_cast<const bf16_t*>(x), W, out, \
                       static_cast<int>(R), static_cast<int>(K), scale, d)
#define DALM_ROWDOT_M_KM(RK, DR) do { if (w_kmajor) DALM_ROWDOT_M(RK, DR, true); else DALM_ROWDOT_M(RK, DR, false); } while (0)
#define DALM_ROWDOT_M_DR(RK) do { if (p > 0.f) DALM_ROWDOT_M_KM(RK, true); else DALM_ROWDOT_M_KM(RK, false); } while (0)
    if (rank == 8) DALM_ROWDOT_M_DR(8); else DALM_ROWDOT_M_DR(16);
#undef DALM_ROWDOT_M_DR
#undef DALM_ROWDOT_M_KM
#undef DALM_ROWDOT_M
    return check_launch(__func__);
  }
  const int rows_per_wg = lora_rows_per_wg(R);
  const dim3 grid(static_cast<unsigned>((R + rows_per_wg - 1) / rows_per_wg));
#define DALM_ROWDOT(TT, RK, DR, KM) \
  hipLaunchKernelGGL((lora_rowdot_kernel<TT, RK, DR, KM>), grid, dim3(512), 0, s, static_cast<const TT*>(x), W, out, \
                     static_cast<int>(R), static_cast<int>(K), rows_per_wg, scale, d)
#define DALM_ROWDOT_KM(TT, RK, DR) do { if (w_kmajor) DALM_ROWDOT(TT, RK, DR, true); else DALM_ROWDOT(TT, RK, DR, false); } while (0)
#define DALM_ROWDOT_DR(TT, RK) do { if (p > 0.f) DALM_ROWDOT_KM(TT, RK, true); else DALM_ROWDOT_KM(TT, RK, false); } while (0)
#define DALM_ROWDOT_RK(TT) do { if (rank == 8) DALM_ROWDOT_DR(TT, 8); else DALM_ROWDOT_DR(TT, 16); } while (0)
  if (dtype == DALM_F32) DALM_ROWDOT_RK(float); else DALM_ROWDOT_RK(bf16_t);
#undef DALM_ROWDOT_RK
#undef DALM_ROWDOT_DR
#undef DALM_ROWDOT_KM
#undef DALM_ROWDOT
  return check_launch(__func__);
}

extern "C" int dalm_lora_rankupd(void* y, int dtype, const float* z, const float* W, int w_cmajor, int64_t R, int64_t C,
                                 int rank, float scale, float p, const void* seed, uint32_t salt, dalm_stream_t stream) {
  DALM_REQUIRE(y && z && W, DALM_E_NULL, "null pointer argument");
  DALM_LORA_COMMON_CHECKS(R, C);
  DALM_REQUIRE(al16(y) && al16(W) && al16(z), DALM_E_ALIGN, "y / z / W must be 16-byte aligned");
  const int64_t cpr = (C + 7) / 8 < 512 ? (C + 7) / 8 : 512;
  const int rows_per_wg = lora_rows_per_wg(R, 512 / cpr);
  const dim3 grid(static_cast<unsigned>((R + rows_per_wg - 1) / rows_per_wg));
  const DropArgs d = drop_args(p, seed, salt);
  hipStream_t s = as_stream(stream);
#define DALM_RANKUPD(TT, RK, DR, CM) \
  hipLaunchKernelGGL((lora_rankupd_kernel<TT, RK, DR, CM>), grid, dim3(512), 0, s, static_cast<TT*>(y), z, W, \
                     static_cast<int>(R), static_cast<int>(C), rows_per_wg, scale, d)
#define DALM_RANKUPD_CM(TT, RK, DR) do { if (w_cmajor) DALM_RANKUPD(TT, RK, DR, true); else DALM_RANKUPD(TT, RK, DR, false); } while (0)
#define DALM_RANKUPD_DR(TT, RK) do { if (p > 0.f) DALM_RANKUPD_CM(TT, RK, true); else DALM_RANKUPD_CM(TT, RK, false); } while (0)
#define DALM_RANKUPD_RK(TT) do { if (rank == 8) DALM_RANKUPD_DR(TT, 8); else DALM_RANKUPD_DR(TT, 16); } while (0)
  if (dtype == DALM_F32) DALM_RANKUPD_RK(float); else DALM_RANKUPD_RK(bf16_t);
#undef DALM_RANKUPD_RK
#undef DALM_RANKUPD_DR
#undef DALM_RANKUPD_CM
#undef DALM_RANKUPD
  return check_launch(__func__);
}

extern "C" size_t dalm_lora_colacc_workspace_bytes(int64_t R, int64_t C, int rank) {
  if (R <= 0 || C <= 0 || rank <= 0) return 0;
  const int64_t rps = colacc_rows_per_split(R, C);
  const int64_t splits = (R + rps - 1) / rps;
  return static_cast<size_t>(splits) * C * rank * sizeof(float);
}

extern "C" int dalm_lora_colacc(const void* x, int dtype, const float* z, int64_t R, int64_t C, int rank, float scale, float p,
                                const void* seed, uint32_t salt, float* out, int out_jmajor, void* ws, size_t ws_bytes,
                                dalm_stream_t stream) {
  DALM_REQUIRE(x && z && out && ws, DALM_E_NULL, "null pointer argument");
  DALM_LORA_COMMON_CHECKS(R, C);
  DALM_REQUIRE(al16(x) && al16(z), DALM_E_ALIGN, "x / z must be 16-byte aligned");
  DALM_REQUIRE(ws_bytes >= dalm_lora_colacc_workspace_bytes(R, C, rank), DALM_E_WORKSPACE, "workspace too small");
  const int rps = colacc_rows_per_split(R, C);
  const int64_t splits = (R + rps - 1) / rps;
  DALM_REQUIRE(splits <= 65535, DALM_E_SHAPE, "too many rows for one launch");
  const dim3 grid(static_cast<unsigned>((C + 63) / 64), static_cast<unsigned>(splits));
  const DropArgs d = drop_args(p, seed, salt);
  hipStream_t s = as_stream(stream);
  float* part = static_cast<float*>(ws);
#define DALM_COLACC(TT, RK, DR) \
  hipLaunchKernelGGL((lora_colacc_kernel<TT, RK, DR>), grid, dim3(256), 0, s, static_cast<const TT*>(x), z, part, \
                     static_cast<int>(R), static_cast<int>(C), rps, d)
#define DALM_COLACC_DR(TT, RK) do { if (p > 0.f) DALM_COLACC(TT, RK, true); else DALM_COLACC(TT, RK, false); } while (0)
#define DALM_COLACC_RK(TT) do { if (rank == 8) DALM_COLACC_DR(TT, 8); else DALM_COLACC_DR(TT, 16); } while (0)
  if (dtype == DALM_F32) DALM_COLACC_RK(float); else DALM_COLACC_RK(bf16_t);
#undef DALM_COLACC_RK
#undef DALM_COLACC_DR
#undef DALM_COLACC
  const unsigned rblocks = static_cast<unsigned>((C * rank + 255) / 256);
  if (rank == 8)
    hipLaunchKernelGGL(lora_colacc_reduce_kernel<8>, dim3(rblocks), dim3(256), 0, s, part, static_cast<int>(splits),
                       static_cast<int>(C), scale, out_jmajor, out);
  else
    hipLaunchKernelGGL(lora_colacc_reduce_kernel<16>, dim3(rblocks), dim3(256), 0, s, part, static_cast<int>(splits),
                       static_cast<int>(C), scale, out_jmajor, out);
  return check_launch(__func__);
}
