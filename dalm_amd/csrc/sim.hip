// K2-K4: in-batch similarity on the gfx950 matrix cores, exact f32.
//
// Stands in for get_cosine_sim + get_nt_xent_loss (+ log_softmax(scores).diag())
//   dalm/training/utils/train_utils.py:76-88,124
//   dalm/training/rag_e2e/train_rage2e.py:441-446
//
// One LDS-tiled GEMM kernel on v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate:
// bit-for-bit an fmaf chain, so |S| <= 100 keeps ~1e-5 absolute error - a bf16
// product would be off by ~0.4 in the logit at logit_scale = 100) with three
// epilogues:
//   EPI_STORE     C = alpha * A.B                       (get_cosine_sim, generic GEMM)
//   EPI_ROWSTATS  per-row (max, sum-exp) partials of S  (S never reaches HBM)
//   EPI_DS        closed-form dL/dS from the saved row/col log-sum-exps
// 256-thread workgroups = 4 waves (2x2), 128x128x32 tiles (64x64 per wave =
// 2x2 MFMA tiles, 64 accumulator VGPRs; 64x64x128 tiles for latency-bound small
// problems), operands staged k-major in LDS so that
// every MFMA fragment read is one conflict-free ds_read_b32 per lane.
// MFMA-bound: 2*m*n*D flop per launch against the 157 TF f32-matrix peak.
#include "common.hpp"
#include <stdlib.h>

namespace dalm {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK_BIG = 32;    // 128x128 tiles: 33 KB LDS, 3 blocks/CU
constexpr int BK_SMALL = 128;  // 64x64 tiles of latency-bound small problems: 4x fewer barrier rounds

enum { EPI_STORE = 0, EPI_ROWSTATS = 1, EPI_DS = 2, EPI_PARTIAL = 3 };

struct GemmParams {
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  int M, N, K;
  unsigned tiles_n;  // number of tile columns (set by the launcher)
  int k_chunk;       // K range per blockIdx.y (split-K); == K when gridDim.y == 1
  float alpha;
  int a_vec, b_vec;  // 16-byte vector loads legal for A / B
  // EPI_STORE / EPI_DS
  float* C; int64_t ldc;
  // EPI_ROWSTATS
  float* part_m; float* part_l;  // [P][M], P = 2 * tiles_n
  float* diag; int64_t diag_offset;
  // EPI_DS
  const float* row_coef; const float* row_lse; const float* col_coef; const float* col_lse;
};

__device__ __forceinline__ float4 guarded_ld4(const float* p, int nvalid, bool vec_ok) {
  if (nvalid >= 4 && vec_ok) return *reinterpret_cast<const float4*>(p);
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nvalid > 0) r.x = p[0];
  if (nvalid > 1) r.y = p[1];
  if (nvalid > 2) r.z = p[2];
  if (nvalid > 3) r.w = p[3];
  return r;
}

// Stages one operand tile (BR rows in the non-K dimension x BK) global -> regs -> LDS[k][r].
// KC: source is X[r][k] (k contiguous) -> transposing ds_write_b32 (stride BR+1: conflict-free)
// !KC: source is X[k][r] (r contiguous) -> ds_write_b128 rows (stride BR+4: 16-byte aligned)
template <int BR, int BK, bool KC>
struct Stage {
  static constexpr int STRIDE = KC ? BR + 1 : BR + 4;
  static constexpr int NV = BR * BK / 4 / 256;
  float4 v[NV];

  __device__ __forceinline__ void load(const float* X, int64_t ld, int r0, int k0, int R, int K, bool vec_ok) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int idx = p * 256 + tid;
      if constexpr (KC) {
        const int rr = idx / (BK / 4), kq = idx % (BK / 4);
        const int r = r0 + rr, k = k0 + kq * 4;
        int nv = (r < R) ? (K - k) : 0;
        nv = nv < 0 ? 0 : nv;
        v[p] = guarded_ld4(X + static_cast<int64_t>(r) * ld + k, nv, vec_ok);
      } else {
        const int kk = idx / (BR / 4), rq = idx % (BR / 4);
        const int k = k0 + kk, r = r0 + rq * 4;
        int nv = (k < K) ? (R - r) : 0;
        nv = nv < 0 ? 0 : nv;
        v[p] = guarded_ld4(X + static_cast<int64_t>(k) * ld + r, nv, vec_ok);
      }
    }
  }
  __device__ __forceinline__ void store(float* S) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int idx = p * 256 + tid;
      if constexpr (KC) {
        const int rr = idx / (BK / 4), kq = idx % (BK / 4);
        float* d = S + (kq * 4) * STRIDE + rr;
        d[0] = v[p].x; d[STRIDE] = v[p].y; d[2 * STRIDE] = v[p].z; d[3 * STRIDE] = v[p].w;
      } else {
        const int kk = idx / (BR / 4), rq = idx % (BR / 4);
        *reinterpret_cast<float4*>(S + kk * STRIDE + rq * 4) = v[p];
      }
    }
  }
};

template <int BM, int BN, int BK, bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_kernel(const GemmParams p) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  using SA = Stage<BM, BK, A_KC>;
  using SB = Stage<BN, BK, B_KC>;
  __shared__ __attribute__((aligned(16))) float lds[BK * SA::STRIDE + BK * SB::STRIDE];
  float* As = lds;
  float* Bs = lds + BK * SA::STRIDE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (1-D grid): workgroup L runs on XCD L % 8, each XCD has its own L2.  Give every
  // XCD a contiguous run of logical tiles (bijective form for any grid size) so the N-tiles that share an
  // A row-panel are co-resident on ONE XCD and the panel is fetched from HBM once instead of 8 times.
  const unsigned nwg = gridDim.x, L = blockIdx.x;
  const unsigned q8 = nwg >> 3, r8 = nwg & 7u, xcd = L & 7u;
  const unsigned wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (L >> 3);
  const int bx = static_cast<int>(wgid % p.tiles_n), by = static_cast<int>(wgid / p.tiles_n);
  const int bm0 = by * BM, bn0 = bx * BN;
  const int l31 = lane & 31, lhi = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  SA sa; SB sb;
  // split-K: blockIdx.y owns K range [k_lo, k_hi) (EPI_PARTIAL writes raw accumulators to its own slab;
  // gridDim.y == 1 elsewhere).  Measured codegen effect on the 128x128 tiles (A/B on one box): with the run-time
  // range the STORE / DS variants are 14 % faster at 16384^2 and the ROWSTATS variant 4-18 % slower, so the
  // rowstats instantiation keeps the compile-time full range.
  int k_lo = 0, k_hi = p.K;
  if constexpr (EPI != EPI_ROWSTATS) {
    k_lo = static_cast<int>(blockIdx.y) * p.k_chunk;
    k_hi = min(p.K, k_lo + p.k_chunk);
  }
  const int nk = (k_hi - k_lo + BK - 1) / BK;
  sa.load(p.A, p.lda, bm0, k_lo, p.M, k_hi, p.a_vec);
  sb.load(p.B, p.ldb, bn0, k_lo, p.N, k_hi, p.b_vec);
  sa.store(As); sb.store(Bs);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) {  // prefetch the next slab into registers while the MFMAs run
      sa.load(p.A, p.lda, bm0, k_lo + (kt + 1) * BK, p.M, k_hi, p.a_vec);
      sb.load(p.B, p.ldb, bn0, k_lo + (kt + 1) * BK, p.N, k_hi, p.b_vec);
    }
    const float* a_base = As + lhi * SA::STRIDE + wm * WM + l31;
    const float* b_base = Bs + lhi * SB::STRIDE + wn * WN + l31;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = a_base[kk * SA::STRIDE + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = b_base[kk * SB::STRIDE + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      sa.store(As); sb.store(Bs);
      __syncthreads();
    }
  }

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31,
  //      row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = bm0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const bool row_ok = row < p.M;
      if constexpr (EPI == EPI_STORE) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = bn0 + wn * WN + j * 32 + l31;
          if (row_ok && col < p.N) p.C[static_cast<int64_t>(row) * p.ldc + col] = __fmul_rn(p.alpha, acc[i][j][r]);
        }
      } else if constexpr (EPI == EPI_PARTIAL) {
        float* slab = p.C + static_cast<int64_t>(blockIdx.y) * p.M * p.ldc;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = bn0 + wn * WN + j * 32 + l31;
          if (row_ok && col < p.N) slab[static_cast<int64_t>(row) * p.ldc + col] = acc[i][j][r];
        }
      } else if constexpr (EPI == EPI_DS) {
        const float rc = row_ok ? p.row_coef[row] : 0.f;
        const float rl = row_ok ? p.row_lse[row] : 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = bn0 + wn * WN + j * 32 + l31;
          if (row_ok && col < p.N) {
            const float s = __fmul_rn(p.alpha, acc[i][j][r]);  // no fma contraction: same S bits as the rowstats pass
            const float cc = p.col_coef[col];
            float d = rc * fast_exp(s - rl) + cc * fast_exp(s - p.col_lse[col]);
            if (static_cast<int64_t>(col) == p.diag_offset + row) d -= (rc + cc);
            p.C[static_cast<int64_t>(row) * p.ldc + col] = d;
          }
        }
      } else {  // EPI_ROWSTATS
        float v[TN];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = bn0 + wn * WN + j * 32 + l31;
          v[j] = (col < p.N) ? __fmul_rn(p.alpha, acc[i][j][r]) : -INFINITY;
          if (row_ok && col < p.N && static_cast<int64_t>(col) == p.diag_offset + row) p.diag[row] = v[j];
          mx = fmaxf(mx, v[j]);
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float s = 0.f;
        if (mx != -INFINITY) {
#pragma unroll
          for (int j = 0; j < TN; ++j) s += fast_exp(v[j] - mx);
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (row_ok && l31 == 0) {
          const int64_t pi = static_cast<int64_t>(bx * 2 + wn) * p.M + row;
          p.part_m[pi] = mx;
          p.part_l[pi] = s;
        }
      }
    }
  }
}

// ---- split-K epilogues: S_ij = alpha * sum_z partial[z][i][j] (fixed order: deterministic, and the
// rowstats and dS passes see bit-identical S).  One block per row; n is small here (<= a few thousand).
__device__ __forceinline__ float splitk_s(const float* __restrict__ part, int64_t slab, int64_t off, int SK,
                                          float alpha) {
  float acc = part[off];
  for (int z = 1; z < SK; ++z) acc += part[z * slab + off];
  return __fmul_rn(alpha, acc);
}

__global__ __launch_bounds__(256) void splitk_rowstats_kernel(const float* __restrict__ part, int SK, int M, int N,
                                                              int64_t ldp, float alpha, int64_t diag_offset,
                                                              float* __restrict__ row_lse, float* __restrict__ diag) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const int64_t slab = static_cast<int64_t>(M) * ldp, base = static_cast<int64_t>(row) * ldp;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < N; j += 256) m = fmaxf(m, splitk_s(part, slab, base + j, SK, alpha));
  m = block_max<256>(m, red);
  float l = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) l += fast_exp(splitk_s(part, slab, base + j, SK, alpha) - m);
  l = block_sum<256>(l, red);
  if (threadIdx.x == 0) {
    row_lse[row] = m + __logf(l);
    diag[row] = splitk_s(part, slab, base + diag_offset + row, SK, alpha);
  }
}

__global__ __launch_bounds__(256) void splitk_ds_kernel(const float* __restrict__ part, int SK, int M, int N,
                                                        int64_t ldp, float alpha, int64_t diag_offset,
                                                        const float* __restrict__ row_coef,
                                                        const float* __restrict__ row_lse,
                                                        const float* __restrict__ col_coef,
                                                        const float* __restrict__ col_lse, float* __restrict__ dS,
                                                        int64_t ldd) {
  const int row = blockIdx.x;
  const int64_t slab = static_cast<int64_t>(M) * ldp, base = static_cast<int64_t>(row) * ldp;
  const float rc = row_coef[row], rl = row_lse[row];
  for (int j = threadIdx.x; j < N; j += 256) {
    const float sij = splitk_s(part, slab, base + j, SK, alpha);
    const float cc = col_coef[j];
    float d = rc * fast_exp(sij - rl) + cc * fast_exp(sij - col_lse[j]);
    if (static_cast<int64_t>(j) == diag_offset + row) d -= (rc + cc);
    dS[static_cast<int64_t>(row) * ldd + j] = d;
  }
}

// merge P partial (max,sum) pairs per row -> row_lse
__global__ __launch_bounds__(256) void rowstats_merge_kernel(const float* __restrict__ part_m,
                                                             const float* __restrict__ part_l, int P, int M,
                                                             float* __restrict__ row_lse) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= M) return;
  // partials are fetched 8 at a time (clamped, unconditional): a `for q < P` loop of loads is a serial latency chain
  float m = -INFINITY, l = 0.f;
  for (int q0 = 0; q0 < P; q0 += 8) {
    float pm[8], pl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t o = static_cast<int64_t>(min(q0 + u, P - 1)) * M + row;
      pm[u] = part_m[o];
      pl[u] = part_l[o];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (q0 + u < P && pm[u] != -INFINITY) {   // online merge in fixed order q = 0..P-1
        const float mn = fmaxf(m, pm[u]);
        l = l * fast_exp(m - mn) + pl[u] * fast_exp(pm[u] - mn);
        m = mn;
      }
    }
  }
  row_lse[row] = m + __logf(l);
}

// ---- small row kernels on a materialised S (drop-in get_nt_xent_loss etc.) ----
// one block per row: row_lse[i] = logsumexp_j S[i,j]; optional doc_lp[i] = S[i,i] - row_lse[i]
__global__ __launch_bounds__(256) void rows_lse_kernel(const float* __restrict__ S, int n_cols, int64_t sr,
                                                       int64_t sc, float* __restrict__ row_lse,
                                                       float* __restrict__ doc_lp) {
  __shared__ float red[4];
  const int i = blockIdx.x;
  const float* row = S + i * sr;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < n_cols; j += 256) m = fmaxf(m, row[j * sc]);
  m = block_max<256>(m, red);
  float l = 0.f;
  for (int j = threadIdx.x; j < n_cols; j += 256) l += fast_exp(row[j * sc] - m);
  l = block_sum<256>(l, red);
  if (threadIdx.x == 0) {
    const float lse = m + __logf(l);
    row_lse[i] = lse;
    if (doc_lp) doc_lp[i] = row[i * sc] - lse;
  }
}

// loss = mean_i (row_lse[i] - S[i,i])      (cross_entropy(S, arange(n)))
__global__ __launch_bounds__(256) void nt_xent_reduce_kernel(const float* __restrict__ S, int n, int64_t sr,
                                                             int64_t sc, const float* __restrict__ row_lse,
                                                             float* __restrict__ loss) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += row_lse[i] - S[i * sr + i * sc];
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) loss[0] = s / static_cast<float>(n);
}

// dS[i,j] (+)= c_i * (exp(S_ij - lse_i) - [i==j]) with c_i = sign * (coef ? coef[i] : gscale/n)
__global__ __launch_bounds__(256) void rows_softmax_grad_kernel(const float* __restrict__ S, int n_cols,
                                                                int64_t sr, int64_t sc,
                                                                const float* __restrict__ row_lse,
                                                                const float* __restrict__ coef,
                                                                const float* __restrict__ gscale, float cdiv,
                                                                float sign, float* dS, int64_t dsr,
                                                                int64_t dsc, int accumulate) {
  const int i = blockIdx.x;
  const float c = sign * (coef ? coef[i] : gscale[0] / cdiv);
  const float lse = row_lse[i];
  for (int j = threadIdx.x; j < n_cols; j += 256) {
    float d = c * (fast_exp(S[i * sr + j * sc] - lse) - ((j == i) ? 1.f : 0.f));
    float* o = dS + i * dsr + j * dsc;
    *o = accumulate ? (*o + d) : d;
  }
}

__global__ __launch_bounds__(256) void contrastive_finalize_kernel(const float* __restrict__ row_lse,
                                                                   const float* __restrict__ col_lse,
                                                                   const float* __restrict__ diag, int n_local,
                                                                   float n_global, float* __restrict__ out,
                                                                   float* __restrict__ doc_lp) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n_local; i += 256) {
    const float d = diag[i];
    s += (row_lse[i] - d) + (col_lse[i] - d);
    if (doc_lp) doc_lp[i] = d - row_lse[i];
  }
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) out[0] = 0.5f * s / n_global;
}

// ---------------------------------------------------------------------------------------------------
// Flash-style backward: dA[m,D] = alpha * dS[m,n] . B[n,D] with dS rebuilt tile by tile - no m x n panel in HBM.
// A workgroup owns 32 rows of A and ALL D output columns (accumulators: 32 x D f32 = NT 32x32 MFMA tiles per
// wave, 4 waves side by side, D = 128*NT <= 1024) and walks its share of the column blocks of B, 128 columns at
// a time.  The f32 MFMA issues once per 64 cycles per SIMD - slow enough that L2/L1 can feed it DIRECTLY:
//   S phase   wave w: S[32 x 32] = A_blk . B_cols^T over K = D, both fragments read straight from k-major copies
//             At[D][m'] / Bt[D][n'] (lane (c,h) reads Xt[2s+h][x0+c]: full 128-byte rows), 16 steps prefetched in
//             registers; no LDS staging, no barrier, k ascending => the same S bits the row statistics saw
//   transform dS = rc_i e^{S-rl_i} + cc_j e^{S-cl_j} - [diag](rc_i + cc_j) in registers -> LDS, k-major
//   dA phase  acc[32 x D] += dS[32 x 128] . B_blk[128 x D]: A-fragments from the dS tile in LDS, B-fragments
//             straight from the row-major B (lane (c,h) reads B[j+h][d+c]), 4 steps prefetched
// Two barriers per block (around the dS tile); both phases issue 512 MFMAs per wave per block at D = 1024.
// The k-major copies are made once per call by transpose_pad_kernel (zero-padded to the tile grid, so the hot
// loops carry no bounds checks).  Column blocks are split over `nsplit` workgroups when m/32 row blocks cannot
// fill the chip; those write raw partial outputs and flash_reduce_kernel sums them in fixed order.
// Workspace: (m' + n') * D floats for the copies + nsplit*m*D for the partials - never m*n.
// MFMA-bound: 4*m*n*D flop per launch against the 157 TF f32-matrix peak.
// ---------------------------------------------------------------------------------------------------
struct FlashParams {
  const float* At; const float* Bt; const float* B;   // At [D][ldm], Bt [D][ldn] (zero-padded), B [n][D]
  int m, n, D, ldm, ldn;
  float alpha;
  int64_t diag_offset;
  const float* row_coef; const float* row_lse; const float* col_coef; const float* col_lse;
  float* out;            // dA (nsplit == 1) or slabs [nsplit][m][D]
  int row_blocks, nsplit, blocks_per_split;
  int debug_hot;         // timing experiment only (DALM_FLASH_DEBUG=1): every step re-reads the same cached rows
};

constexpr int FBM = 32, FBN = 128;

// X [R][K] row-major -> Xt [K][ld] with Xt[k][r] = X[r][k] for r < R, 0 for R <= r < ld.  32x32 tiles through
// LDS; grid (ld/32, ceil(K/32), 2): blockIdx.z selects (X0 -> Xt0) or (X1 -> Xt1) so both operands go in one launch.
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ X0, int R0, int ld0,
                                                            float* __restrict__ Xt0, const float* __restrict__ X1,
                                                            int R1, int ld1, float* __restrict__ Xt1, int K, int Kpad,
                                                            unsigned* __restrict__ tickets = nullptr, int ntickets = 0) {
  __shared__ float tile[32][33];
  // the arrival tickets of the kernel that follows in the same call (streaming row statistics): zeroed here, so the caller
  // owes no initialisation and a poisoned workspace cannot survive a call
  if (tickets && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    for (int i = threadIdx.x; i < ntickets; i += 256) tickets[i] = 0u;
  const bool second = blockIdx.z != 0;
  const float* X = second ? X1 : X0;
  float* Xt = second ? Xt1 : Xt0;
  const int R = second ? R1 : R0, ld = second ? ld1 : ld0;
  const int r0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  if (r0 >= ld) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = r0 + ty + 8 * q, k = k0 + tx;
    tile[ty + 8 * q][tx] = (r < R && k < K) ? X[static_cast<int64_t>(r) * K + k] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = k0 + ty + 8 * q, r = r0 + tx;
    if (k < Kpad) Xt[static_cast<int64_t>(k) * ld + r] = tile[tx][ty + 8 * q];   // rows K..Kpad-1 are zeros
  }
}

template <int NT>
__global__ __launch_bounds__(256, 2) void sim_flash_grad_kernel(const FlashParams p) {
  constexpr int DS_STRIDE = FBM + 1;
  constexpr int SG = 8;    // S-phase steps per prefetch group (2 dwords each)
  __shared__ float Ds[FBN * DS_STRIDE];
  __shared__ float rc_s[FBM], rl_s[FBM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int rb = static_cast<int>(blockIdx.x) % p.row_blocks, z = static_cast<int>(blockIdx.x) / p.row_blocks;
  const int i0 = rb * FBM;
  const int c0 = wave * NT * 32;  // first output column of this wave
  const int nblocks = (p.n + FBN - 1) / FBN;
  const int jb_lo = z * p.blocks_per_split, jb_hi = min(nblocks, jb_lo + p.blocks_per_split);

  if (tid < FBM) {
    const int r = min(i0 + tid, p.m - 1);
    rc_s[tid] = p.row_coef[r];
    rl_s[tid] = p.row_lse[r];
  }

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int nsteps = p.D / 2;                       // D = 128 * NT: a multiple of 2 * SG
  // every load below is (wave-uniform row pointer) + (per-lane 32-bit offset): SGPR-base addressing, so a
  // prefetch group of 32 loads does not hold 32 64-bit VGPR addresses
  // (buffer_load_dword v, v_off, s[rsrc], s_off offen: the compiler otherwise materialises one 64-bit VGPR address
  // per load and spills)
  const __amdgpu_buffer_rsrc_t at_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.At), 0, p.D * p.ldm * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t bt_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Bt), 0, p.D * p.ldn * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B), 0, p.n * p.D * 4, 0x00020000);
  const int a_off = (lhi * p.ldm + i0 + l31) * 4;                       // bytes
  const int a_step = p.debug_hot ? 0 : 2 * p.ldm * 4, b_step = p.debug_hot ? 0 : 2 * p.ldn * 4;   // bytes per two-wide step
  const int d_base = (c0 + l31) * 4;                                    // dA phase: column c0 + l31 (+ 32 t)

#pragma unroll 1
  for (int jb = jb_lo; jb < jb_hi; ++jb) {
    const int j0 = jb * FBN;
    const int col = j0 + wave * 32 + l31;
    // ---------------- S phase: LDS-free, register double buffer of SG steps ----------------
    f32x16 sacc;   // one chain, k ascending: the same S bits the row statistics saw (two chains measured no faster)
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
    const int b_off = (lhi * p.ldn + col) * 4;
    float fa0[SG], fb0[SG], fa1[SG], fb1[SG], fa2[SG], fb2[SG];
    auto loads = [&](float (&fa)[SG], float (&fb)[SG], int g) {   // group g = steps [g*SG, (g+1)*SG)
#pragma unroll
      for (int u = 0; u < SG; ++u) {
        fa[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(at_rs, a_off, (g * SG + u) * a_step, 0));
        fb[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bt_rs, b_off, (g * SG + u) * b_step, 0));
      }
    };
    auto mfmas = [&](const float (&fa)[SG], const float (&fb)[SG]) {
#pragma unroll
      for (int u = 0; u < SG; ++u) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u], fb[u], sacc, 0, 0, 0);
    };
    // three register buffers in rotation: every group is requested two groups (16 MFMAs ~ 1000 cycles, about one
    // L2 round trip under load) before it is consumed; with two buffers (512 cycles) a lone wave sat at 54 % MFMA
    const int G = nsteps / SG;   // 8 * NT groups
    loads(fa0, fb0, 0);
    loads(fa1, fb1, 1);
    const bool col_ok = col < p.n;
    const int colc = min(col, p.n - 1);
    const float ccj = p.col_coef[colc], clj = p.col_lse[colc];
    // issue order: each MFMA is followed by two of the loads of a later group.  Left alone, hipcc emits the 16 loads
    // of a group back to back (~10 issue cycles each) and the matrix pipe idles meanwhile - measured 58 % MFMA rate
    // for a lone wave whether the loads hit the cache or not
    auto interleave = [&]() {
#pragma unroll
      for (int u = 0; u < SG; ++u) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // 2 VMEM reads
      }
    };
    int g = 0;
#pragma unroll 1
    for (; g + 3 <= G; g += 3) {   // branch-free body: loads past the last group are clamped (re-read, unused)
      loads(fa2, fb2, g + 2);
      mfmas(fa0, fb0);
      interleave();
      loads(fa0, fb0, min(g + 3, G - 1));
      mfmas(fa1, fb1);
      interleave();
      loads(fa1, fb1, min(g + 4, G - 1));
      mfmas(fa2, fb2);
      interleave();
    }
    if (g < G) mfmas(fa0, fb0);
    if (g + 1 < G) mfmas(fa1, fb1);
    // first fragments of the dA phase: independent of dS, issued before the barrier
    float b0[2][NT], b1[2][NT];
    auto loadb = [&](float (&dst)[2][NT], int s0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        // lane half h reads row j + h.  Rows past n are clamped onto row n-1 (uniformly: the row offset is an SGPR,
        // and the h = 1 half drops its +1 when row j is the last one): their dS entries are exactly 0
        const int ju = p.debug_hot ? 0 : min(j0 + 2 * (s0 + u), p.n - 1);
        const int hstride = (ju + 1 < p.n) ? p.D * 4 : 0;
        const int voff = d_base + lhi * hstride;
#pragma unroll
        for (int t = 0; t < NT; ++t)
          dst[u][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rs, voff + 128 * t, ju * p.D * 4, 0));
      }
    };
    loadb(b0, 0);
    __syncthreads();   // previous block's dA phase has finished reading Ds (and rc_s/rl_s are visible)
    // ---------------- dS tile -> LDS (k-major: Ds[j_local][row]) ----------------
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const int row = i0 + rl;
      const float sv = __fmul_rn(p.alpha, sacc[r]);
      const float rci = rc_s[rl];
      float d = rci * fast_exp(sv - rl_s[rl]) + ccj * fast_exp(sv - clj);
      if (static_cast<int64_t>(col) == p.diag_offset + row) d -= (rci + ccj);
      Ds[(wave * 32 + l31) * DS_STRIDE + rl] = (col_ok && row < p.m) ? d : 0.f;
    }
    __syncthreads();
    // ---------------- dA phase ----------------
    auto compute = [&](const float (&src)[2][NT], int s0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float a = Ds[(2 * (s0 + u) + lhi) * DS_STRIDE + l31];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, src[u][t], acc[t], 0, 0, 0);
      }
    };
    auto interleave_da = [&]() {
#pragma unroll
      for (int u = 0; u < 2 * NT; ++u) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
      }
    };
#pragma unroll 1
    for (int s0 = 0; s0 < FBN / 2; s0 += 4) {   // branch-free: the load past the last step is clamped (unused)
      loadb(b1, s0 + 2);
      compute(b0, s0);
      interleave_da();
      loadb(b0, min(s0 + 4, FBN / 2 - 2));
      compute(b1, s0 + 2);
      interleave_da();
    }
  }

  float* out = p.out + (p.nsplit > 1 ? static_cast<int64_t>(z) * p.m * p.D : 0);
  const float oscale = (p.nsplit > 1) ? 1.f : p.alpha;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = i0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const int cc = c0 + 32 * t + l31;
      if (row < p.m) out[static_cast<int64_t>(row) * p.D + cc] = __fmul_rn(oscale, acc[t][r]);
    }
}

// ---------------------------------------------------------------------------------------------------
// Streaming row statistics (the forward of the large-batch contrastive loss): the S phase of the kernel above
// without the dA phase.  A workgroup owns 32*RT rows and walks its share of the 128-column blocks; every wave
// keeps an online (max, sum-exp) per (row, its column lane) in registers and folds each finished 32x32 S tile into
// it - no LDS, no barrier until the final reduction over lanes and waves.  RT = 2 row tiles per wave halves the
// B-fragment loads per MFMA and gives two independent accumulation chains (used once there are enough row blocks).
// Output: partial (max, sum-exp) per column split, merged by rowstats_merge_kernel; diag[i] = S[i, off + i].
// MFMA-bound: 2*m*n*D flop per launch.
// ---------------------------------------------------------------------------------------------------
struct StreamStatsParams {
  const float* At; const float* Bt;   // [Kpad][ldm], [Kpad][ldn], zero-padded
  int m, n, Kpad, ldm, ldn;
  float alpha;
  int64_t diag_offset;
  float* part_m; float* part_l;       // [nsplit][m]  (GEMM-epilogue paths; the streaming kernel publishes granules)
  unsigned long long* part;           // streaming kernel: [nsplit][m] (max, sum exp) granules, merged IN the launch
  unsigned* tickets;                  // [row_blocks], zeroed by transpose_pad_kernel at the start of the call
  float* row_lse;
  float* diag;
  int row_blocks, nsplit, blocks_per_split;
  // top-k modes
  float* gmax; int ng;                // MODE_GMAX: gmax[m][ng], one maximum per 32-column group
};
enum { MODE_STATS = 0, MODE_GMAX = 1 };

// KS > 1 (MODE_STATS, mid-size problems whose 128-column blocks cannot fill the chip): the 4 waves are 4/KS column
// tiles x KS slices of K; a workgroup then owns 32*RT rows x 128/KS columns per block, the K slices of a tile are summed
// through LDS in fixed order (slice 0 + 1 + 2 + 3) before the statistics see them - one barrier per column block.
template <int RT, int MODE, int KS = 1>
__global__ __launch_bounds__(256, KS > 1 ? 3 : 2) void sim_rowstats_stream_kernel(const StreamStatsParams p) {
  static_assert(KS == 1 || (MODE == MODE_STATS && RT == 1), "K slices: statistics mode, one row tile");
  constexpr int SG = 8;
  constexpr int CBN = FBN / KS;       // columns per workgroup and block
  __shared__ float red_m[4][32 * RT], red_l[4][32 * RT];
  __shared__ float xchg[KS > 1 ? 2 * (4 / KS) * (KS - 1) * 16 * 64 : 1];   // [parity][tile][slice-1][r][lane]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the K-slice offsets below stay in SGPRs
  const int ct = wave / KS, kp = wave % KS;
  const __amdgpu_buffer_rsrc_t at_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.At), 0, p.Kpad * p.ldm * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t bt_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Bt), 0, p.Kpad * p.ldn * 4, 0x00020000);
  const int rb = static_cast<int>(blockIdx.x) % p.row_blocks, z = static_cast<int>(blockIdx.x) / p.row_blocks;
  const int i0 = rb * 32 * RT;
  const int nblocks = (p.n + CBN - 1) / CBN;
  const int jb_lo = z * p.blocks_per_split, jb_hi = min(nblocks, jb_lo + p.blocks_per_split);
  const int a_off = (lhi * p.ldm + i0 + l31) * 4;
  const int a_step = 2 * p.ldm * 4, b_step = 2 * p.ldn * 4;
  const int G = p.Kpad / (2 * SG) / KS;   // groups of 16 k per wave; Kpad is a multiple of 16 * KS
  const int g0 = kp * G;

  float rmax[RT][16], rsum[RT][16];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { rmax[t][r] = -INFINITY; rsum[t][r] = 0.f; }

#pragma unroll 1
  for (int jb = jb_lo; jb < jb_hi; ++jb) {
    const int col = jb * CBN + ct * 32 + l31;
    const int b_off = (lhi * p.ldn + col) * 4;
    f32x16 sacc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[t][r] = 0.f;
    float fa0[RT][SG], fb0[SG], fa1[RT][SG], fb1[SG], fa2[RT][SG], fb2[SG];
    auto loads = [&](float (&fa)[RT][SG], float (&fb)[SG], int g) {
#pragma unroll
      for (int u = 0; u < SG; ++u) {
#pragma unroll
        for (int t = 0; t < RT; ++t)
          fa[t][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(at_rs, a_off + 128 * t, ((g0 + g) * SG + u) * a_step, 0));
        fb[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bt_rs, b_off, ((g0 + g) * SG + u) * b_step, 0));
      }
    };
    auto mfmas = [&](const float (&fa)[RT][SG], const float (&fb)[SG]) {
#pragma unroll
      for (int u = 0; u < SG; ++u)
#pragma unroll
        for (int t = 0; t < RT; ++t) sacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t][u], fb[u], sacc[t], 0, 0, 0);
    };
    auto interleave = [&]() {
#pragma unroll
      for (int u = 0; u < SG; ++u) {   // RT + 1 loads per RT MFMAs
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        if constexpr (RT == 2) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
    };
    loads(fa0, fb0, 0);
    loads(fa1, fb1, min(1, G - 1));
    int g = 0;
#pragma unroll 1
    for (; g + 3 <= G; g += 3) {
      loads(fa2, fb2, g + 2);
      mfmas(fa0, fb0);
      interleave();
      loads(fa0, fb0, min(g + 3, G - 1));
      mfmas(fa1, fb1);
      interleave();
      loads(fa1, fb1, min(g + 4, G - 1));
      mfmas(fa2, fb2);
      interleave();
    }
    if (g < G) mfmas(fa0, fb0);
    if (g + 1 < G) mfmas(fa1, fb1);
    const bool col_ok = col < p.n;
    if constexpr (KS > 1) {
      float* xb = xchg + ((jb & 1) * (4 / KS) + ct) * (KS - 1) * 16 * 64;
      if (kp != 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xb[((kp - 1) * 16 + r) * 64 + lane] = sacc[0][r];
      }
      __syncthreads();   // parity double buffer: the next block's writes go to the other half
      if (kp != 0) continue;
#pragma unroll
      for (int q = 0; q < KS - 1; ++q)
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 8) {   // 8 LDS reads in flight at a time (the fragment registers are dead here,
          __builtin_amdgcn_sched_barrier(0);   // but an unbounded hoist of all 16 (KS - 1) reads spills)
#pragma unroll
          for (int r = r0; r < r0 + 8; ++r) sacc[0][r] += xb[(q * 16 + r) * 64 + lane];
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MODE == MODE_STATS) {
      // ---- fold the finished tile(s) into the online row statistics ----
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          const float v = col_ok ? __fmul_rn(p.alpha, sacc[t][r]) : -INFINITY;
          if (col_ok && row < p.m && static_cast<int64_t>(col) == p.diag_offset + row) p.diag[row] = v;
          const float mn = fmaxf(rmax[t][r], v);
          if (mn != -INFINITY) rsum[t][r] = rsum[t][r] * fast_exp(rmax[t][r] - mn) + fast_exp(v - mn);
          rmax[t][r] = mn;
        }
    } else if constexpr (MODE == MODE_GMAX) {
      // ---- top-k pass 1: one maximum per (row, 32-column group): 1/32 of the score matrix ----
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          float v = col_ok ? __fmul_rn(p.alpha, sacc[t][r]) : -INFINITY;
#pragma unroll
          for (int off = 16; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
          if (l31 == 0 && row < p.m) p.gmax[static_cast<int64_t>(row) * p.ng + jb * 4 + wave] = v;
        }
    }
  }
  if constexpr (MODE != MODE_STATS) return;
  // ---- reduce over the 32 column lanes of each half, then over the 4 waves ----
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float mx = rmax[t][r], l = rsum[t][r];
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
        const float om = __shfl_xor(mx, off, 64), ol = __shfl_xor(l, off, 64);
        const float mn = fmaxf(mx, om);
        l = (mn == -INFINITY) ? 0.f : l * fast_exp(mx - mn) + ol * fast_exp(om - mn);
        mx = mn;
      }
      if (l31 == 0) {
        const int rl = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        red_m[wave][rl] = mx;
        red_l[wave][rl] = l;
      }
    }
  __syncthreads();
  if (tid < 32 * RT && i0 + tid < p.m) {
    float mx = red_m[0][tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) mx = fmaxf(mx, red_m[w][tid]);
    float l = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w)
      if (red_m[w][tid] != -INFINITY) l += red_l[w][tid] * fast_exp(red_m[w][tid] - mx);
    if (p.nsplit == 1) {
      p.row_lse[i0 + tid] = mx + __logf(l);
    } else {
      st_pair(p.part + static_cast<int64_t>(z) * p.m + i0 + tid, mx, l);   // write-through (sc1) granule
    }
  }
  if (p.nsplit == 1) return;
  // ---- round 4: the merge of the column splits moved INTO the launch (it was rowstats_merge_kernel, a 4 us launch at
  // 1200^2): the last of a row block's nsplit workgroups folds the nsplit granules of its rows in split order z = 0 ..
  // nsplit-1 (the order the merge kernel used: same bits whoever arrives last).  Hand-off: common.hpp. ----
  unsigned* slot = reinterpret_cast<unsigned*>(&red_l[3][0]);
  if (draw_ticket(p.tickets + rb, slot) != static_cast<unsigned>(p.nsplit - 1)) return;
  if (tid == 0) __hip_atomic_store(p.tickets + rb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid < 32 * RT && i0 + tid < p.m) {
    float m = -INFINITY, l = 0.f;
    for (int q0 = 0; q0 < p.nsplit; q0 += 8) {
      float pm[8], pl[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) ld_pair(p.part + static_cast<int64_t>(min(q0 + u, p.nsplit - 1)) * p.m + i0 + tid, pm[u], pl[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (q0 + u < p.nsplit && pm[u] != -INFINITY) {
          const float mn = fmaxf(m, pm[u]);
          l = l * fast_exp(m - mn) + pl[u] * fast_exp(pm[u] - mn);
          m = mn;
        }
      }
    }
    p.row_lse[i0 + tid] = m + __logf(l);
  }
}

// ---- exact top-k on the streaming kernel (dalm/eval/utils.py:18-68 builds an approximate hnswlib
// index; here the search is exact and the [Nq, Nc] score matrix never exists) ----
// The MFMA pass (MODE_GMAX) leaves one maximum per 32 corpus columns.  The k-th largest of a row's group maxima is a
// lower bound T of its k-th largest score (k distinct scores >= T exist: those maxima), so every top-k member lives in
// a group whose maximum is >= T - about k groups per row.  topk_threshold_kernel finds T; topk_refine_kernel
// recomputes only those groups' 32 scores (plain FMA dot products against the k-major corpus copy: ~k*32 of them per
// query instead of a second pass over the whole corpus), keeps the scores >= T and sorts them (value descending,
// lower corpus index first on ties: deterministic).
__global__ __launch_bounds__(256) void topk_threshold_kernel(const float* __restrict__ gmax, int ng, int k, float abs_slack,
                                                             float* __restrict__ thr) {
  __shared__ float red[4];
  __shared__ float redc[4];
  const float* row = gmax + static_cast<int64_t>(blockIdx.x) * ng;
  float bound = INFINITY;   // values >= bound are already counted
  int have = 0;
  float cur = -INFINITY;
  for (int round = 0; round < k; ++round) {
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < ng; i += 256) {
      const float v = row[i];
      if (v < bound) mx = fmaxf(mx, v);
    }
    mx = block_max<256>(mx, red);
    if (mx == -INFINITY) break;
    float c = 0.f;
    for (int i = threadIdx.x; i < ng; i += 256) c += (row[i] == mx) ? 1.f : 0.f;
    c = block_sum<256>(c, redc);
    cur = mx;
    have += static_cast<int>(c);
    bound = mx;
    if (have >= k) break;
  }
  // slack: the refine pass re-evaluates the scores with a VALU fma chain (same k order as the MFMA chain, so normally the same
  // bits).  Relative part for large scores, absolute part (~ 2 eps sqrt(K) |alpha| for O(1)-norm embeddings) for scores near 0,
  // where a purely relative slack vanishes; a row that still ends up with fewer than k candidates reports overflow (refine).
  if (threadIdx.x == 0) thr[blockIdx.x] = (have >= k) ? cur - fabsf(cur) * 1e-6f - abs_slack : -INFINITY;
}

// one workgroup per query row.  LDS: q[Kpad] | glist[cap] | cand_val[cap] | cand_idx[cap]
__global__ __launch_bounds__(256) void topk_refine_kernel(const float* __restrict__ At, int ldm, const float* __restrict__ Bt,
                                                          int ldn, int Kpad, int n, float alpha,
                                                          const float* __restrict__ gmax, int ng,
                                                          const float* __restrict__ thr, int cap, int k,
                                                          float* __restrict__ out_val, int64_t* __restrict__ out_idx,
                                                          int* __restrict__ overflow) {
  extern __shared__ float sm[];
  __shared__ int n_groups, n_cand;
  __shared__ float red[4];
  __shared__ int redi[4];
  float* q = sm;
  int* glist = reinterpret_cast<int*>(sm + Kpad);
  float* cv = sm + Kpad + cap;
  int* ci = reinterpret_cast<int*>(sm + Kpad + 2 * cap);
  const int row = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) { n_groups = 0; n_cand = 0; }
  for (int kk = tid; kk < Kpad; kk += 256) q[kk] = At[static_cast<int64_t>(kk) * ldm + row];
  const float T = thr[row];
  __syncthreads();
  for (int g = tid; g < ng; g += 256)
    if (gmax[static_cast<int64_t>(row) * ng + g] >= T) {
      const int pos = atomicAdd(&n_groups, 1);
      if (pos < cap) glist[pos] = g;
    }
  __syncthreads();
  const int total_groups = n_groups;
  if (total_groups > cap) {   // massive ties: report and let the caller fall back
    if (tid == 0) atomicAdd(overflow, 1);
    return;
  }
  // half-wave per group: lane c of the half evaluates column g*32 + c
  const int hw = tid >> 5, c = tid & 31;
  for (int gi = hw; gi < total_groups; gi += 8) {
    const int col = glist[gi] * 32 + c;
    const float* bcol = Bt + min(col, ldn - 1);
    float s0 = 0.f;
#pragma unroll 8
    for (int kk = 0; kk < Kpad; ++kk) s0 = fmaf(q[kk], bcol[static_cast<int64_t>(kk) * ldn], s0);
    const float v = __fmul_rn(alpha, s0);
    if (col < n && v >= T) {
      const int slot = atomicAdd(&n_cand, 1);
      if (slot < cap) { cv[slot] = v; ci[slot] = col; }
    }
  }
  __syncthreads();
  const int cnt = n_cand;
  if (cnt > cap || cnt < min(k, n)) {   // too many ties, or a recomputed score slipped under the threshold: caller falls back
    if (tid == 0) atomicAdd(overflow, 1);
    return;
  }
  for (int i = cnt + tid; i < cap; i += 256) { cv[i] = -INFINITY; ci[i] = 0x7fffffff; }
  __syncthreads();
  for (int j = 0; j < k; ++j) {
    float mx = -INFINITY;
    for (int i = tid; i < cap; i += 256) mx = fmaxf(mx, cv[i]);
    mx = block_max<256>(mx, red);
    int bi = 0x7fffffff;   // smallest corpus index among the maxima
    for (int i = tid; i < cap; i += 256) if (cv[i] == mx) bi = min(bi, ci[i]);
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bi = min(bi, __shfl_xor(bi, off, 64));
    if (lane == 0) redi[w] = bi;
    __syncthreads();
    bi = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
    __syncthreads();
    if (tid == 0) {
      out_val[static_cast<int64_t>(row) * k + j] = mx;
      out_idx[static_cast<int64_t>(row) * k + j] = (mx == -INFINITY) ? -1 : bi;
    }
    for (int i = tid; i < cap; i += 256) if (cv[i] == mx && ci[i] == bi) cv[i] = -INFINITY;
    __syncthreads();
  }
}

// dA = alpha * sum_z slab[z] (fixed order), float4-wide; n4 = m*D/4 (D % 4 == 0 on this path)
__global__ __launch_bounds__(256) void flash_reduce_kernel(const float4* __restrict__ slab, int nsplit, int64_t n4,
                                                           float alpha, float4* __restrict__ out) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += gridDim.x * 256ll) {
    float4 a = slab[i];
    for (int zz = 1; zz < nsplit; ++zz) {
      const float4 b = slab[zz * n4 + i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    out[i] = make_float4(__fmul_rn(alpha, a.x), __fmul_rn(alpha, a.y), __fmul_rn(alpha, a.z), __fmul_rn(alpha, a.w));
  }
}

struct FlashPlan { bool ok; int nt, row_blocks, nsplit, blocks_per_split; int64_t ldm, ldn; };
inline FlashPlan flash_plan(int64_t m, int64_t n, int64_t D) {
  FlashPlan f{false, 0, 0, 1, 0, 0, 0};
  if (D > 1024 || D % 128 != 0) return f;        // accumulators hold 32 x D per workgroup, D = 128 * NT
  if ((n + 128) * D * 4 >= (1ll << 31) || (m + 32) * D * 4 >= (1ll << 31)) return f;   // 32-bit buffer offsets
  f.nt = static_cast<int>(D / 128);
  f.row_blocks = static_cast<int>((m + FBM - 1) / FBM);
  const int64_t col_blocks = (n + FBN - 1) / FBN;
  // Column split: cost model fitted to MI355X timings (us).  One workgroup alone on a CU retires a 128-column
  // block in ~38 us (+ ~5 us of fill/drain), two co-resident ones a pair of blocks in ~66 us (+ ~15 us): two per CU
  // wins in steady state (4096^2: 4 splits), one per CU wins when a workgroup only has a block or two
  // (1200^2: 5 splits -> 190 workgroups instead of 10 -> 380 on 256 CUs).  Partial-output traffic is charged at 4 TB/s.
  double best = 1e30;
  int64_t best_ns = 1;
  for (int64_t ns = 1; ns <= col_blocks && ns <= 64; ++ns) {
    const int64_t bps = (col_blocks + ns - 1) / ns;
    const int64_t real_ns = (col_blocks + bps - 1) / bps;
    if (real_ns != ns) continue;
    const int64_t W = static_cast<int64_t>(f.row_blocks) * ns;
    double t;
    if (W <= 256) t = 5.0 + 38.0 * bps;
    else t = 15.0 + 66.0 * bps * ((W + 511) / 512);
    if (ns > 1) t += static_cast<double>(ns) * m * D * 8.0 / 4.0e6;
    if (t < best) { best = t; best_ns = ns; }
  }
  f.blocks_per_split = static_cast<int>((col_blocks + best_ns - 1) / best_ns);
  f.nsplit = static_cast<int>(best_ns);
  f.ldm = static_cast<int64_t>(f.row_blocks) * FBM;
  f.ldn = col_blocks * FBN;
  f.ok = true;
  return f;
}
inline size_t flash_copy_floats(const FlashPlan& f, int64_t D) { return static_cast<size_t>(D) * (f.ldm + f.ldn); }

// Streaming row statistics: used once the problem is MFMA-sized (the small-batch path and the latency-bound
// split-K form cover everything below); any D (the k-major copies are zero-padded to a multiple of 16).
struct StreamPlan { bool ok; int rt, row_blocks, nsplit, blocks_per_split; int64_t ldm, ldn, kpad; int ks; };
inline StreamPlan stream_plan(int64_t m, int64_t n, int64_t D, bool always = false) {
  StreamPlan f{false, 1, 0, 1, 0, 0, 0, 0, 1};
  static const bool off = getenv("DALM_SIM_ROWSTATS") && getenv("DALM_SIM_ROWSTATS")[0] == 'g';   // "gemm": the LDS-tiled form
  if (off && !always) return f;
  if (!always && (m * n < 512 * 512 || D < 64)) return f;
  static const int force_rt = getenv("DALM_STREAM_RT") ? atoi(getenv("DALM_STREAM_RT")) : 0;
  static const int force_ks = getenv("DALM_STREAM_KS") ? atoi(getenv("DALM_STREAM_KS")) : 0;
  f.rt = force_rt ? force_rt : ((m >= 4096) ? 2 : 1);   // measured: 2048^2 86 vs 77 TF, 4096^2 100 vs 107, 16384^2 118 vs 141
  const int64_t bm = 32 * f.rt;
  f.row_blocks = static_cast<int>((m + bm - 1) / bm);
  // K slices per column tile, for the sizes whose 128-column blocks give the chip only one or two workgroups per CU
  // (grid = row blocks x column blocks; measured, whole call, us: 768^2 (144) 36.0 / 37.7 / 37.5 for 1 / 2 / 4 slices,
  // 1200^2 (380) 57.2 / 54.0 / 56.2, 1536^2 (576) 74.3 / 76.4 / 68.1, 2048^2 (1024) 90.8 / 103.6 / 108.2).  What did NOT
  // move these sizes (profiles/history/r03_sim_midsize_experiments.txt): 16-byte operand loads from 4-k granule copies (slower),
  // capping workgroups per CU through LDS padding (slower), XCD rectangles + an L2 warm-up pass (no change).
  if (!always && f.rt == 1) {
    const int64_t g128 = static_cast<int64_t>(f.row_blocks) * ((n + 127) / 128);
    f.ks = (g128 >= 450 && g128 < 640) ? 4 : (g128 >= 300 && g128 < 450) ? 2 : 1;
    if (force_ks == 1 || force_ks == 2 || force_ks == 4) f.ks = force_ks;
  }
  f.kpad = (D + 16 * f.ks - 1) / (16 * f.ks) * (16 * f.ks);
  if ((n + 128) * f.kpad * 4 >= (1ll << 31) || (m + 64) * f.kpad * 4 >= (1ll << 31)) return f;   // 32-bit buffer offsets
  const int64_t col_blocks = (n + FBN / f.ks - 1) / (FBN / f.ks);
  int64_t ns = ((f.rt == 2 ? 512 : 768) + f.row_blocks - 1) / f.row_blocks;      // 2-3 workgroups per CU (by registers)
  if (ns > col_blocks) ns = col_blocks;
  if (ns < 1) ns = 1;
  if (ns > 64) ns = 64;
  f.blocks_per_split = static_cast<int>((col_blocks + ns - 1) / ns);
  f.nsplit = static_cast<int>((col_blocks + f.blocks_per_split - 1) / f.blocks_per_split);
  f.ldm = static_cast<int64_t>(f.row_blocks) * bm;
  f.ldn = (n + FBN - 1) / FBN * FBN;
  f.ok = true;
  return f;
}

inline bool vec_ok(const float* p, int64_t ld) {
  return (reinterpret_cast<uintptr_t>(p) % 16 == 0) && (ld % 4 == 0);
}

template <int BM, int BN, int BK, int EPI>
void launch_gemm_tile(bool a_kc, bool b_kc, GemmParams p, hipStream_t s, int splitk = 1) {
  p.tiles_n = static_cast<unsigned>((p.N + BN - 1) / BN);
  p.k_chunk = (splitk <= 1) ? p.K : ((p.K + splitk - 1) / splitk + BK - 1) / BK * BK;
  const dim3 grid(p.tiles_n * static_cast<unsigned>((p.M + BM - 1) / BM), static_cast<unsigned>(splitk <= 1 ? 1 : splitk));
  if (a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, BK, true, true, EPI>), grid, dim3(256), 0, s, p);
  else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, BK, true, false, EPI>), grid, dim3(256), 0, s, p);
  else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, BK, false, true, EPI>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gemm_f32_mfma_kernel<BM, BN, BK, false, false, EPI>), grid, dim3(256), 0, s, p);
}

// 128x128 tiles once the grid fills the chip, 64x64 below that.
inline bool use_big_tiles(int64_t M, int64_t N) {
  return ((M + 127) / 128) * ((N + 127) / 128) >= 128;
}
inline int64_t rowstats_parts(int64_t m, int64_t n) {
  const int64_t bn = use_big_tiles(m, n) ? 128 : 64;
  return 2 * ((n + bn - 1) / bn);
}

inline int64_t round_up4(int64_t x) { return (x + 3) / 4 * 4; }

// Split-K for S = A.B^T problems that cannot fill 256 CUs with 64x64 tiles (the real batch sizes: 18 ... ~1200
// rows): SK partial slabs + a one-block-per-row epilogue.  Returns 1 when the direct kernels are used.
inline int sim_splitk(int64_t m, int64_t n, int64_t D) {
  if (use_big_tiles(m, n)) return 1;
  const int64_t tiles = ((m + 63) / 64) * ((n + 63) / 64);
  if (tiles > 512 || D < 256 || n > 8192) return 1;
  int sk = 1;
  while (sk < 8 && tiles * sk * 2 <= 1024 && D / (sk * 2) >= BK_SMALL) sk *= 2;
  return sk;
}

template <int EPI>
void launch_gemm(bool a_kc, bool b_kc, const GemmParams& p, hipStream_t s) {
  if (use_big_tiles(p.M, p.N)) launch_gemm_tile<128, 128, BK_BIG, EPI>(a_kc, b_kc, p, s);
  else launch_gemm_tile<64, 64, BK_SMALL, EPI>(a_kc, b_kc, p, s);
}

int check_gemm_dims(int64_t M, int64_t N, int64_t K, const char* fn) {
  if (M <= 0 || N <= 0 || K <= 0) return fail(DALM_E_SHAPE, fn, "dimensions must be positive");
  if (M > 0x7fffffffll - 256 || N > 0x7fffffffll - 256 || K > 0x7fffffffll - 256)
    return fail(DALM_E_SHAPE, fn, "dimension exceeds int32 range");
  if (((M + 63) / 64) * ((N + 63) / 64) > 0x7fffffffll) return fail(DALM_E_SHAPE, fn, "too many tiles");
  return 0;
}

}  // namespace
}  // namespace dalm

using namespace dalm;

extern "C" int dalm_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                             const float* A, int64_t lda, const float* Bm, int64_t ldb, float* C,
                             int64_t ldc, dalm_stream_t stream) {
  DALM_REQUIRE(A && Bm && C, DALM_E_NULL, "null pointer argument");
  if (int e = check_gemm_dims(M, N, K, __func__)) return e;
  DALM_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, DALM_E_SHAPE,
               "leading dimension too small");
  GemmParams p{};
  p.A = A; p.lda = lda; p.B = Bm; p.ldb = ldb;
  p.M = static_cast<int>(M); p.N = static_cast<int>(N); p.K = static_cast<int>(K);
  p.alpha = alpha; p.a_vec = vec_ok(A, lda); p.b_vec = vec_ok(Bm, ldb);
  p.C = C; p.ldc = ldc;
  // A k-contiguous <=> not transposed; B k-contiguous <=> transposed ([N,K])
  launch_gemm<EPI_STORE>(!transA, transB != 0, p, as_stream(stream));
  return check_launch(__func__);
}

extern "C" int dalm_sim_matmul(const float* A, const float* Bm, int64_t m, int64_t n, int64_t D,
                               float scale, float* S, int64_t ldS, dalm_stream_t stream) {
  return dalm_gemm_f32(0, 1, m, n, D, scale, A, D, Bm, D, S, ldS, stream);
}

// Problems large enough for the bf16x3 form (csrc/lmhead.hip: 256 x 256 tiles, a few waves of them over the 256 CUs) go to
// the bf16 matrix cores; DALM_SIM_BF16X3=0 keeps everything on the f32 MFMA kernels, DALM_SIM_BF16X3_MIN moves the threshold.
static bool use_bf16x3(int64_t m, int64_t n, int64_t D, const float* A, const float* Bm) {
  static const int min_rows = [] {
    const char* off = getenv("DALM_SIM_BF16X3");
    if (off && off[0] == '0') return -1;
    const char* e = getenv("DALM_SIM_BF16X3_MIN");
    return e ? atoi(e) : 3072;   // measured crossover: 2048^2 64 vs 100 TF (f32 wins), 3072^2 142 vs 122 TF (profiles/history/r04_sim_midsize_merge_inlaunch.txt)
  }();
  if (min_rows < 0 || m < min_rows || n < min_rows) return false;
  if (A && (reinterpret_cast<uintptr_t>(A) % 16 || reinterpret_cast<uintptr_t>(Bm) % 16)) return false;
  return dalm_sim_rowstats_bf16x3_supported(m, n, D) != 0;
}

static size_t rowstats_ws_f32(int64_t m, int64_t n, int64_t D) {
  if (const StreamPlan f = stream_plan(m, n, D); f.ok)   // k-major copies | (max, sum) granules [nsplit][m] (8 B each) | tickets
    return (static_cast<size_t>(f.kpad) * (f.ldm + f.ldn) + 2 * static_cast<size_t>(f.nsplit) * m + f.row_blocks + 2) * sizeof(float);
  const int sk = sim_splitk(m, n, D);
  if (sk > 1) return static_cast<size_t>(sk) * static_cast<size_t>(m) * static_cast<size_t>(round_up4(n)) * sizeof(float);
  return static_cast<size_t>(rowstats_parts(m, n)) * static_cast<size_t>(m) * 2 * sizeof(float);
}

extern "C" size_t dalm_sim_rowstats_workspace_bytes(int64_t m, int64_t n, int64_t D) {
  if (m <= 0 || n <= 0) return 0;
  const size_t f32 = rowstats_ws_f32(m, n, D);
  if (!use_bf16x3(m, n, D, nullptr, nullptr)) return f32;
  const size_t x3 = dalm_sim_rowstats_bf16x3_workspace_bytes(m, n, D);   // an unaligned operand still takes the f32 kernel
  return x3 > f32 ? x3 : f32;
}

extern "C" int dalm_sim_rowstats(const float* A, const float* Bm, int64_t m, int64_t n, int64_t D,
                                 float scale, int64_t diag_offset, float* row_lse, float* diag, void* ws,
                                 size_t ws_bytes, dalm_stream_t stream) {
  DALM_REQUIRE(A && Bm && row_lse && diag && ws, DALM_E_NULL, "null pointer argument");
  if (int e = check_gemm_dims(m, n, D, __func__)) return e;
  DALM_REQUIRE(ws_bytes >= dalm_sim_rowstats_workspace_bytes(m, n, D), DALM_E_WORKSPACE, "workspace too small");
  if (use_bf16x3(m, n, D, A, Bm))
    return dalm_sim_rowstats_bf16x3(A, Bm, m, n, D, scale, diag_offset, row_lse, diag, ws, ws_bytes, stream);
  return dalm_sim_rowstats_f32(A, Bm, m, n, D, scale, diag_offset, row_lse, diag, ws, ws_bytes, stream);
}

extern "C" int dalm_sim_rowstats_f32(const float* A, const float* Bm, int64_t m, int64_t n, int64_t D,
                                     float scale, int64_t diag_offset, float* row_lse, float* diag, void* ws,
                                     size_t ws_bytes, dalm_stream_t stream) {
  DALM_REQUIRE(A && Bm && row_lse && diag && ws, DALM_E_NULL, "null pointer argument");
  if (int e = check_gemm_dims(m, n, D, __func__)) return e;
  DALM_REQUIRE(diag_offset >= 0 && diag_offset + m <= n, DALM_E_SHAPE, "diag_offset + m must be <= n");
  DALM_REQUIRE(ws_bytes >= rowstats_ws_f32(m, n, D), DALM_E_WORKSPACE, "workspace too small");
  DALM_REQUIRE(reinterpret_cast<uintptr_t>(ws) % 4 == 0, DALM_E_ALIGN, "workspace must be 4-byte aligned");
  hipStream_t s = as_stream(stream);
  if (const StreamPlan f = stream_plan(m, n, D); f.ok) {
    float* At = static_cast<float*>(ws);
    float* Bt = At + static_cast<size_t>(f.kpad) * f.ldm;
    float* pg = Bt + static_cast<size_t>(f.kpad) * f.ldn;
    if (reinterpret_cast<uintptr_t>(pg) % 8) ++pg;                                   // 8-byte granules
    unsigned* tickets = reinterpret_cast<unsigned*>(pg + 2 * static_cast<size_t>(f.nsplit) * m);
    const int64_t ldmax = f.ldm > f.ldn ? f.ldm : f.ldn;
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(static_cast<unsigned>(ldmax / 32), static_cast<unsigned>(f.kpad / 32 + (f.kpad % 32 != 0)), 2),
                       dim3(256), 0, s, A, static_cast<int>(m), static_cast<int>(f.ldm), At, Bm, static_cast<int>(n),
                       static_cast<int>(f.ldn), Bt, static_cast<int>(D), static_cast<int>(f.kpad), tickets, f.row_blocks);
    StreamStatsParams q{};
    q.At = At; q.Bt = Bt; q.m = static_cast<int>(m); q.n = static_cast<int>(n); q.Kpad = static_cast<int>(f.kpad);
    q.ldm = static_cast<int>(f.ldm); q.ldn = static_cast<int>(f.ldn); q.alpha = scale; q.diag_offset = diag_offset;
    q.part = reinterpret_cast<unsigned long long*>(pg); q.tickets = tickets; q.row_lse = row_lse; q.diag = diag;
    q.row_blocks = f.row_blocks; q.nsplit = f.nsplit; q.blocks_per_split = f.blocks_per_split;
    const dim3 grid(static_cast<unsigned>(f.row_blocks * f.nsplit));
    if (f.rt == 2) hipLaunchKernelGGL((sim_rowstats_stream_kernel<2, MODE_STATS>), grid, dim3(256), 0, s, q);
    else if (f.ks == 4) hipLaunchKernelGGL((sim_rowstats_stream_kernel<1, MODE_STATS, 4>), grid, dim3(256), 0, s, q);
    else if (f.ks == 2) hipLaunchKernelGGL((sim_rowstats_stream_kernel<1, MODE_STATS, 2>), grid, dim3(256), 0, s, q);
    else hipLaunchKernelGGL((sim_rowstats_stream_kernel<1, MODE_STATS>), grid, dim3(256), 0, s, q);
    return check_launch(__func__);      // (the merge of the column splits happens inside the kernel since round 4)
  }
  const int64_t P = rowstats_parts(m, n);
  GemmParams p{};
  p.A = A; p.lda = D; p.B = Bm; p.ldb = D;
  p.M = static_cast<int>(m); p.N = static_cast<int>(n); p.K = static_cast<int>(D);
  p.alpha = scale; p.a_vec = vec_ok(A, D); p.b_vec = vec_ok(Bm, D);
  if (const int sk = sim_splitk(m, n, D); sk > 1) {
    p.C = static_cast<float*>(ws); p.ldc = round_up4(n);
    launch_gemm_tile<64, 64, BK_SMALL, EPI_PARTIAL>(true, true, p, s, sk);
    hipLaunchKernelGGL(splitk_rowstats_kernel, dim3(static_cast<unsigned>(m)), dim3(256), 0, s, p.C, sk, p.M, p.N,
                       p.ldc, scale, diag_offset, row_lse, diag);
    return check_launch(__func__);
  }
  p.part_m = static_cast<float*>(ws);
  p.part_l = p.part_m + P * m;
  p.diag = diag; p.diag_offset = diag_offset;
  launch_gemm<EPI_ROWSTATS>(true, true, p, s);
  hipLaunchKernelGGL(rowstats_merge_kernel, dim3(static_cast<unsigned>((m + 255) / 256)), dim3(256), 0, s,
                     p.part_m, p.part_l, static_cast<int>(P), p.M, row_lse);
  return check_launch(__func__);
}


// Which form dalm_sim_grad takes: 1 = flash (no dS panel), 0 = dS panel + NN GEMM (D > 1024 or odd D, or
// DALM_SIM_GRAD=panel for A/B comparisons).
static bool use_flash_grad(int64_t m, int64_t n, int64_t D) {
  static const int forced = [] {
    const char* e = getenv("DALM_SIM_GRAD");
    if (!e) return -1;
    return (e[0] == 'p') ? 0 : 1;
  }();
  if (forced == 0) return false;
  return flash_plan(m, n, D).ok;
}

extern "C" size_t dalm_sim_grad_workspace_bytes(int64_t m, int64_t n, int64_t D) {
  if (m <= 0 || n <= 0) return 0;
  if (use_flash_grad(m, n, D)) {
    const FlashPlan f = flash_plan(m, n, D);   // k-major operand copies (+ per-split partial outputs)
    return (flash_copy_floats(f, D) + (f.nsplit > 1 ? static_cast<size_t>(f.nsplit) * m * D : 0)) * sizeof(float);
  }
  const size_t panel = static_cast<size_t>(m) * static_cast<size_t>(round_up4(n)) * sizeof(float);
  const int sk = sim_splitk(m, n, D);
  return panel * static_cast<size_t>(sk > 1 ? 1 + sk : 1);  // dS panel (+ split-K slabs)
}

template <int NT>
static void launch_flash(const FlashParams& p, hipStream_t s) {
  hipLaunchKernelGGL((sim_flash_grad_kernel<NT>), dim3(static_cast<unsigned>(p.row_blocks * p.nsplit)), dim3(256), 0, s, p);
}

extern "C" int dalm_sim_grad(const float* A, const float* Bm, int64_t m, int64_t n, int64_t D, float scale,
                             int64_t diag_offset, const float* row_coef, const float* row_lse,
                             const float* col_coef, const float* col_lse, float* dA, void* ws,
                             size_t ws_bytes, dalm_stream_t stream) {
  DALM_REQUIRE(A && Bm && row_coef && row_lse && col_coef && col_lse && dA && ws, DALM_E_NULL,
               "null pointer argument");
  if (int e = check_gemm_dims(m, n, D, __func__)) return e;
  DALM_REQUIRE(diag_offset >= 0 && diag_offset + m <= n, DALM_E_SHAPE, "diag_offset + m must be <= n");
  DALM_REQUIRE(ws_bytes >= dalm_sim_grad_workspace_bytes(m, n, D), DALM_E_WORKSPACE, "workspace too small");
  DALM_REQUIRE(reinterpret_cast<uintptr_t>(ws) % 16 == 0, DALM_E_ALIGN, "workspace must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  if (use_flash_grad(m, n, D)) {
    DALM_REQUIRE(reinterpret_cast<uintptr_t>(dA) % 16 == 0, DALM_E_ALIGN, "dA must be 16-byte aligned");
    const FlashPlan f = flash_plan(m, n, D);
    float* At = static_cast<float*>(ws);
    float* Bt = At + static_cast<size_t>(D) * f.ldm;
    float* slabs = Bt + static_cast<size_t>(D) * f.ldn;
    const int64_t ldmax = f.ldm > f.ldn ? f.ldm : f.ldn;
    hipLaunchKernelGGL(transpose_pad_kernel, dim3(static_cast<unsigned>(ldmax / 32), static_cast<unsigned>((D + 31) / 32), 2),
                       dim3(256), 0, s, A, static_cast<int>(m), static_cast<int>(f.ldm), At, Bm, static_cast<int>(n),
                       static_cast<int>(f.ldn), Bt, static_cast<int>(D), static_cast<int>(D));
    FlashParams p{};
    p.At = At; p.Bt = Bt; p.B = Bm; p.m = static_cast<int>(m); p.n = static_cast<int>(n); p.D = static_cast<int>(D);
    p.ldm = static_cast<int>(f.ldm); p.ldn = static_cast<int>(f.ldn);
    p.alpha = scale; p.diag_offset = diag_offset;
    p.row_coef = row_coef; p.row_lse = row_lse; p.col_coef = col_coef; p.col_lse = col_lse;
    p.out = f.nsplit > 1 ? slabs : dA;
    p.row_blocks = f.row_blocks; p.nsplit = f.nsplit; p.blocks_per_split = f.blocks_per_split;
    p.debug_hot = getenv("DALM_FLASH_DEBUG") ? 1 : 0;
    switch (f.nt) {
      case 1: launch_flash<1>(p, s); break;
      case 2: launch_flash<2>(p, s); break;
      case 3: launch_flash<3>(p, s); break;
      case 4: launch_flash<4>(p, s); break;
      case 5: launch_flash<5>(p, s); break;
      case 6: launch_flash<6>(p, s); break;
      case 7: launch_flash<7>(p, s); break;
      default: launch_flash<8>(p, s); break;
    }
    if (f.nsplit > 1) {
      const int64_t n4 = m * D / 4;
      int64_t blocks = (n4 + 255) / 256;
      if (blocks > 2048) blocks = 2048;
      hipLaunchKernelGGL(flash_reduce_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                         reinterpret_cast<const float4*>(slabs), f.nsplit, n4, scale, reinterpret_cast<float4*>(dA));
    }
    return check_launch(__func__);
  }
  const int64_t ldd = round_up4(n);
  float* dS = static_cast<float*>(ws);
  {  // dS panel: recompute S tiles on the MFMA, transform in the epilogue
    GemmParams p{};
    p.A = A; p.lda = D; p.B = Bm; p.ldb = D;
    p.M = static_cast<int>(m); p.N = static_cast<int>(n); p.K = static_cast<int>(D);
    p.alpha = scale; p.a_vec = vec_ok(A, D); p.b_vec = vec_ok(Bm, D);
    if (const int sk = sim_splitk(m, n, D); sk > 1) {
      float* part = dS + m * ldd;  // slabs live behind the dS panel
      p.C = part; p.ldc = ldd;
      launch_gemm_tile<64, 64, BK_SMALL, EPI_PARTIAL>(true, true, p, s, sk);
      hipLaunchKernelGGL(splitk_ds_kernel, dim3(static_cast<unsigned>(m)), dim3(256), 0, s, part, sk, p.M, p.N, ldd,
                         scale, diag_offset, row_coef, row_lse, col_coef, col_lse, dS, ldd);
    } else {
      p.C = dS; p.ldc = ldd; p.diag_offset = diag_offset;
      p.row_coef = row_coef; p.row_lse = row_lse; p.col_coef = col_coef; p.col_lse = col_lse;
      launch_gemm<EPI_DS>(true, true, p, s);
    }
  }
  {  // dA[m,D] = scale * dS[m,n] . B[n,D]
    GemmParams p{};
    p.A = dS; p.lda = ldd; p.B = Bm; p.ldb = D;
    p.M = static_cast<int>(m); p.N = static_cast<int>(D); p.K = static_cast<int>(n);
    p.alpha = scale; p.a_vec = vec_ok(dS, ldd); p.b_vec = vec_ok(Bm, D);
    p.C = dA; p.ldc = D;
    launch_gemm<EPI_STORE>(true, false, p, s);
  }
  return check_launch(__func__);
}

extern "C" int dalm_nt_xent_fwd(const float* S, int64_t n, int64_t stride_r, int64_t stride_c, float* loss,
                                float* row_lse, dalm_stream_t stream) {
  DALM_REQUIRE(S && loss && row_lse, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(n > 0 && n <= 0x7fffffffll, DALM_E_SHAPE, "n must be positive");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(rows_lse_kernel, dim3(static_cast<unsigned>(n)), dim3(256), 0, s, S, static_cast<int>(n),
                     stride_r, stride_c, row_lse, static_cast<float*>(nullptr));
  hipLaunchKernelGGL(nt_xent_reduce_kernel, dim3(1), dim3(256), 0, s, S, static_cast<int>(n), stride_r,
                     stride_c, row_lse, loss);
  return check_launch(__func__);
}

extern "C" int dalm_nt_xent_bwd(const float* S, int64_t n, int64_t stride_r, int64_t stride_c,
                                const float* row_lse, const float* gscale, float* dS, int accumulate,
                                dalm_stream_t stream) {
  DALM_REQUIRE(S && row_lse && gscale && dS, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(n > 0 && n <= 0x7fffffffll, DALM_E_SHAPE, "n must be positive");
  hipLaunchKernelGGL(rows_softmax_grad_kernel, dim3(static_cast<unsigned>(n)), dim3(256), 0, as_stream(stream),
                     S, static_cast<int>(n), stride_r, stride_c, row_lse, static_cast<const float*>(nullptr),
                     gscale, static_cast<float>(n), 1.f, dS, stride_r, stride_c, accumulate);
  return check_launch(__func__);
}

extern "C" int dalm_doc_logprob_fwd(const float* S, int64_t n, int64_t ldS, float* doc_lp, float* row_lse,
                                    dalm_stream_t stream) {
  DALM_REQUIRE(S && doc_lp && row_lse, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(n > 0 && n <= 0x7fffffffll && ldS >= n, DALM_E_SHAPE, "need n>0, ldS>=n");
  hipLaunchKernelGGL(rows_lse_kernel, dim3(static_cast<unsigned>(n)), dim3(256), 0, as_stream(stream), S,
                     static_cast<int>(n), ldS, static_cast<int64_t>(1), row_lse, doc_lp);
  return check_launch(__func__);
}

extern "C" int dalm_doc_logprob_bwd(const float* S, int64_t n, int64_t ldS, const float* row_lse,
                                    const float* coef, float* dS, int64_t lddS, int accumulate,
                                    dalm_stream_t stream) {
  DALM_REQUIRE(S && row_lse && coef && dS, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(n > 0 && n <= 0x7fffffffll && ldS >= n && lddS >= n, DALM_E_SHAPE, "need n>0, ld>=n");
  // d doc_lp[b] / dS[b,j] = [j==b] - softmax_row(S)[b,j]  => sign = -1 on (softmax - onehot)
  hipLaunchKernelGGL(rows_softmax_grad_kernel, dim3(static_cast<unsigned>(n)), dim3(256), 0, as_stream(stream),
                     S, static_cast<int>(n), ldS, static_cast<int64_t>(1), row_lse, coef,
                     static_cast<const float*>(nullptr), 1.f, -1.f, dS, lddS, static_cast<int64_t>(1),
                     accumulate);
  return check_launch(__func__);
}

extern "C" int dalm_contrastive_finalize(const float* row_lse, const float* col_lse, const float* diag,
                                         int64_t n_local, int64_t n_global, float* out, float* doc_lp,
                                         dalm_stream_t stream) {
  DALM_REQUIRE(row_lse && col_lse && diag && out, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(n_local > 0 && n_global >= n_local && n_local <= 0x7fffffffll, DALM_E_SHAPE,
               "need 0 < n_local <= n_global");
  hipLaunchKernelGGL(contrastive_finalize_kernel, dim3(1), dim3(256), 0, as_stream(stream), row_lse, col_lse,
                     diag, static_cast<int>(n_local), static_cast<float>(n_global), out, doc_lp);
  return check_launch(__func__);
}


// ---- exact top-k (eval retrieval): scores = scale * Q . C^T, k largest per query, no score matrix -------------
namespace {
// First pass (one maximum per row and 32 corpus columns) on the bf16 matrix cores at f32 accuracy (csrc/lmhead.hip, round 4)
// once the score matrix has >= 1024 tiles of 256 x 256: 1.6x the f32 MFMA kernel's rate there.  DALM_TOPK_BF16X3 = 0 / 1
// forces the f32 / bf16x3 pass (read per call: tests switch it).
inline bool topk_use_x3(int64_t m, int64_t n, int64_t D) {
  if (!dalm_sim_rowstats_bf16x3_supported(m, n, D)) return false;
  if (const char* e = getenv("DALM_TOPK_BF16X3")) return e[0] == '1';
  return ((m + 255) / 256) * ((n + 255) / 256) >= 1024;
}
struct TopkLayout { StreamPlan f; int cap; size_t at, bt, gmax, thr, x3, x3_bytes, total; };
inline TopkLayout topk_layout(int64_t m, int64_t n, int64_t D, int64_t k) {
  TopkLayout L{};
  L.f = stream_plan(m, n, D, true);
  L.cap = static_cast<int>(8 * k + 64);
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 15) / 16 * 16; return at; };
  L.at = take(static_cast<size_t>(L.f.kpad) * L.f.ldm * 4);
  L.bt = take(static_cast<size_t>(L.f.kpad) * L.f.ldn * 4);
  L.gmax = take(static_cast<size_t>(m) * (L.f.ldn / 32) * 4);
  L.thr = take(static_cast<size_t>(m) * 4);
  // room for the bf16x3 operand images whenever that pass COULD be taken (the choice may be forced per call)
  L.x3_bytes = dalm_sim_rowstats_bf16x3_supported(m, n, D) ? dalm_x3_group_max_workspace_bytes(m, n, D) : 0;
  L.x3 = take(L.x3_bytes);
  L.total = o;
  return L;
}
}  // namespace

extern "C" int dalm_sim_topk_supported(int64_t D, int64_t k) {
  if (D <= 0 || k <= 0 || k > 1024) return 0;
  const size_t kpad = static_cast<size_t>((D + 15) / 16 * 16);       // as stream_plan pads the embedding width
  return (kpad + 3 * static_cast<size_t>(8 * k + 64)) * 4 <= 60 * 1024;
}

extern "C" size_t dalm_sim_topk_workspace_bytes(int64_t m, int64_t n, int64_t D, int64_t k) {
  if (m <= 0 || n <= 0 || D <= 0 || k <= 0) return 0;
  const TopkLayout L = topk_layout(m, n, D, k);
  return L.f.ok ? L.total : 0;
}

extern "C" int dalm_sim_topk(const float* Q, const float* C, int64_t m, int64_t n, int64_t D, float scale, int64_t k,
                             float* out_val, int64_t* out_idx, int* overflow, void* ws, size_t ws_bytes,
                             dalm_stream_t stream) {
  DALM_REQUIRE(Q && C && out_val && out_idx && overflow && ws, DALM_E_NULL, "null pointer argument");
  if (int e = check_gemm_dims(m, n, D, __func__)) return e;
  DALM_REQUIRE(k > 0 && k <= n && k <= 1024, DALM_E_SHAPE, "need 0 < k <= min(n, 1024)");
  const TopkLayout L = topk_layout(m, n, D, k);
  DALM_REQUIRE(L.f.ok, DALM_E_SHAPE, "corpus block too large for 32-bit buffer offsets: search it in blocks of <= 2^19 rows");
  DALM_REQUIRE(ws_bytes >= L.total, DALM_E_WORKSPACE, "workspace too small");
  DALM_REQUIRE(reinterpret_cast<uintptr_t>(ws) % 16 == 0, DALM_E_ALIGN, "workspace must be 16-byte aligned");
  // validated BEFORE anything is enqueued (dalm_sim_topk_supported is the same test for callers that want to fall back)
  DALM_REQUIRE(dalm_sim_topk_supported(D, k), DALM_E_SHAPE,
               "D and k too large for the refine kernel's LDS (padded D + 3*(8k+64) floats <= 15360)");
  hipStream_t s = as_stream(stream);
  char* base = static_cast<char*>(ws);
  float* At = reinterpret_cast<float*>(base + L.at);
  float* Bt = reinterpret_cast<float*>(base + L.bt);
  const StreamPlan& f = L.f;
  const int64_t ldmax = f.ldm > f.ldn ? f.ldm : f.ldn;
  hipLaunchKernelGGL(transpose_pad_kernel, dim3(static_cast<unsigned>(ldmax / 32), static_cast<unsigned>((f.kpad + 31) / 32), 2),
                     dim3(256), 0, s, Q, static_cast<int>(m), static_cast<int>(f.ldm), At, C, static_cast<int>(n),
                     static_cast<int>(f.ldn), Bt, static_cast<int>(D), static_cast<int>(f.kpad));
  StreamStatsParams q{};
  q.At = At; q.Bt = Bt; q.m = static_cast<int>(m); q.n = static_cast<int>(n); q.Kpad = static_cast<int>(f.kpad);
  q.ldm = static_cast<int>(f.ldm); q.ldn = static_cast<int>(f.ldn); q.alpha = scale;
  q.row_blocks = f.row_blocks; q.nsplit = f.nsplit; q.blocks_per_split = f.blocks_per_split;
  q.gmax = reinterpret_cast<float*>(base + L.gmax); q.ng = static_cast<int>(f.ldn / 32);
  float* thr = reinterpret_cast<float*>(base + L.thr);
  const dim3 grid(static_cast<unsigned>(f.row_blocks * f.nsplit));
  if (hipError_t e = hipMemsetAsync(overflow, 0, 4, s); e != hipSuccess) return static_cast<int>(e);
  float abs_slack = 2.4e-7f * sqrtf(static_cast<float>(f.kpad)) * fabsf(scale);
  if (topk_use_x3(m, n, D) && reinterpret_cast<uintptr_t>(Q) % 16 == 0 && reinterpret_cast<uintptr_t>(C) % 16 == 0) {
    // the group maxima come from the bf16x3 kernel (every score within 2e-6 |scale| of the exact one): the threshold gives
    // way by twice that on top of the f32 slack, the refine pass below still recomputes the candidates in f32
    if (int e = dalm_x3_group_max(Q, C, m, n, D, scale, q.gmax, q.ng, base + L.x3, L.x3_bytes, stream)) return e;
    abs_slack += 5e-6f * fabsf(scale);
  } else if (f.rt == 2) hipLaunchKernelGGL((sim_rowstats_stream_kernel<2, MODE_GMAX>), grid, dim3(256), 0, s, q);
  else hipLaunchKernelGGL((sim_rowstats_stream_kernel<1, MODE_GMAX>), grid, dim3(256), 0, s, q);
  hipLaunchKernelGGL(topk_threshold_kernel, dim3(static_cast<unsigned>(m)), dim3(256), 0, s, q.gmax, q.ng,
                     static_cast<int>(k), abs_slack, thr);
  const size_t lds = (static_cast<size_t>(f.kpad) + 3 * static_cast<size_t>(L.cap)) * 4;
  hipLaunchKernelGGL(topk_refine_kernel, dim3(static_cast<unsigned>(m)), dim3(256), lds, s, At, static_cast<int>(f.ldm),
                     Bt, static_cast<int>(f.ldn), static_cast<int>(f.kpad), static_cast<int>(n), scale, q.gmax, q.ng, thr,
                     L.cap, static_cast<int>(k), out_val, out_idx, overflow);
  return check_launch(__func__);
}
