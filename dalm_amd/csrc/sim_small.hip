// K2-K4, small-batch form: the in-batch similarity / contrastive loss at the batch sizes the trainers really
// use (18 ... a few hundred rows per GPU), where the whole problem is a few MFLOP and everything is latency.
//
// Stands in for the same reference lines as sim.hip
//   dalm/training/utils/train_utils.py:76-88,124   dalm/training/rag_e2e/train_rage2e.py:441-446
// in THREE launches per step instead of ~10:
//   small_partial_kernel  S is computed ONCE: 32x32 tiles x split-K over ~256 workgroups; operands go
//                         global -> registers in MFMA layout (no LDS staging, no barrier in the K loop: the
//                         K index is permuted identically for both operands, which a dot product allows);
//                         the 4 waves of a workgroup split its K range and are summed through LDS in fixed
//                         order; raw partial tiles go to slabs, row-major and (for column stats) transposed
//   small_stats_kernel    one wave per row of S and per row of S^T: sums the slabs in fixed order, writes S
//                         (<= 4 MB, L2-resident; reused by the backward), row/column log-sum-exp and diag
//   small_grad_kernel     dQ = s dS P and dP = s dS^T Q in ONE launch: every workgroup owns a 32x32 tile of an
//                         output, rebuilds its strip of the closed-form dS from the saved S in LDS, and feeds
//                         v_mfma_f32_32x32x2_f32 with the other operand straight from global memory
// Exact f32 throughout (|S| <= 100, see sim.hip).  Deterministic: fixed summation orders, no float atomics.
// Bound: launch/latency (46 MFLOP and 1.2 MB at 150^2 x 1024); the roofline-relevant similarity kernels are
// the large-batch ones in sim.hip.
#include "common.hpp"

namespace dalm {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float4 ld4_guard(const float* p, int nvalid, bool vec_ok) {
  if (nvalid >= 4 && vec_ok) return *reinterpret_cast<const float4*>(p);
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (nvalid > 0) r.x = p[0];
  if (nvalid > 1) r.y = p[1];
  if (nvalid > 2) r.z = p[2];
  if (nvalid > 3) r.w = p[3];
  return r;
}

// C/D layout of v_mfma_f32_32x32x2_f32: acc[r] of lane l is (row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31)
__device__ __forceinline__ int mfma_row(int r, int lhi) { return (r & 3) + 8 * (r >> 2) + 4 * lhi; }

constexpr int RED_STRIDE = 33;
constexpr int SK_MAX = 16;       // most K slices a tile is split into

// sums the 4 waves' 32x32 accumulators through LDS (fixed order w = 0..3); afterwards thread t owns tile
// elements e = t + 256 q.  red: [4][32][33] floats.
__device__ __forceinline__ void stash_acc(float* red, const f32x16& acc, int wave, int l31, int lhi) {
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 32 + mfma_row(r, lhi)) * RED_STRIDE + l31] = acc[r];
}
__device__ __forceinline__ float red_sum(const float* red, int row, int col) {
  float s = red[row * RED_STRIDE + col];
#pragma unroll
  for (int w = 1; w < 4; ++w) s += red[(w * 32 + row) * RED_STRIDE + col];
  return s;
}

// ---------------------------------------------------------------------------------------------------
// ONE-LAUNCH forward (round 4): the statistics pass moves INTO the partial-tile kernel - the last
// workgroup to arrive finishes what the others left, in a fixed order, so the result does not depend on who was last:
//   level 1  (sk > 1)  every K slice of a tile publishes its 32 x 32 partial (tile-major private slab, 4 KB); the LAST of
//                      the sk slices sums them in slice order z = 0 .. sk-1, scales, writes S / diag and reduces the
//                      finished tile to one (max, sum exp) pair per row (and per column) of the tile;
//   level 2            those pairs are published per tile; the LAST tile of a row strip merges the strip's tiles_n pairs
//                      per row in tile order -> row_lse; likewise the last tile of a column strip -> col_lse.
// Hand-off form (guide section 6, Guideline 16; MI355X: 8 XCDs with private L2s, per-CU L1 never refreshed by other CUs):
// payload with WRITE-THROUGH agent-scope stores (sc1: 4-byte slab words, 8-byte (max, sum) granules - the natural widths
// of this epilogue), every storing wave drains (`s_waitcnt vmcnt(0)`), __syncthreads(), ONE lane draws an agent-scope
// ticket; the last arriver reads the payload with agent-scope (sc1, L1-bypassing) loads - no fences, no polling, no
// spinning.  Every payload line has exactly one writer and one reader and is not read before it is complete, so no stale
// copy can exist in the reader's L2 within the launch; kernel boundaries take care of the previous call's lines.
// Tickets: zero on entry, reset to zero by the last arriver (the caller zeroes them ONCE, at allocation).
// ---------------------------------------------------------------------------------------------------
struct FusedFwd {
  float* S; int64_t ldS;
  float* row_lse; float* diag; float* col_lse;     // col_lse may be null (rows only)
  float alpha; int64_t diag_offset;
  float* tslab;                 // [sk][tiles][1024] partial tiles (unused when sk == 1)
  unsigned long long* rowpart;  // [tiles_n][m_pad] (max, sum exp) granules, m_pad = tiles_m * 32
  unsigned long long* colpart;  // [tiles_m][n_pad]
  unsigned* tickets;            // [tiles] tile tickets | [tiles_m] row-strip tickets | [tiles_n] column-strip tickets
  int sk, tiles_m, tiles_n;
};

// red: the four waves' 32 x 32 accumulators (stash_acc layout), already synchronised.  LDS beyond the first wave's tile is
// reused as scratch once the sums are in registers.
__device__ __forceinline__ void fused_fwd_epilogue(const FusedFwd& p, float* red, int ti, int tj, int z, int m, int n) {
  const int tid = threadIdx.x;
  const int i0 = ti * 32, j0 = tj * 32;
  const int tile = ti * p.tiles_n + tj, tiles = p.tiles_m * p.tiles_n;
  float v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + 256 * q;
    v[q] = red_sum(red, e >> 5, e & 31);
  }
  __syncthreads();                                   // red is free from here on
  unsigned* slot = reinterpret_cast<unsigned*>(red + 3 * 32 * RED_STRIDE);   // ticket broadcast words (3 used)
  if (p.sk > 1) {
    float* mine = p.tslab + (static_cast<int64_t>(z) * tiles + tile) * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) st_agent(mine + tid + 256 * q, v[q]);
    if (draw_ticket(p.tickets + tile, slot) != static_cast<unsigned>(p.sk - 1)) return;
    if (tid == 0) __hip_atomic_store(p.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // every slice's partial, all loads of a thread in flight together, summed in slice order (not arrival order)
    float w[4][SK_MAX];
#pragma unroll
    for (int zz = 0; zz < SK_MAX; ++zz) {
      const float* src = p.tslab + (static_cast<int64_t>(min(zz, p.sk - 1)) * tiles + tile) * 1024;
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q][zz] = ld_agent(src + tid + 256 * q);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float a = w[q][0];
#pragma unroll
      for (int zz = 1; zz < SK_MAX; ++zz) a += (zz < p.sk) ? w[q][zz] : 0.f;
      v[q] = a;
    }
  }
  // ---- the finished tile: S, diag, and its (max, sum exp) per row / per column ----
  float* Ts = red;                                   // [32][33], -inf outside the matrix
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + 256 * q, row = e >> 5, col = e & 31;
    const float sv = __fmul_rn(p.alpha, v[q]);
    const bool in = (i0 + row < m) && (j0 + col < n);
    if (in) {
      p.S[static_cast<int64_t>(i0 + row) * p.ldS + j0 + col] = sv;
      if (static_cast<int64_t>(j0 + col) == p.diag_offset + i0 + row) p.diag[i0 + row] = sv;
    }
    Ts[row * RED_STRIDE + col] = in ? sv : -INFINITY;
  }
  __syncthreads();
  const int m_pad = p.tiles_m * 32, n_pad = p.tiles_n * 32;
  if (tid < 64) {                                    // lanes 0-31: the tile's rows; lanes 32-63: its columns
    const bool colside = tid >= 32;
    const int r = tid & 31;
    if (!colside || p.col_lse) {
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < 32; ++c) mx = fmaxf(mx, colside ? Ts[c * RED_STRIDE + r] : Ts[r * RED_STRIDE + c]);
      float l = 0.f;
      if (mx > -INFINITY) {
#pragma unroll
        for (int c = 0; c < 32; ++c) l += fast_exp((colside ? Ts[c * RED_STRIDE + r] : Ts[r * RED_STRIDE + c]) - mx);
      }
      if (colside) st_pair(p.colpart + static_cast<int64_t>(ti) * n_pad + j0 + r, mx, l);
      else st_pair(p.rowpart + static_cast<int64_t>(tj) * m_pad + i0 + r, mx, l);
    }
  }
  unsigned* t_row = p.tickets + tiles + ti;
  unsigned* t_col = p.tickets + tiles + p.tiles_m + tj;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    slot[1] = __hip_atomic_fetch_add(t_row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    slot[2] = p.col_lse ? __hip_atomic_fetch_add(t_col, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
  }
  __syncthreads();
  const bool last_row = slot[1] == static_cast<unsigned>(p.tiles_n - 1);
  const bool last_col = p.col_lse && slot[2] == static_cast<unsigned>(p.tiles_m - 1);
  if (!last_row && !last_col) return;
  __syncthreads();                                   // slot[] has been read by everyone before the scratch below reuses LDS
  float* pm = red;                                   // [8][32] group maxima, [8][32] group sums
  float* pl = red + 256;
  const int r = tid & 31, g = tid >> 5;
  for (int side = 0; side < 2; ++side) {
    if (!(side ? last_col : last_row)) continue;     // workgroup-uniform
    if (tid == 0) __hip_atomic_store(side ? t_col : t_row, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int parts = side ? p.tiles_m : p.tiles_n;
    const unsigned long long* base = side ? p.colpart + j0 + r : p.rowpart + i0 + r;
    const int64_t stride = side ? n_pad : m_pad;
    float mx = -INFINITY, l = 0.f;
    for (int t0 = g; t0 < parts; t0 += 8 * 4) {      // group g folds tiles g, g + 8, ... in ascending order, 4 loads in flight
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) ld_pair(base + static_cast<int64_t>(min(t0 + 8 * u, parts - 1)) * stride, a[u], b[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (t0 + 8 * u < parts && a[u] > -INFINITY) {
          const float mn = fmaxf(mx, a[u]);
          l = l * fast_exp(mx - mn) + b[u] * fast_exp(a[u] - mn);   // exp(-inf) = 0 on the first fold
          mx = mn;
        }
      }
    }
    pm[g * 32 + r] = mx;
    pl[g * 32 + r] = l;
    __syncthreads();
    if (tid < 32) {
      float M = pm[tid];
#pragma unroll
      for (int gg = 1; gg < 8; ++gg) M = fmaxf(M, pm[gg * 32 + tid]);
      float L = 0.f;
#pragma unroll
      for (int gg = 0; gg < 8; ++gg) L += (pm[gg * 32 + tid] == -INFINITY) ? 0.f : pl[gg * 32 + tid] * fast_exp(pm[gg * 32 + tid] - M);
      const int idx = (side ? j0 : i0) + tid;
      if (idx < (side ? n : m)) (side ? p.col_lse : p.row_lse)[idx] = M + __logf(L);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// partial tiles.  grid (tiles_m * tiles_n, SK); 256 threads; wave w of slice z contracts
//   k in [z*k_chunk + w*k_chunk/4, ... + k_chunk/4)   (k_chunk is a multiple of 32)
// lane (r = l&31, h = l>>5) loads A[i0+r][k+4h .. k+4h+3] and B[j0+r][k+4h .. +3] for k = lo, lo+8, ...;
// MFMA step t of such a pair multiplies A[.][k+4h+t] with B[.][k+4h+t] over h = 0,1: every k exactly once.
// ---------------------------------------------------------------------------------------------------
template <bool FAST, bool FUSED = false>
__global__ __launch_bounds__(256) void small_partial_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                            int m, int n, int K, int k_chunk, int tiles_n,
                                                            int a_vec, int b_vec, float* __restrict__ slab,
                                                            int ldn, float* __restrict__ slabT, int ldm,
                                                            const FusedFwd fz) {
  __shared__ float red[4 * 32 * RED_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n, z = blockIdx.y;
  const int i0 = ti * 32, j0 = tj * 32;
  const int kq = k_chunk >> 2;
  const int k_lo = z * k_chunk + wave * kq;
  const int k_hi = min(K, k_lo + kq);
  const bool arow = (i0 + l31) < m, brow = (j0 + l31) < n;
  const float* ap = A + static_cast<int64_t>(FAST ? min(i0 + l31, m - 1) : i0 + l31) * K;
  const float* bp = B + static_cast<int64_t>(FAST ? min(j0 + l31, n - 1) : j0 + l31) * K;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // 4 pairs of float4 per operand per round (32 of K).  Measured and slower on every shape (round 3): 8 pairs per round
  // (+0.7 ... +1.6 us per call), a wave's whole K range (128 = 4 rounds) in flight before its first MFMA (150 x 1200: 18.9
  // instead of 17.0 us per call, 512^2: 21.3 instead of 20.5), more split-K workgroups (DALM_SMALL_TARGET 384 - 1024), fewer (128)
  constexpr int NU = 4;
  for (int k0 = k_lo; k0 < k_hi; k0 += 8 * NU) {
    float4 av[NU], bv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int k = k0 + 8 * u + 4 * lhi;
      if constexpr (FAST) {
        // K % 8 == 0, 16-byte aligned rows: unconditional loads (rows past m / n are clamped to the last row and
        // produce tile rows that are never stored; k past the range is clamped and its MFMAs are skipped)
        const int kc = min(k, K - 4);
        av[u] = *reinterpret_cast<const float4*>(ap + kc);
        bv[u] = *reinterpret_cast<const float4*>(bp + kc);
      } else {
        const int nv = k_hi - k;
        av[u] = ld4_guard(ap + k, arow ? nv : 0, a_vec);
        bv[u] = ld4_guard(bp + k, brow ? nv : 0, b_vec);
      }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (k0 + 8 * u < k_hi) {  // wave-uniform
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u].w, acc, 0, 0, 0);
      }
    }
  }
  stash_acc(red, acc, wave, l31, lhi);
  __syncthreads();
  if constexpr (FUSED) {            // one-launch forward: the statistics are finished by the last arrivers (see above)
    fused_fwd_epilogue(fz, red, ti, tj, z, m, n);
    return;
  }
  float* sl = slab + static_cast<int64_t>(z) * m * ldn;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + 256 * q, row = e >> 5, col = e & 31;
    if (i0 + row < m && j0 + col < n) sl[static_cast<int64_t>(i0 + row) * ldn + j0 + col] = red_sum(red, row, col);
  }
  if (slabT) {
    float* st = slabT + static_cast<int64_t>(z) * n * ldm;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q, col = e >> 5, row = e & 31;
      if (i0 + row < m && j0 + col < n) st[static_cast<int64_t>(j0 + col) * ldm + i0 + row] = red_sum(red, row, col);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The same partial tiles with the operand loads SOFTWARE-PIPELINED (round 4): the loop above is load -> wait -> 16 MFMAs
// per round of 32 k, i.e. every round pays a full L2 / fabric round trip (the ISA shows s_waitcnt vmcnt(0..2) in front of
// every group of four MFMAs and wave-uniform branches around the later groups).  Here a wave's K range is a compile-time
// number of whole rounds (ROUNDS x 32, no tail: the planner only sends exact covers), DEPTH rounds are in flight, and the
// group of four MFMAs that has just consumed a pair of float4 registers is followed by the loads that refill them for
// round r + DEPTH.  Same summation order as small_partial_kernel (k ascending per wave, waves 0..3 through LDS): same bits.
// ---------------------------------------------------------------------------------------------------
template <int ROUNDS, int DEPTH, bool FUSED = false>
__global__ __launch_bounds__(256) void small_partial_pipe_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                 int m, int n, int K, int k_chunk, int tiles_n,
                                                                 float* __restrict__ slab, int ldn,
                                                                 float* __restrict__ slabT, int ldm, const FusedFwd fz) {
  __shared__ float red[4 * 32 * RED_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n, z = blockIdx.y;
  const int i0 = ti * 32, j0 = tj * 32;
  const int k_lo = z * k_chunk + wave * (ROUNDS * 32) + 4 * lhi;
  // rows past m / n are clamped to the last row: their tile rows are never stored
  const float* ap = A + static_cast<int64_t>(min(i0 + l31, m - 1)) * K + k_lo;
  const float* bp = B + static_cast<int64_t>(min(j0 + l31, n - 1)) * K + k_lo;
  constexpr int NB = DEPTH < ROUNDS ? DEPTH : ROUNDS;
  float4 av[NB][4], bv[NB][4];
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      av[r][u] = *reinterpret_cast<const float4*>(ap + 32 * r + 8 * u);
      bv[r][u] = *reinterpret_cast<const float4*>(bp + 32 * r + 8 * u);
    }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    constexpr int dummy = 0; (void)dummy;
    const int b = r % NB;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b][u].x, bv[b][u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b][u].y, bv[b][u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b][u].z, bv[b][u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b][u].w, bv[b][u].w, acc, 0, 0, 0);
      if (r + NB < ROUNDS) {
        av[b][u] = *reinterpret_cast<const float4*>(ap + 32 * (r + NB) + 8 * u);
        bv[b][u] = *reinterpret_cast<const float4*>(bp + 32 * (r + NB) + 8 * u);
      }
    }
  }
  stash_acc(red, acc, wave, l31, lhi);
  __syncthreads();
  if constexpr (FUSED) {            // one-launch forward: the statistics are finished by the last arrivers (see above)
    fused_fwd_epilogue(fz, red, ti, tj, z, m, n);
    return;
  }
  float* sl = slab + static_cast<int64_t>(z) * m * ldn;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + 256 * q, row = e >> 5, col = e & 31;
    if (i0 + row < m && j0 + col < n) sl[static_cast<int64_t>(i0 + row) * ldn + j0 + col] = red_sum(red, row, col);
  }
  if (slabT) {
    float* st = slabT + static_cast<int64_t>(z) * n * ldm;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q, col = e >> 5, row = e & 31;
      if (i0 + row < m && j0 + col < n) st[static_cast<int64_t>(j0 + col) * ldm + i0 + row] = red_sum(red, row, col);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// row statistics of S (blockIdx.y == 0: slab, also writes S and diag) and of S^T (blockIdx.y == 1: slabT).
// One wave per row; slabs summed in fixed order z = 0..SK-1, then scaled with a separate rounding so that the
// S read by the backward and the S the statistics saw are the same bits.
// ---------------------------------------------------------------------------------------------------

// S value at one position: the SK slab entries are fetched as ONE batch (unconditional, slab index clamped) and
// summed in fixed order z = 0..SK-1 - a `for z < SK` loop of loads compiles to a serial latency chain
// (measured: 6-9 us for this kernel at 18..150 rows, all of it waiting).
__device__ __forceinline__ float slab_s(const float* __restrict__ p, int64_t slab_stride, int SK, float alpha) {
  float v[SK_MAX];
#pragma unroll
  for (int z = 0; z < SK_MAX; ++z) v[z] = p[min(z, SK - 1) * slab_stride];
  float a = v[0];
#pragma unroll
  for (int z = 1; z < SK_MAX; ++z) a += (z < SK) ? v[z] : 0.f;
  return __fmul_rn(alpha, a);
}

__global__ __launch_bounds__(256) void small_stats_kernel(const float* __restrict__ slab, int ldn,
                                                          const float* __restrict__ slabT, int ldm, int SK, int m,
                                                          int n, float alpha, int64_t diag_offset,
                                                          float* __restrict__ S, int64_t ldS,
                                                          float* __restrict__ row_lse, float* __restrict__ diag,
                                                          float* __restrict__ col_lse) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool cols = blockIdx.y != 0;
  const int R = cols ? n : m, C = cols ? m : n, ld = cols ? ldm : ldn;
  const int row = blockIdx.x * 4 + wave;
  if (row >= R) return;
  const float* base = (cols ? slabT : slab) + static_cast<int64_t>(row) * ld;
  const int64_t ss = static_cast<int64_t>(R) * ld;
  float mx = -INFINITY, l = 0.f;
  if (C <= 256) {  // values stay in registers; chunks past C are skipped wave-uniformly
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = -INFINITY;
      if (64 * i < C) {
        const int c = lane + 64 * i;
        const float sv = slab_s(base + min(c, C - 1), ss, SK, alpha);
        if (c < C) {
          v[i] = sv;
          if (!cols) {
            S[static_cast<int64_t>(row) * ldS + c] = sv;
            if (static_cast<int64_t>(c) == diag_offset + row) diag[row] = sv;
          }
        }
      }
      mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
#pragma unroll
    for (int i = 0; i < 4; ++i) l += fast_exp(v[i] - mx);  // exp(-inf) = 0 for the padding
  } else {
    for (int c0 = 0; c0 < C; c0 += 64) {
      const int c = c0 + lane;
      const float sv = slab_s(base + min(c, C - 1), ss, SK, alpha);
      if (c < C) {
        mx = fmaxf(mx, sv);
        if (!cols) {
          S[static_cast<int64_t>(row) * ldS + c] = sv;
          if (static_cast<int64_t>(c) == diag_offset + row) diag[row] = sv;
        }
      }
    }
    mx = wave_max(mx);
    for (int c0 = 0; c0 < C; c0 += 64) {
      const int c = c0 + lane;
      // rows: this lane wrote S[row][c] in the first pass (own writes are visible to the thread) - one load instead of SK
      const float sv = cols ? slab_s(base + min(c, C - 1), ss, SK, alpha) : S[static_cast<int64_t>(row) * ldS + min(c, C - 1)];
      if (c < C) l += fast_exp(sv - mx);
    }
  }
  l = wave_sum(l);
  if (lane == 0) (cols ? col_lse : row_lse)[row] = mx + __logf(l);
}

// Rows longer than 256 (the per-rank blocks of a sharded batch: 150 x 1200, 18 x 144 stays above): ONE WORKGROUP per row of
// S (or of S^T) instead of one wave - 4x the workgroups (150 instead of 38 at 150 x 1200, where the one-wave form took
// 17 us of pure latency) and every thread's SK x 8 slab loads of a batch are issued together.  Each thread folds its values
// into an online (max, sum exp) pair; the 256 pairs are combined through LDS in fixed order.
__global__ __launch_bounds__(256) void small_stats_wide_kernel(const float* __restrict__ slab, int ldn,
                                                               const float* __restrict__ slabT, int ldm, int SK, int m,
                                                               int n, float alpha, int64_t diag_offset,
                                                               float* __restrict__ S, int64_t ldS,
                                                               float* __restrict__ row_lse, float* __restrict__ diag,
                                                               float* __restrict__ col_lse) {
  __shared__ float red[8];
  const int tid = threadIdx.x;
  const bool cols = blockIdx.y != 0;
  const int R = cols ? n : m, C = cols ? m : n, ld = cols ? ldm : ldn;
  const int row = blockIdx.x;
  if (row >= R) return;
  const float* base = (cols ? slabT : slab) + static_cast<int64_t>(row) * ld;
  const int64_t ss = static_cast<int64_t>(R) * ld;
  float mx = -INFINITY, l = 0.f;
  for (int c0 = 0; c0 < C; c0 += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + tid + 256 * u;
      v[u] = slab_s(base + min(c, C - 1), ss, SK, alpha);      // unconditional, clamped: one batch in flight
    }
    float bm = -INFINITY;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + tid + 256 * u;
      if (c < C) {
        bm = fmaxf(bm, v[u]);
        if (!cols) {
          S[static_cast<int64_t>(row) * ldS + c] = v[u];
          if (static_cast<int64_t>(c) == diag_offset + row) diag[row] = v[u];
        }
      } else {
        v[u] = -INFINITY;
      }
    }
    if (bm > -INFINITY) {
      const float mn = fmaxf(mx, bm);
      l *= fast_exp(mx - mn);                                 // exp(-inf) = 0 on the first batch
#pragma unroll
      for (int u = 0; u < 8; ++u) l += fast_exp(v[u] - mn);
      mx = mn;
    }
  }
  const float M = block_max<256>(mx, red);
  const float L = block_sum<256>((mx == -INFINITY) ? 0.f : l * fast_exp(mx - M), red);
  if (tid == 0) (cols ? col_lse : row_lse)[row] = M + __logf(L);
}

// dst[i] = sum_z src[z * stride + i] (fixed order), float4-wide with a scalar tail: combines the contraction slices of
// small_grad_kernel
__global__ __launch_bounds__(256) void small_slice_sum_kernel(const float* __restrict__ src, int nsl, int64_t count,
                                                              float* __restrict__ dst) {
  const int64_t n4 = count >> 2;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += gridDim.x * 256ll) {
    float4 a = reinterpret_cast<const float4*>(src)[i];
    for (int z = 1; z < nsl; ++z) {
      const float4 b = reinterpret_cast<const float4*>(src + z * count)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(dst)[i] = a;
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < count; i += 256) {
      float a = src[i];
      for (int z = 1; z < nsl; ++z) a += src[z * count + i];
      dst[i] = a;
    }
}

// ---------------------------------------------------------------------------------------------------
// backward from the saved S.  grid (ceil(R/32), ceil(D/32), ndir * nsl):
//   dir 0: dA[i0.., d0..] = alpha * sum_j dS[i][j] Bm[j][d]      (R = m, contraction over n)
//   dir 1: dB[j0.., d0..] = alpha * sum_i dS[i][j] A[i][d]       (R = n, contraction over m)
//   dS[i][j] = rc[i] e^{S_ij - rl[i]} + cc[j] e^{S_ij - cl[j]} - [j == off + i] (rc[i] + cc[j])
// The contraction runs in chunks of 256: the 32 x 256 strip of dS is built in LDS (k-major, stride 33:
// conflict-free ds_read_b32 fragments), the 4 waves split the chunk and read the other operand's fragments
// straight from global memory (lane (c,h) reads X[k+h][d0+c]: 128-byte rows, L2-resident), 8 steps in flight.
// ---------------------------------------------------------------------------------------------------
constexpr int GK = 256;

// All loads below are UNCONDITIONAL with clamped indices (a predicated load compiles to a branch plus
// s_waitcnt vmcnt(0), which serialises what should be one batch in flight); out-of-range entries are zeroed by
// value.  Vocabulary: "row" = one of the 32 output rows of this workgroup, "k" = contraction index.
//   dir 0: row = i (query), k = j (passage), S element S[row][k], diag at k == off + row
//   dir 1: row = j (passage), k = i (query),  S element S[k][row], diag at k == row - off
template <bool LONG>
__global__ __launch_bounds__(256, 4) void small_grad_kernel(const float* __restrict__ S, int64_t ldS,
                                                         const float* __restrict__ A, const float* __restrict__ Bm,
                                                         int m, int n, int D, float alpha, int64_t diag_offset,
                                                         const float* __restrict__ rc, const float* __restrict__ rl,
                                                         const float* __restrict__ cc, const float* __restrict__ cl,
                                                         float* __restrict__ dA, float* __restrict__ dB, int dir0, int nsl,
                                                         float* __restrict__ slabA, float* __restrict__ slabB,
                                                         unsigned* __restrict__ tickets) {
  // the wave-sum scratch of the epilogue (4 x 32 x 33 floats) overlays the dS strip, which is dead by then: 34 KB instead of
  // 51 KB of LDS per workgroup = 4 resident workgroups per CU (the VGPR limit) instead of 3
  __shared__ float Ds[GK * RED_STRIDE];
  float* red = Ds;
  static_assert(GK * RED_STRIDE >= 4 * 32 * RED_STRIDE, "the reduction scratch must fit into the strip");
  __shared__ float rowc_s[32], rowl_s[32];
  __shared__ float kc_s[GK], kl_s[GK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  // nsl > 1: the contraction is cut into nsl slices of whole 256-chunks, slice z of direction d is block z = d * nsl + slice
  // and writes its partial tile into slab[slice] - small_slice_sum_kernel adds them in fixed order (the per-rank blocks of a
  // sharded batch have few row tiles and a long contraction: 150 x 1200 was 160 workgroups x 5 chunks, 30 us of latency)
  const int dir = dir0 + static_cast<int>(blockIdx.z) / nsl, slice = static_cast<int>(blockIdx.z) % nsl;
  const int R = dir ? n : m, Kd = dir ? m : n;
  const int r0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
  if (r0 >= R) return;
  const float* X = dir ? A : Bm;                      // the other operand, [Kd, D]
  const float* rowc = dir ? cc : rc; const float* rowl = dir ? cl : rl;
  const float* kc = dir ? rc : cc;   const float* kl = dir ? rl : cl;
  float* out = dir ? dB : dA;
  if (nsl > 1 && !tickets) out = (dir ? slabB : slabA) + static_cast<int64_t>(slice) * R * D;
  const int nchunks = (Kd + GK - 1) / GK, cps = (nchunks + nsl - 1) / nsl;
  const int k_begin = slice * cps * GK, k_end = min(Kd, (slice + 1) * cps * GK);
  const int dcol = min(d0 + l31, D - 1);
  const int64_t doff = dir ? -diag_offset : diag_offset;   // diag at k == row + doff

  if (tid < 32) {
    const int r = min(r0 + tid, R - 1);
    rowc_s[tid] = rowc[r];
    rowl_s[tid] = rowl[r];
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int k0 = k_begin; k0 < k_end; k0 += GK) {
    const int kn = min(GK, Kd - k0);
    // The other operand's fragments (lane (c, h) of step s reads X[k0 + 2s + h][d0 + c]) are independent of dS.
    //   LONG (contractions longer than 64: 32 two-wide steps per wave): ALL of a wave's fragments are fetched up front - the first half
    //     before the strip is built (its S loads, the exps and the barrier hide the latency), the second half once the strip's
    //     32 S values have left their registers (covered by the barrier and the first 16 MFMAs); steps past the chunk's end
    //     multiply the zero rows of the strip.  Before (round 3 measurement): rounds of 8 steps with the next round fetched
    //     during the current round's 8 dependent MFMAs = 0.2 us of cover for an L2 round trip, three exposed stalls per
    //     chunk (512^2: 31.8 -> 28.4 us, 150 x 1200: 20.2 -> 19.1 us per call);
    //   !LONG (the 18-row batches: one or two steps per wave): the round form - 32 unconditional steps per wave cost
    //     those shapes 1.1 us.
    constexpr int SPW = GK / 8;            // LONG: steps per wave
    const int ns = (kn + 1) >> 1;          // two-wide MFMA steps in the chunk
    const int per = LONG ? SPW : (ns + 3) >> 2;
    const int s_lo = wave * per, s_hi = min(ns, s_lo + per);
    float bv[LONG ? SPW : 16];
    auto fetch = [&](int u_lo, int cnt, int s0) {      // bv[u_lo + i] <- fragment of step s0 + i
#pragma unroll
      for (int i = 0; i < cnt; ++i) {
        const int kk = min(k0 + 2 * (s0 + i) + lhi, Kd - 1);
        bv[u_lo + i] = X[static_cast<int64_t>(kk) * D + dcol];
      }
    };
    if constexpr (LONG) fetch(0, SPW / 2, s_lo);
    else fetch(0, 8, s_lo);
    // ---- build the dS strip: Ds[kk][r] = dS(row r0 + r, k0 + kk), 32 S values per thread in flight ----
    float sv[32];
    if (dir == 0) {
      const int k = min(k0 + tid, Kd - 1);            // this thread's k for all 32 rows
#pragma unroll
      for (int r = 0; r < 32; ++r) sv[r] = S[static_cast<int64_t>(min(r0 + r, R - 1)) * ldS + k];
      const float kcv = kc[k], klv = kl[k];
      __syncthreads();                                 // rowc_s / rowl_s visible (and Ds free: see loop end)
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const float rcv = rowc_s[r];
        float d = rcv * fast_exp(sv[r] - rowl_s[r]) + kcv * fast_exp(sv[r] - klv);
        if (static_cast<int64_t>(k0 + tid) == r0 + r + doff) d -= (rcv + kcv);
        Ds[tid * RED_STRIDE + r] = (tid < kn && r0 + r < R) ? d : 0.f;
      }
    } else {
      const int r = tid & 31, kq = tid >> 5;
      const int row = min(r0 + r, R - 1);
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int k = min(k0 + kq + 8 * q, Kd - 1);
        sv[q] = S[static_cast<int64_t>(k) * ldS + row];
      }
      // the chunk's per-k coefficients go through LDS (one value per thread) instead of 2 x 32 registers per thread: the
      // kernel drops from 189 to ~128 VGPRs = 4 instead of 2 resident workgroups per CU
      kc_s[tid] = kc[min(k0 + tid, Kd - 1)];
      kl_s[tid] = kl[min(k0 + tid, Kd - 1)];
      __syncthreads();
      const float rcv = rowc_s[r], rlv = rowl_s[r];
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int kk = kq + 8 * q;
        const float kcv = kc_s[kk], klv = kl_s[kk];
        float d = rcv * fast_exp(sv[q] - rlv) + kcv * fast_exp(sv[q] - klv);
        if (static_cast<int64_t>(k0 + kk) == r0 + r + doff) d -= (rcv + kcv);
        Ds[kk * RED_STRIDE + r] = (kk < kn && r0 + r < R) ? d : 0.f;
      }
    }
    if constexpr (LONG) fetch(SPW / 2, SPW / 2, s_lo + SPW / 2);   // in flight across the barrier and the first 16 MFMAs
    __syncthreads();
    // ---- MFMA over this wave's quarter of the chunk ----
    if constexpr (LONG) {
#pragma unroll
      for (int u = 0; u < SPW; ++u)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ds[(2 * (s_lo + u) + lhi) * RED_STRIDE + l31], bv[u], acc, 0, 0, 0);
    } else {
      for (int s0 = s_lo; s0 < s_hi; s0 += 8) {       // 8 steps per round, the next round prefetched into bv[8..15]
        if (s0 + 8 < s_hi) fetch(8, 8, s0 + 8);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (s0 + u < s_hi) {  // wave-uniform
            const int kk = min(2 * (s0 + u) + lhi, GK - 1);   // entries at kk >= kn are 0
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ds[kk * RED_STRIDE + l31], bv[u], acc, 0, 0, 0);
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) bv[u] = bv[8 + u];
      }
    }
    // the barrier after the next chunk's loads (or the one below) orders these Ds reads before its rewrite
  }
  __syncthreads();
  stash_acc(red, acc, wave, l31, lhi);
  __syncthreads();
  if (nsl > 1 && tickets) {
    // ---- round 4: the slices of a contraction are summed IN the launch (it was small_slice_sum_kernel, a 5 us launch at
    // 150 x 1200): every slice publishes its 32 x 32 partial (tile-major private slab, write-through stores), the LAST
    // slice of an output tile to arrive adds them in slice order 0 .. nsl-1 - the order the separate kernel used, so the
    // same bits whoever is last.  Hand-off: common.hpp.  One direction per launch on this path (dir == dir0). ----
    const int tile = blockIdx.x * gridDim.y + blockIdx.y, tiles = gridDim.x * gridDim.y;
    float* sl = dir ? slabB : slabA;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q;
      v[q] = __fmul_rn(alpha, red_sum(red, e >> 5, e & 31));
      st_agent(sl + (static_cast<int64_t>(slice) * tiles + tile) * 1024 + e, v[q]);
    }
    __syncthreads();                                  // red is read out: its last words carry the ticket broadcast
    unsigned* slot = reinterpret_cast<unsigned*>(red + 3 * 32 * RED_STRIDE);
    if (draw_ticket(tickets + tile, slot) != static_cast<unsigned>(nsl - 1)) return;
    if (tid == 0) __hip_atomic_store(tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int z0 = 0; z0 < nsl; z0 += 8) {             // 8 slices' loads in flight at a time, added in slice order
      float w[4][8];
#pragma unroll
      for (int zz = 0; zz < 8; ++zz) {
        const float* src = sl + (static_cast<int64_t>(min(z0 + zz, nsl - 1)) * tiles + tile) * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q][zz] = ld_agent(src + tid + 256 * q);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int zz = 0; zz < 8; ++zz) {
          if (z0 + zz == 0) a[q] = w[q][0];           // first term as is (0 + x would turn -0 into +0)
          else if (z0 + zz < nsl) a[q] += w[q][zz];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q, row = e >> 5, col = e & 31;
      if (r0 + row < R && d0 + col < D) out[static_cast<int64_t>(r0 + row) * D + d0 + col] = a[q];
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + 256 * q, row = e >> 5, col = e & 31;
    if (r0 + row < R && d0 + col < D)
      out[static_cast<int64_t>(r0 + row) * D + d0 + col] = __fmul_rn(alpha, red_sum(red, row, col));
  }
}

// ---- loss assembly for the RAG-e2e step in one launch (replaces contrastive_finalize + ce_finalize + add) ----
//   out[1] = L_con = 0.5 (sum_i (lse_r[i]-diag[i]) + sum_j (lse_c[j]-diag[j])) / n_global
//   doc_lp[i] = diag[i] - lse_r[i]
//   out[2] = L_gen = (sum_r row_nll[r] - sum_b Nb[b] doc_lp[b]) / M ;  out[0] = L_con + L_gen
__global__ __launch_bounds__(1024) void rag_loss_finalize_kernel(const float* __restrict__ row_nll, int64_t R,
                                                                 const float* __restrict__ Nb,
                                                                 const float* __restrict__ lse_r,
                                                                 const float* __restrict__ lse_c,
                                                                 const float* __restrict__ diag, int n_local,
                                                                 float n_global, const float* __restrict__ stats,
                                                                 float* __restrict__ out,
                                                                 float* __restrict__ doc_lp) {
  __shared__ float red[16];
  float g = 0.f, c = 0.f;
  for (int64_t i = threadIdx.x; i < R; i += 1024) g += row_nll[i];
  for (int i = threadIdx.x; i < n_local; i += 1024) {
    const float d = diag[i], lp = d - lse_r[i];
    c += (lse_r[i] - d) + (lse_c[i] - d);
    g -= Nb[i] * lp;
    if (doc_lp) doc_lp[i] = lp;
  }
  g = block_sum<1024>(g, red);
  c = block_sum<1024>(c, red);
  if (threadIdx.x == 0) {
    const float con = 0.5f * c / n_global, gen = g / stats[0];
    out[0] = con + gen; out[1] = con; out[2] = gen;
  }
}

inline int64_t round_up(int64_t x, int64_t q) { return (x + q - 1) / q * q; }

constexpr int kDefaultPipeDepth = 4;     // measured: profiles/history/r04_small_pipe.txt (512^2 -8 %, 150 x 1200 -4 %; depth 2-4 alike)
struct SmallPlan { int sk, k_chunk; int64_t ldn, ldm; };
inline SmallPlan small_plan(int64_t m, int64_t n, int64_t D) {
  const int64_t tiles = ((m + 31) / 32) * ((n + 31) / 32);
  static const int64_t target = getenv("DALM_SMALL_TARGET") ? atoi(getenv("DALM_SMALL_TARGET")) : 256;   // workgroups aimed at
  int64_t sk = (target + tiles - 1) / tiles;
  const int64_t sk_max = (D + 31) / 32;        // every workgroup gets at least 32 of K (8 per wave)
  if (sk > SK_MAX) sk = SK_MAX;
  if (sk > sk_max) sk = sk_max;
  if (sk < 1) sk = 1;
  const int64_t k_chunk = round_up((D + sk - 1) / sk, 32);
  sk = (D + k_chunk - 1) / k_chunk;
  return {static_cast<int>(sk), static_cast<int>(k_chunk), round_up(n, 4), round_up(m, 4)};
}
inline bool vec16(const float* p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) % 16 == 0) && (ld % 4 == 0); }

}  // namespace
}  // namespace dalm

using namespace dalm;

// The small path applies when S (m x n floats) stays L2-sized and the grid of 32x32 tiles is modest.
extern "C" int dalm_sim_small_supported(int64_t m, int64_t n, int64_t D) {
  if (m <= 0 || n <= 0 || D <= 0) return 0;
  if (m > 1024 || n > 8192 || m * n > (1ll << 20) || D > (1 << 20)) return 0;
  return 1;
}

extern "C" size_t dalm_sim_small_workspace_bytes(int64_t m, int64_t n, int64_t D, int want_cols) {
  if (!dalm_sim_small_supported(m, n, D)) return 0;
  const SmallPlan pl = small_plan(m, n, D);
  size_t b = static_cast<size_t>(pl.sk) * m * pl.ldn;
  if (want_cols) b += static_cast<size_t>(pl.sk) * n * pl.ldm;
  return b * sizeof(float);
}

extern "C" int dalm_sim_small_fwd(const float* A, const float* Bm, int64_t m, int64_t n, int64_t D, float scale,
                                  int64_t diag_offset, float* S, int64_t ldS, float* row_lse, float* diag,
                                  float* col_lse, void* ws, size_t ws_bytes, dalm_stream_t stream) {
  DALM_REQUIRE(A && Bm && S && row_lse && diag && ws, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dalm_sim_small_supported(m, n, D), DALM_E_SHAPE, "shape outside the small-batch path (see dalm_sim_small_supported)");
  DALM_REQUIRE(ldS >= n, DALM_E_SHAPE, "ldS must be >= n");
  DALM_REQUIRE(diag_offset >= 0 && diag_offset + m <= n, DALM_E_SHAPE, "diag_offset + m must be <= n");
  const int want_cols = col_lse != nullptr;
  DALM_REQUIRE(ws_bytes >= dalm_sim_small_workspace_bytes(m, n, D, want_cols), DALM_E_WORKSPACE, "workspace too small");
  DALM_REQUIRE(reinterpret_cast<uintptr_t>(ws) % 4 == 0, DALM_E_ALIGN, "workspace must be 4-byte aligned");
  hipStream_t s = as_stream(stream);
  const SmallPlan pl = small_plan(m, n, D);
  float* slab = static_cast<float*>(ws);
  float* slabT = want_cols ? slab + static_cast<size_t>(pl.sk) * m * pl.ldn : nullptr;
  const int tiles_m = static_cast<int>((m + 31) / 32), tiles_n = static_cast<int>((n + 31) / 32);
  const dim3 pgrid(static_cast<unsigned>(tiles_m * tiles_n), static_cast<unsigned>(pl.sk));
  // pipelined form: the split covers K exactly and a wave's share is 4 or 8 whole rounds of 32 (D = 1024 with 1 or 2
  // slices: 512^2 ... 1024^2 on one GPU, the 150 x 1200 per-rank blocks); DALM_SMALL_PIPE = 0 (off) | 2 | 3 | 4 = rounds in flight
  static const int pipe_depth = getenv("DALM_SMALL_PIPE") ? atoi(getenv("DALM_SMALL_PIPE")) : kDefaultPipeDepth;
  const int rounds = (pl.k_chunk % 128 == 0 && static_cast<int64_t>(pl.k_chunk) * pl.sk == D) ? pl.k_chunk / 128 : 0;
  const bool fast = vec16(A, D) && vec16(Bm, D) && D % 8 == 0;
  const FusedFwd none{};
#define DALM_PIPE(R, DP) hipLaunchKernelGGL((small_partial_pipe_kernel<R, DP>), pgrid, dim3(256), 0, s, A, Bm, static_cast<int>(m), \
    static_cast<int>(n), static_cast<int>(D), pl.k_chunk, tiles_n, slab, static_cast<int>(pl.ldn), slabT, static_cast<int>(pl.ldm), none)
  if (fast && pipe_depth >= 2 && (rounds == 4 || rounds == 8)) {
    if (rounds == 4) { if (pipe_depth == 2) DALM_PIPE(4, 2); else if (pipe_depth == 3) DALM_PIPE(4, 3); else DALM_PIPE(4, 4); }
    else { if (pipe_depth == 2) DALM_PIPE(8, 2); else if (pipe_depth == 3) DALM_PIPE(8, 3); else DALM_PIPE(8, 4); }
  } else if (fast)
#undef DALM_PIPE
    hipLaunchKernelGGL(small_partial_kernel<true>, pgrid, dim3(256), 0, s, A, Bm, static_cast<int>(m),
                       static_cast<int>(n), static_cast<int>(D), pl.k_chunk, tiles_n, 1, 1, slab,
                       static_cast<int>(pl.ldn), slabT, static_cast<int>(pl.ldm), none);
  else
    hipLaunchKernelGGL(small_partial_kernel<false>, pgrid, dim3(256), 0, s, A, Bm, static_cast<int>(m),
                       static_cast<int>(n), static_cast<int>(D), pl.k_chunk, tiles_n, static_cast<int>(vec16(A, D)),
                       static_cast<int>(vec16(Bm, D)), slab, static_cast<int>(pl.ldn), slabT, static_cast<int>(pl.ldm), none);
  const int64_t rmax = want_cols ? (m > n ? m : n) : m;
  const int64_t cmax = want_cols ? (m > n ? m : n) : n;                  // longest row the statistics pass reduces
  if (cmax > 256)
    hipLaunchKernelGGL(small_stats_wide_kernel, dim3(static_cast<unsigned>(rmax), want_cols ? 2u : 1u), dim3(256), 0,
                       s, slab, static_cast<int>(pl.ldn), slabT, static_cast<int>(pl.ldm), pl.sk, static_cast<int>(m),
                       static_cast<int>(n), scale, diag_offset, S, ldS, row_lse, diag, col_lse);
  else
    hipLaunchKernelGGL(small_stats_kernel, dim3(static_cast<unsigned>((rmax + 3) / 4), want_cols ? 2u : 1u), dim3(256), 0,
                       s, slab, static_cast<int>(pl.ldn), slabT, static_cast<int>(pl.ldm), pl.sk, static_cast<int>(m),
                       static_cast<int>(n), scale, diag_offset, S, ldS, row_lse, diag, col_lse);
  return check_launch(__func__);
}

// ---- one-launch forward --------------------------------------------------------------------------------------
namespace {
struct Fused1Layout { size_t tslab, rowpart, colpart, total; int tiles_m, tiles_n; };
inline Fused1Layout fused1_layout(int64_t m, int64_t n, int64_t D, int want_cols) {
  const SmallPlan pl = small_plan(m, n, D);
  Fused1Layout L{};
  L.tiles_m = static_cast<int>((m + 31) / 32); L.tiles_n = static_cast<int>((n + 31) / 32);
  const size_t tiles = static_cast<size_t>(L.tiles_m) * L.tiles_n;
  size_t o = 256;                                           // slack for aligning the caller's pointer up to 256 bytes
  L.tslab = o; o += (pl.sk > 1 ? static_cast<size_t>(pl.sk) * tiles * 4096 : 0);
  L.rowpart = o; o += static_cast<size_t>(L.tiles_n) * L.tiles_m * 32 * 8;
  L.colpart = o; o += want_cols ? static_cast<size_t>(L.tiles_m) * L.tiles_n * 32 * 8 : 0;
  L.total = o;
  return L;
}
}  // namespace

extern "C" size_t dalm_sim_small_fwd1_workspace_bytes(int64_t m, int64_t n, int64_t D, int want_cols) {
  if (!dalm_sim_small_supported(m, n, D)) return 0;
  return fused1_layout(m, n, D, want_cols).total;
}

// Whether the one-launch forward is the faster form for a shape (profiles/history/r04_small_one_launch.txt): a tile grid of >= 128
// tiles with at most 2 K slices per tile - the hand-offs then replace a statistics kernel that had real work to do
// (512^2: 19.7 -> 18.6 us, 150 x 1200: 16.4 -> 15.7 us); with few tiles and 16 slices the last slice's serial sum loses
// (18^2: 6.3 -> 8.7 us).  Callers that follow this advice keep BOTH forms available: dalm_sim_small_fwd needs no tickets.
extern "C" int dalm_sim_small_fwd1_preferred(int64_t m, int64_t n, int64_t D) {
  if (!dalm_sim_small_supported(m, n, D)) return 0;
  const SmallPlan pl = small_plan(m, n, D);
  const int64_t tiles = ((m + 31) / 32) * ((n + 31) / 32);
  return (tiles >= 128 && pl.sk <= 2) ? 1 : 0;
}

extern "C" size_t dalm_sim_small_fwd1_ticket_words(int64_t m, int64_t n) {
  if (m <= 0 || n <= 0) return 0;
  const size_t tm = static_cast<size_t>((m + 31) / 32), tn = static_cast<size_t>((n + 31) / 32);
  return tm * tn + tm + tn;
}

extern "C" int dalm_sim_small_fwd1(const float* A, const float* Bm, int64_t m, int64_t n, int64_t D, float scale,
                                   int64_t diag_offset, float* S, int64_t ldS, float* row_lse, float* diag,
                                   float* col_lse, void* ws, size_t ws_bytes, unsigned* tickets, dalm_stream_t stream) {
  DALM_REQUIRE(A && Bm && S && row_lse && diag && ws && tickets, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dalm_sim_small_supported(m, n, D), DALM_E_SHAPE, "shape outside the small-batch path (see dalm_sim_small_supported)");
  DALM_REQUIRE(ldS >= n, DALM_E_SHAPE, "ldS must be >= n");
  DALM_REQUIRE(diag_offset >= 0 && diag_offset + m <= n, DALM_E_SHAPE, "diag_offset + m must be <= n");
  const int want_cols = col_lse != nullptr;
  const Fused1Layout L = fused1_layout(m, n, D, want_cols);
  DALM_REQUIRE(ws_bytes >= L.total, DALM_E_WORKSPACE, "workspace too small");
  DALM_REQUIRE(reinterpret_cast<uintptr_t>(tickets) % 4 == 0, DALM_E_ALIGN, "tickets must be 4-byte aligned");
  hipStream_t s = as_stream(stream);
  const SmallPlan pl = small_plan(m, n, D);
  unsigned char* base = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(ws) + 255) / 256 * 256) - 256;
  FusedFwd fz{};
  fz.S = S; fz.ldS = ldS; fz.row_lse = row_lse; fz.diag = diag; fz.col_lse = col_lse;
  fz.alpha = scale; fz.diag_offset = diag_offset;
  fz.tslab = reinterpret_cast<float*>(base + L.tslab);
  fz.rowpart = reinterpret_cast<unsigned long long*>(base + L.rowpart);
  fz.colpart = want_cols ? reinterpret_cast<unsigned long long*>(base + L.colpart) : nullptr;
  fz.tickets = tickets; fz.sk = pl.sk; fz.tiles_m = L.tiles_m; fz.tiles_n = L.tiles_n;
  const dim3 pgrid(static_cast<unsigned>(L.tiles_m * L.tiles_n), static_cast<unsigned>(pl.sk));
  const int rounds = (pl.k_chunk % 128 == 0 && static_cast<int64_t>(pl.k_chunk) * pl.sk == D) ? pl.k_chunk / 128 : 0;
  const bool fast = vec16(A, D) && vec16(Bm, D) && D % 8 == 0;
  float* nof = nullptr;
#define DALM_PIPE1(R) hipLaunchKernelGGL((small_partial_pipe_kernel<R, 4, true>), pgrid, dim3(256), 0, s, A, Bm, static_cast<int>(m), \
    static_cast<int>(n), static_cast<int>(D), pl.k_chunk, L.tiles_n, nof, 0, nof, 0, fz)
  if (fast && rounds == 4) DALM_PIPE1(4);
  else if (fast && rounds == 8) DALM_PIPE1(8);
#undef DALM_PIPE1
  else if (fast)
    hipLaunchKernelGGL((small_partial_kernel<true, true>), pgrid, dim3(256), 0, s, A, Bm, static_cast<int>(m),
                       static_cast<int>(n), static_cast<int>(D), pl.k_chunk, L.tiles_n, 1, 1, nof, 0, nof, 0, fz);
  else
    hipLaunchKernelGGL((small_partial_kernel<false, true>), pgrid, dim3(256), 0, s, A, Bm, static_cast<int>(m),
                       static_cast<int>(n), static_cast<int>(D), pl.k_chunk, L.tiles_n, static_cast<int>(vec16(A, D)),
                       static_cast<int>(vec16(Bm, D)), nof, 0, nof, 0, fz);
  return check_launch(__func__);
}

namespace {
// contraction slices of the backward: as many as keep the grid at <= 1024 workgroups, at most one per 256-chunk; only
// when ONE direction is asked for (the sharded form; with both directions the grid is already two problems wide)
inline int small_bwd_slices(int64_t m, int64_t n, int64_t D, bool want_dA, bool want_dB) {
  if (want_dA == want_dB) return 1;
  const int64_t R = want_dA ? m : n, Kd = want_dA ? n : m;
  const int64_t blocks = ((R + 31) / 32) * ((D + 31) / 32), chunks = (Kd + GK - 1) / GK;
  int64_t nsl = blocks > 0 ? 1024 / blocks : 1;
  if (nsl > chunks) nsl = chunks;
  return static_cast<int>(nsl < 1 ? 1 : nsl);
}
}  // namespace

extern "C" size_t dalm_sim_small_bwd_workspace_bytes(int64_t m, int64_t n, int64_t D, int want_dA, int want_dB) {
  if (!dalm_sim_small_supported(m, n, D)) return 0;
  const int nsl = small_bwd_slices(m, n, D, want_dA != 0, want_dB != 0);
  if (nsl <= 1) return 0;
  return static_cast<size_t>(nsl) * static_cast<size_t>(want_dA ? m : n) * static_cast<size_t>(D) * sizeof(float);
}

// One-launch form of the sliced backward (round 4): the slices of a contraction are summed by the last slice of every output
// tile to arrive instead of by small_slice_sum_kernel.  `tickets`: dalm_sim_small_bwd1_ticket_words words, ZERO on entry, left
// zero (the forward's ticket buffer can be shared: calls are stream-ordered).  Workspace: tile-major partial tiles.
extern "C" size_t dalm_sim_small_bwd1_workspace_bytes(int64_t m, int64_t n, int64_t D, int want_dA, int want_dB) {
  if (!dalm_sim_small_supported(m, n, D)) return 0;
  const int nsl = small_bwd_slices(m, n, D, want_dA != 0, want_dB != 0);
  if (nsl <= 1) return 0;
  const int64_t R = want_dA ? m : n;
  return static_cast<size_t>(nsl) * static_cast<size_t>((R + 31) / 32) * static_cast<size_t>((D + 31) / 32) * 4096;
}

extern "C" size_t dalm_sim_small_bwd1_ticket_words(int64_t m, int64_t n, int64_t D) {
  if (m <= 0 || n <= 0 || D <= 0) return 0;
  const int64_t R = m > n ? m : n;
  return static_cast<size_t>((R + 31) / 32) * static_cast<size_t>((D + 31) / 32);
}

static int small_bwd_impl(const float* S, int64_t ldS, const float* A, const float* Bm, int64_t m, int64_t n, int64_t D,
                          float scale, int64_t diag_offset, const float* row_coef, const float* row_lse, const float* col_coef,
                          const float* col_lse, float* dA, float* dB, void* ws, size_t ws_bytes, unsigned* tickets,
                          dalm_stream_t stream, const char* fn);

extern "C" int dalm_sim_small_bwd1(const float* S, int64_t ldS, const float* A, const float* Bm, int64_t m, int64_t n,
                                   int64_t D, float scale, int64_t diag_offset, const float* row_coef,
                                   const float* row_lse, const float* col_coef, const float* col_lse, float* dA,
                                   float* dB, void* ws, size_t ws_bytes, unsigned* tickets, dalm_stream_t stream) {
  DALM_REQUIRE(tickets, DALM_E_NULL, "tickets are required (dalm_sim_small_bwd_ws is the two-launch form)");
  return small_bwd_impl(S, ldS, A, Bm, m, n, D, scale, diag_offset, row_coef, row_lse, col_coef, col_lse, dA, dB, ws, ws_bytes,
                        tickets, stream, __func__);
}

extern "C" int dalm_sim_small_bwd_ws(const float* S, int64_t ldS, const float* A, const float* Bm, int64_t m, int64_t n,
                                     int64_t D, float scale, int64_t diag_offset, const float* row_coef,
                                     const float* row_lse, const float* col_coef, const float* col_lse, float* dA,
                                     float* dB, void* ws, size_t ws_bytes, dalm_stream_t stream) {
  return small_bwd_impl(S, ldS, A, Bm, m, n, D, scale, diag_offset, row_coef, row_lse, col_coef, col_lse, dA, dB, ws, ws_bytes,
                        nullptr, stream, __func__);
}

static int small_bwd_impl(const float* S, int64_t ldS, const float* A, const float* Bm, int64_t m, int64_t n, int64_t D,
                          float scale, int64_t diag_offset, const float* row_coef, const float* row_lse, const float* col_coef,
                          const float* col_lse, float* dA, float* dB, void* ws, size_t ws_bytes, unsigned* tickets,
                          dalm_stream_t stream, const char* fn) {
  if (!(S && A && Bm && row_coef && row_lse && col_coef && col_lse)) return fail(DALM_E_NULL, fn, "null pointer argument");
  if (!(dA || dB)) return fail(DALM_E_NULL, fn, "at least one of dA / dB is required");
  if (!dalm_sim_small_supported(m, n, D)) return fail(DALM_E_SHAPE, fn, "shape outside the small-batch path");
  if (ldS < n) return fail(DALM_E_SHAPE, fn, "ldS must be >= n");
  const int dir0 = dA ? 0 : 1, ndir = (dA && dB) ? 2 : 1;
  int nsl = small_bwd_slices(m, n, D, dA != nullptr, dB != nullptr);
  const size_t need = tickets ? dalm_sim_small_bwd1_workspace_bytes(m, n, D, dA != nullptr, dB != nullptr)
                              : dalm_sim_small_bwd_workspace_bytes(m, n, D, dA != nullptr, dB != nullptr);
  if (nsl > 1 && (!ws || ws_bytes < need || reinterpret_cast<uintptr_t>(ws) % 16 != 0)) nsl = 1;   // no workspace: unsliced form
  if (nsl <= 1) tickets = nullptr;
  const int64_t rmax = (ndir == 2) ? (m > n ? m : n) : (dir0 ? n : m);
  const dim3 grid(static_cast<unsigned>((rmax + 31) / 32), static_cast<unsigned>((D + 31) / 32),
                  static_cast<unsigned>(ndir * nsl));
  float* slab = static_cast<float*>(ws);
  hipStream_t s = as_stream(stream);
  // the shortest contraction among the launched directions decides the form (dir 0 contracts over n, dir 1 over m)
  const int64_t kd_min = (ndir == 2) ? (m < n ? m : n) : (dir0 ? m : n);
  if (kd_min > 64)
    hipLaunchKernelGGL(small_grad_kernel<true>, grid, dim3(256), 0, s, S, ldS, A, Bm, static_cast<int>(m),
                       static_cast<int>(n), static_cast<int>(D), scale, diag_offset, row_coef, row_lse, col_coef, col_lse,
                       dA, dB, dir0, nsl, dA ? slab : nullptr, dA ? nullptr : slab, tickets);
  else
    hipLaunchKernelGGL(small_grad_kernel<false>, grid, dim3(256), 0, s, S, ldS, A, Bm, static_cast<int>(m),
                       static_cast<int>(n), static_cast<int>(D), scale, diag_offset, row_coef, row_lse, col_coef, col_lse,
                       dA, dB, dir0, nsl, dA ? slab : nullptr, dA ? nullptr : slab, tickets);
  if (nsl > 1 && !tickets) {
    const int64_t count = (dA ? m : n) * D;
    int64_t blocks = (count / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(small_slice_sum_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, slab, nsl, count,
                       dA ? dA : dB);
  }
  return check_launch(fn);
}

extern "C" int dalm_sim_small_bwd(const float* S, int64_t ldS, const float* A, const float* Bm, int64_t m, int64_t n,
                                  int64_t D, float scale, int64_t diag_offset, const float* row_coef,
                                  const float* row_lse, const float* col_coef, const float* col_lse, float* dA,
                                  float* dB, dalm_stream_t stream) {
  return dalm_sim_small_bwd_ws(S, ldS, A, Bm, m, n, D, scale, diag_offset, row_coef, row_lse, col_coef, col_lse, dA, dB,
                               nullptr, 0, stream);
}

extern "C" int dalm_rag_loss_finalize(const float* row_nll, int64_t num_rows, const float* Nb, const float* row_lse,
                                      const float* col_lse, const float* diag, int64_t n_local, int64_t n_global,
                                      const float* stats, float* out, float* doc_lp, dalm_stream_t stream) {
  DALM_REQUIRE(row_nll && Nb && row_lse && col_lse && diag && stats && out, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(num_rows > 0 && n_local > 0 && n_global >= n_local && n_local <= 0x7fffffffll, DALM_E_SHAPE,
               "need num_rows>0, 0 < n_local <= n_global");
  hipLaunchKernelGGL(rag_loss_finalize_kernel, dim3(1), dim3(1024), 0, as_stream(stream), row_nll, num_rows, Nb,
                     row_lse, col_lse, diag, static_cast<int>(n_local), static_cast<float>(n_global), stats, out, doc_lp);
  return check_launch(__func__);
}
