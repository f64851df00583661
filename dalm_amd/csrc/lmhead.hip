// lm_head + log-sum-exp + label gather in ONE bf16 MFMA kernel: forward value of the generator loss with the
// [rows, V] logits never written anywhere (evaluation path of SURVEY.md section 8 f1).
//
// Stands in for   logits = lm_head(hidden)            dalm/models/rag_e2e_base_model.py:104-106
//                 logsumexp / gather of the label     dalm/training/utils/train_utils.py:113-131
// for rows that carry loss.  Each workgroup owns a 128 x 128 tile of (row, vocabulary) pairs, contracts it over the
// hidden width on v_mfma_f32_32x32x16_bf16 and reduces the tile IN REGISTERS to a per-row (max, sum exp) pair per 64
// columns plus the label's logit; lm_head_lse_merge_kernel folds the 2*V/128 partials of a row.
// The backward is NOT fused this way (it needs the finished log-sum-exp before any softmax * W product exists, i.e. a
// third GEMM - see DESIGN.md section 9 f1): training keeps the chunked hipBLASLt path of dalm_amd/fused.py.
//
// Data flow per K step of 64: 16-byte global loads (full 128-byte lines per row) -> registers -> ds_write_b128 into
// rows padded to 144 bytes (conflict-free 16-byte fragment reads) -> ds_read_b128 fragments -> MFMA.  Both operands are
// K-contiguous ("B^T input"), so A and B fragments use the same addressing; the loads of step i+1 are in flight while
// step i is multiplied.
#include "common.hpp"

namespace dalm {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int LBN = 128;

struct LmParams {
  const unsigned short* H;   // [R, K] bf16
  const unsigned short* W;   // [V, K] bf16
  const int64_t* labels;     // [R]; < 0: row carries no loss
  int R, V, K, MT, NT, xcd_order;
  float* pm;                 // [2*NT, R] partial maxima
  float* pl;                 // [2*NT, R] partial sums of exp(x - max)
  float* z;                  // [R] logit of the label
};

// TM = 32-row MFMA tiles per wave along the rows: the workgroup tile is (64 TM) x 128, waves as 2 x 2, each 32 TM x 64.
// TM = 4 halves the LDS fragment bytes per MFMA of TM = 2 (0.75 instead of 1 fragment per MFMA).
// Measured at 3584 x 32000 x 4096 (MI355X): TM = 2: 838 TF/s, TM = 4: 881 TF/s (936 at V = 65024).  Tried and slower: a second
// LDS stage with one barrier per step (786 / 635 TF/s for TM = 2 / 4: the occupancy it costs was already doing the overlapping),
// K steps of 128 (783 TF/s).  This is the two-barrier LDS structure's ceiling (guide, section 5: ~900 TF/s); hipBLASLt's
// hand-scheduled kernel plus the forward CE kernel runs the same problem at 1285 TF/s.
template <int TM, int LBK>
__global__ __launch_bounds__(256, 2) void lm_head_lse_kernel(const LmParams p) {
  constexpr int LBM = 64 * TM;
  constexpr int LROW = LBK * 2 + 16;    // bytes per LDS row: LBK bf16 + 16 bytes of padding (conflict-free 16-byte reads)
  constexpr int CH = LBK / 8, RP = 256 / CH;   // 16-byte chunks per row; rows covered by one pass of the 256 threads
  constexpr int ATILE = LBM * LROW;     // A tile in LDS; the B tile (128 rows) follows it
  __shared__ __attribute__((aligned(16))) unsigned char lds[ATILE + LBN * LROW];
  __shared__ int lab_s[LBM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware order: workgroup L runs on XCD L % 8 (own 4 MB L2).  Every XCD gets a contiguous run of tile ids = a range
  // of vocabulary tiles with all their row tiles, so a 1 MB W tile is pulled into ONE L2 and shared by the row tiles that
  // use it (plain order: each W tile lands in up to 8 L2s and is shared by < 2 workgroups per XCD).  DALM_LM_HEAD_XCD=0 disables.
  unsigned wgid = blockIdx.x;
  if (p.xcd_order) {
    const unsigned nwg = gridDim.x, L = blockIdx.x;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7u, xcd = L & 7u;
    wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (L >> 3);
  }
  const int mt = static_cast<int>(wgid) % p.MT, nt = static_cast<int>(wgid) / p.MT;
  const int r0 = mt * LBM, c0 = nt * LBN;

  for (int t = tid; t < LBM; t += 256) {
    const int r = r0 + t;
    int y = -1;
    if (r < p.R) {
      const int64_t yl = p.labels[r];
      y = (yl < 0) ? -1 : (yl >= p.V ? -2 : static_cast<int>(yl));
    }
    lab_s[t] = y;
  }

  // staging geometry: thread t moves the 16-byte chunk (t % CH) of rows (t / CH) + RP j of both tiles
  const int srow = tid / CH, sch = tid % CH;
  constexpr int NA = LBM / RP, NB = LBN / RP;   // 16-byte loads per thread and K step
  const unsigned short* ga[NA];
  const unsigned short* gb[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int ra = min(r0 + srow + RP * j, p.R - 1);          // clamped: unconditional loads, results masked later
    ga[j] = p.H + static_cast<int64_t>(ra) * p.K + sch * 8;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int rb = min(c0 + srow + RP * j, p.V - 1);
    gb[j] = p.W + static_cast<int64_t>(rb) * p.K + sch * 8;
  }
  unsigned char* sa = lds + srow * LROW + sch * 16;
  unsigned char* sb = lds + ATILE + srow * LROW + sch * 16;
  const unsigned char* fa = lds + (wm * 32 * TM + l31) * LROW + lhi * 16;
  const unsigned char* fb = lds + ATILE + (wn * 64 + l31) * LROW + lhi * 16;

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 ra4[NA], rb4[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) ra4[j] = *reinterpret_cast<const u32x4*>(ga[j]);
#pragma unroll
  for (int j = 0; j < NB; ++j) rb4[j] = *reinterpret_cast<const u32x4*>(gb[j]);
#pragma unroll
  for (int j = 0; j < NA; ++j) *reinterpret_cast<u32x4*>(sa + RP * j * LROW) = ra4[j];
#pragma unroll
  for (int j = 0; j < NB; ++j) *reinterpret_cast<u32x4*>(sb + RP * j * LROW) = rb4[j];
  __syncthreads();

  const int nk = p.K / LBK;
  for (int it = 0; it < nk; ++it) {
    const int kn = min(it + 1, nk - 1) * LBK;                  // last step re-reads its own tile: no branch around loads
#pragma unroll
    for (int j = 0; j < NA; ++j) ra4[j] = *reinterpret_cast<const u32x4*>(ga[j] + kn);
#pragma unroll
    for (int j = 0; j < NB; ++j) rb4[j] = *reinterpret_cast<const u32x4*>(gb[j] + kn);
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch AHEAD of the multiply (the scheduler otherwise sinks it to the barrier)
#pragma unroll
    for (int kk = 0; kk < LBK / 16; ++kk) {
      bf16x8 a[TM], b[2];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(fa + i * 32 * LROW + kk * 32);
#pragma unroll
      for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const bf16x8*>(fb + i * 32 * LROW + kk * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                                           // every wave has read this step's tiles
#pragma unroll
    for (int j = 0; j < NA; ++j) *reinterpret_cast<u32x4*>(sa + RP * j * LROW) = ra4[j];
#pragma unroll
    for (int j = 0; j < NB; ++j) *reinterpret_cast<u32x4*>(sb + RP * j * LROW) = rb4[j];
    __syncthreads();
  }

  // ---- epilogue: the tile never leaves the registers ----
  // C layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const int colA = c0 + wn * 64 + l31, colB = colA + 32;
  const bool okA = colA < p.V, okB = colB < p.V;
  const int64_t prow = static_cast<int64_t>(nt * 2 + wn) * p.R;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = wm * 32 * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const int row = r0 + rl;
      const float x0 = okA ? acc[i][0][r] : -INFINITY;
      const float x1 = okB ? acc[i][1][r] : -INFINITY;
      float m = fmaxf(x0, x1);
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));   // stays inside the 32-lane half
      const float mref = (m == -INFINITY) ? 0.f : m;           // a 64-column strip entirely beyond V: (max -inf, sum 0)
      float s = __builtin_amdgcn_exp2f((x0 - mref) * kLog2e) + __builtin_amdgcn_exp2f((x1 - mref) * kLog2e);
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (row < p.R) {
        if (l31 == 0) {
          p.pm[prow + row] = m;
          p.pl[prow + row] = s;
        }
        const int y = lab_s[rl];
        if (y == colA) p.z[row] = x0;
        else if (y == colB) p.z[row] = x1;
      }
    }
  }
}

// One workgroup = 64 rows; wave w folds the partials p = w, w+4, ... of its rows (lane = row), 8 loads in flight,
// then the four waves are combined through LDS in fixed order.
__global__ __launch_bounds__(256) void lm_head_lse_merge_kernel(const float* __restrict__ pm, const float* __restrict__ pl,
                                                                const float* __restrict__ z,
                                                                const int64_t* __restrict__ labels, int R, int V, int P,
                                                                float* __restrict__ row_lse, float* __restrict__ row_nll) {
  __shared__ float ms[4][64], ls[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 64 + lane;
  const int rr = min(row, R - 1);
  float m = -INFINITY, l = 0.f;
  for (int p0 = w; p0 < P; p0 += 32) {
    float vm[8], vl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int pp = min(p0 + 4 * u, P - 1);
      vm[u] = pm[static_cast<int64_t>(pp) * R + rr];
      vl[u] = pl[static_cast<int64_t>(pp) * R + rr];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (p0 + 4 * u < P && vm[u] != -INFINITY) {
        const float mn = fmaxf(m, vm[u]);
        l = l * __builtin_amdgcn_exp2f((m - mn) * kLog2e) + vl[u] * __builtin_amdgcn_exp2f((vm[u] - mn) * kLog2e);
        m = mn;
      }
    }
  }
  ms[w][lane] = m;
  ls[w][lane] = l;
  __syncthreads();
  if (w == 0 && row < R) {
    float M = ms[0][lane];
#pragma unroll
    for (int i = 1; i < 4; ++i) M = fmaxf(M, ms[i][lane]);
    float L = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) L += (ms[i][lane] == -INFINITY) ? 0.f : ls[i][lane] * __builtin_amdgcn_exp2f((ms[i][lane] - M) * kLog2e);
    const float lse = M + __logf(L);
    row_lse[row] = lse;
    const int64_t y = labels[row];
    row_nll[row] = (y < 0) ? 0.f : (y < V ? lse - z[row] : __builtin_nanf(""));
  }
}

}  // namespace
}  // namespace dalm

using namespace dalm;

extern "C" size_t dalm_lm_head_lse_workspace_bytes(int64_t R, int64_t V) {
  if (R <= 0 || V <= 0) return 0;
  const int64_t NT = (V + LBN - 1) / LBN;
  return static_cast<size_t>(2 * NT * 2 + 1) * static_cast<size_t>(R) * sizeof(float);
}

extern "C" int dalm_lm_head_lse_fwd(const void* hidden, const void* weight, const int64_t* labels, int64_t R,
                                    int64_t V, int64_t K, float* row_lse, float* row_nll, void* ws, size_t ws_bytes,
                                    dalm_stream_t stream) {
  DALM_REQUIRE(hidden && weight && labels && row_lse && row_nll && ws, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(R > 0 && V > 0 && K > 0 && R <= 0x7fffff00ll && V <= 0x7fffff00ll, DALM_E_SHAPE, "need R, V, K > 0");
  DALM_REQUIRE(K % 64 == 0, DALM_E_SHAPE, "the hidden width must be a multiple of 64");
  DALM_REQUIRE(reinterpret_cast<uintptr_t>(hidden) % 16 == 0 && reinterpret_cast<uintptr_t>(weight) % 16 == 0,
               DALM_E_ALIGN, "hidden / weight must be 16-byte aligned");
  DALM_REQUIRE(ws_bytes >= dalm_lm_head_lse_workspace_bytes(R, V), DALM_E_SHAPE, "workspace too small");
  // 256-row tiles (fewer LDS bytes per MFMA) once they still give every CU several tiles; 128-row tiles below that
  static const char* tm_env = getenv("DALM_LM_HEAD_TM");
  const int64_t NT = (V + LBN - 1) / LBN;
  int tm = (((R + 255) / 256) * NT >= 1024) ? 4 : 2;
  if (tm_env) tm = (atoi(tm_env) == 4) ? 4 : 2;
  const int64_t LBM = 64 * tm, MT = (R + LBM - 1) / LBM;
  DALM_REQUIRE(MT * NT <= 0x7fffffffll, DALM_E_SHAPE, "too many tiles for one launch");
  hipStream_t s = as_stream(stream);
  LmParams p;
  p.H = static_cast<const unsigned short*>(hidden);
  p.W = static_cast<const unsigned short*>(weight);
  p.labels = labels;
  p.R = static_cast<int>(R); p.V = static_cast<int>(V); p.K = static_cast<int>(K);
  p.MT = static_cast<int>(MT); p.NT = static_cast<int>(NT);
  static const char* xcd_env = getenv("DALM_LM_HEAD_XCD");
  p.xcd_order = xcd_env ? atoi(xcd_env) != 0 : 1;
  float* f = static_cast<float*>(ws);
  p.pm = f;
  p.pl = f + 2 * NT * R;
  p.z = f + 4 * NT * R;
  const dim3 grid(static_cast<unsigned>(MT * NT));
  if (tm == 4) hipLaunchKernelGGL((lm_head_lse_kernel<4, 64>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((lm_head_lse_kernel<2, 64>), grid, dim3(256), 0, s, p);
  hipLaunchKernelGGL(lm_head_lse_merge_kernel, dim3(static_cast<unsigned>((R + 63) / 64)), dim3(256), 0, s, p.pm, p.pl,
                     p.z, labels, p.R, p.V, static_cast<int>(2 * NT), row_lse, row_nll);
  return check_launch(__func__);
}
