// lm_head + log-sum-exp + label gather in ONE bf16 MFMA kernel: forward value of the generator loss with the
// [rows, V] logits never written anywhere (evaluation path of SURVEY.md section 8 f1).
//
// Stands in for   logits = lm_head(hidden)            dalm/models/rag_e2e_base_model.py:104-106
//                 logsumexp / gather of the label     dalm/training/utils/train_utils.py:113-131
// for rows that carry loss.  Each workgroup owns a 128 x 128 tile of (row, vocabulary) pairs, contracts it over the
// hidden width on v_mfma_f32_32x32x16_bf16 and reduces the tile IN REGISTERS to a per-row (max, sum exp) pair per 64
// columns plus the label's logit; lm_head_lse_merge_kernel folds the 2*V/128 partials of a row.
// Round 5: the backward on the same core (dalm_lm_head_dlogits / dalm_lm_head_dhidden below): with the finished row
// log-sum-exp the logits tile is RECOMPUTED per vocabulary chunk, turned into coef * (softmax - onehot) in registers, staged
// in bf16 in a chunk-sized workspace and contracted with the (transposed) chunk of W for d(hidden) - a logits-free training
// step that is hand-written end to end.
//
// Data flow per K step of 64: 16-byte global loads (full 128-byte lines per row) -> registers -> ds_write_b128 into
// rows padded to 144 bytes (conflict-free 16-byte fragment reads) -> ds_read_b128 fragments -> MFMA.  Both operands are
// K-contiguous ("B^T input"), so A and B fragments use the same addressing; the loads of step i+1 are in flight while
// step i is multiplied.
#include "common.hpp"
#include <type_traits>

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

// ------------------------------------------------------------------------------------------------------------------
// Round 3: the same contraction on a structure built from measurements on the MI355X (tools/lm_head_ablate.py,
// profiles/history/r03_lm_head_*): 256 x 256 x 64 tiles, FOUR waves = one per SIMD with the whole 512-register file each,
// operands straight from global memory into LDS (buffer_load_dwordx4 ... lds), ONE barrier per K tile, a VALU-only epilogue,
// one workgroup per tile.  What the intermediate forms showed:
//   * 8 waves in ping-pong with two barriers per 8 MFMAs (the published "8-phase" shape): even its MFMA-and-barriers-only
//     skeleton stops at 1.2 PF/s - an s_barrier hand-off costs ~150 cycles during which nothing multiplies;
//   * 32-lane __shfl_xor reductions in the epilogue = 640 ds_bpermute round trips per lane and tile: 30 % of the kernel;
//   * every direct-to-LDS piece costs the issuing wave ~25-50 cycles that no second wave per SIMD hides (8 free-running
//     waves measured SLOWER than 4), issuing a tile's 16 pieces as early as possible beats spreading them, and a four-stage
//     ring of 32-deep tiles (3 stages of prefetch) is no faster than two 64-deep buffers;
//   * L2 hit rate 50 % -> 78 % with the banded tile order below.
//
// Workgroup = 4 waves as 2 x 2; a wave owns 128 x 128 outputs = 16 accumulator tiles of 32 x 32 (256 AGPRs).  A K tile
// (64 deep) is walked in four k-steps of 16 MFMAs (512 matrix-pipe cycles); the 8 fragments of the NEXT k-step are read from
// LDS and the load pieces of a later K tile are issued in the shadow of the current k-step's MFMAs (explicit
// sched_group_barrier interleave: MFMA / ds_read / MFMA / ds_read ... MFMA / buffer_load ...):
//     k-step 0..2 of tile t : multiply | read k-step +1 of tile t            | 8 of the 16 pieces of tile t+1 (k-step 0)
//     s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier          -> tile t+1 has landed for everybody, tile t's buffer is free
//     k-step 3 of tile t    : multiply | read k-step 0 of tile t+1           | the first 8 pieces of tile t+2 (tile t's buffer)
// LDS image of a K tile: per operand [256 rows][128 bytes], 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7): the
// 16 lanes a ds_read_b128 services together (16 rows, one k-chunk) hit 16 different 16-byte slots of the 256-byte bank row.
// The LDS-DMA writes lane-linear, so the permutation is applied to the per-lane SOURCE offset (within one 128-byte line: still
// full-line requests) and again on the read.  Rows / vocabulary entries past the end read as zero through the buffer
// descriptor's range check (no clamping, no branches); the zero-filled vocabulary columns leave the reduction in the epilogue.
// One workgroup per output tile, dispatched in the banded order below.  (A persistent form - every workgroup walking its share
// of the tile list and requesting the next tile's first K tile ahead of the epilogue - measured 3-4 % SLOWER: static
// assignment loses the dispatcher's load balancing and the extra live state spilled.)
struct Lm8Params {
  const void* H;             // [R, K] bf16
  const void* W;             // [V, K] bf16
  const int64_t* labels;     // [R]
  int R, V, K, MT, NT, xcd_order;
  int gh;                    // row tiles per band of the tile order (see lm_tile_of)
  unsigned bytesH, bytesW;
  unsigned pitchH, pitchW;   // bytes between rows of H / W (K * 2 for the lm_head; 3 D * 2 for the bf16x3 similarity operands)
  int tpd;                   // SPLIT3: K tiles (64 deep) per segment = D / 64
  unsigned tpd_inv;          // SPLIT3: ceil(2^24 / tpd): u / tpd == (u * tpd_inv) >> 24 for u < 6 tpd, tpd <= 1024 (checked exhaustively up to 1686)
  unsigned seg_bytes;        // SPLIT3: D * 2
  float* gmax;               // MODE_GMAX: [R][ng] one maximum per (row, 32 consecutive columns)
  int ng;
  float* pm;                 // [4*NT, R]
  float* pl;                 // [4*NT, R]
  float* z;                  // [R]
  // EPI_DLOGITS / EPI_CSTORE (round 5: the backward of the fused head)
  const float* row_lse;      // [R] log-sum-exp of every row over the WHOLE vocabulary
  const float* coef;         // [R] d loss / d log-prob of the row's label (0 for rows without loss)
  int col_base;              // first vocabulary entry of this chunk (labels are global ids)
  void* out;                 // EPI_DLOGITS: bf16 [R][out_pitch];  EPI_CSTORE: f32 [R][out_pitch]
  int64_t out_pitch;         // elements
  int out_cols;              // columns that exist in `out` (EPI_DLOGITS: zero-filled from V up to here)
  int accumulate;            // EPI_CSTORE: add to what `out` holds
  // EPI_DS3 (round 5: the similarity backward on the bf16 pipe); row statistics ride in row_lse / coef above
  const float* col_lse;      // [V] column log-sum-exp (global column = col_base + local)
  const float* col_coef;     // [V]
  int64_t diag_offset;       // the positive of row i is global column diag_offset + i
  int64_t out_third;         // elements between the hi / mid / lo images inside a row of `out`
};

constexpr int L8_BUF = 65536;
constexpr int EPI_LSE = 0, EPI_DLOGITS = 2, EPI_CSTORE = 3, EPI_DS3 = 4, EPI_LOGITS = 5;

// Tile order.  The ordered tile list walks the output in bands of gh row tiles - within a band vocabulary tile by vocabulary
// tile, the band's row tiles innermost - and every XCD (workgroup L runs on XCD L % 8, own 4 MB L2) works through ONE
// contiguous run of that list.  The 32 tiles an XCD has in flight are then ~gh row panels x 32/gh vocabulary panels
// (7 + 4.6 instead of 14 + 2.3 panels of 2 MB at 14 row tiles): every panel a CU pulls into the L2 is shared by 4-7 neighbours.
__device__ __forceinline__ void lm_tile_of(const Lm8Params& p, unsigned block, unsigned nblocks, int& mt, int& nt) {
  unsigned g = block;
  if (p.xcd_order) {
    const unsigned q8 = nblocks >> 3, r8 = nblocks & 7u, xcd = block & 7u;
    g = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (block >> 3);
  }
  const unsigned gh = static_cast<unsigned>(p.gh), NT = static_cast<unsigned>(p.NT), MT = static_cast<unsigned>(p.MT);
  const unsigned nfull = MT / gh, full = nfull * gh * NT;
  if (g < full) {
    const unsigned mg = g / (gh * NT), r = g % (gh * NT);
    nt = static_cast<int>(r / gh);
    mt = static_cast<int>(mg * gh + r % gh);
  } else {
    const unsigned rem = MT - nfull * gh, r = g - full;
    nt = static_cast<int>(r / rem);
    mt = static_cast<int>(nfull * gh + r % rem);
  }
}

typedef __attribute__((address_space(3))) void lds_void_t;

// Four independent reductions over the 16 lanes of a DPP row at once (xor 1, xor 2 as quad_perm, then the two mirrors; every
// lane receives the result), written out in assembly: (1) fmaxf() lowers to a canonicalising v_max x, x in front of every
// v_max and keeps the DPP move separate (3 instructions + s_nop 1 per step); (2) a VALU result needs two wait states before a
// DPP instruction may read it - with four chains interleaved the next step of a chain is 3 instructions behind its producer
// and no s_nop is needed.  v_max_f32 returns the other operand for a NaN, as fmaxf does.
#define DALM_DPP4(OP, CTRL)                                                                                          \
  OP " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t" OP " %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"  \
  OP " %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t" OP " %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void row16_max4(float& a, float& b, float& c, float& d) {
  asm volatile("s_nop 1\n\t" DALM_DPP4("v_max_f32_dpp", "quad_perm:[1,0,3,2]") DALM_DPP4("v_max_f32_dpp", "quad_perm:[2,3,0,1]")
               DALM_DPP4("v_max_f32_dpp", "row_half_mirror") DALM_DPP4("v_max_f32_dpp", "row_mirror") "s_nop 0"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void row16_sum4(float& a, float& b, float& c, float& d) {
  asm volatile("s_nop 1\n\t" DALM_DPP4("v_add_f32_dpp", "quad_perm:[1,0,3,2]") DALM_DPP4("v_add_f32_dpp", "quad_perm:[2,3,0,1]")
               DALM_DPP4("v_add_f32_dpp", "row_half_mirror") DALM_DPP4("v_add_f32_dpp", "row_mirror") "s_nop 0"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// max over the 32 lanes of each half-wave for four registers at once: the 16-lane butterflies of row16_max4, then lane 15
// of rows 0 / 2 (which holds its row's maximum) is broadcast into rows 1 / 3 (`row_bcast:15`, a GFX9 DPP control; row_mask
// 0xa enables rows 1 and 3 only): the lanes of rows 1 and 3 end up with the half-wave maximum, rows 0 and 2 keep theirs.
__device__ __forceinline__ void half32_max4(float& a, float& b, float& c, float& d) {
  asm volatile("s_nop 1\n\t" DALM_DPP4("v_max_f32_dpp", "quad_perm:[1,0,3,2]") DALM_DPP4("v_max_f32_dpp", "quad_perm:[2,3,0,1]")
               DALM_DPP4("v_max_f32_dpp", "row_half_mirror") DALM_DPP4("v_max_f32_dpp", "row_mirror")
               "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "v_max_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "v_max_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "v_max_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "s_nop 0"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
#undef DALM_DPP4
// max of four finite-or--inf values without the canonicalising self-max fmaxf() puts in front of every operand
__device__ __forceinline__ float max4_raw(float a, float b, float c, float d) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
  return r;
}

// one direct-to-LDS piece: 64 lanes x 16 bytes -> dst + 16 * lane (dst wave-uniform); its own function so that the host pass,
// which has no such builtin, never sees it inside a kernel body
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* dst, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)dst, 16, static_cast<int>(voff), soff, 0, 0);
}

// Epilogue of a wave's 128 x 128 block, all in registers (VALU only).  A 16-lane DPP row holds 16 of the 32 columns of every
// 32 x 32 tile: it reduces ITS 64 columns of a row to one (max, sum exp) partial - 2 partials per row and wave, no exchange
// between the two 16-lane rows of a half-wave.  After the butterflies every lane of the row holds the result, lane k keeps the
// one of accumulator register k and the 16 rows of a 32 x 32 tile leave in ONE store instruction.
// Labels go through the (now free) LDS: lab_s[256] = label - c0 of the tile's rows when the label falls into the tile's 256
// columns (and below V), else -1: no lane matches.
// ABL & 64 (measurement only): keep the accumulators alive, skip the reduction.
template <int ABL>
__device__ __forceinline__ void lm_tile_epilogue(const Lm8Params& p, f32x16 (&acc)[4][4], unsigned char* lds, int r0, int c0,
                                                 int nt, int wr, int wc, int tid) {
  const int lane = tid & 63, l31 = lane & 31, lhi = lane >> 5, l15 = lane & 15;
  if (ABL & 64) {
    float t0 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t0 += acc[i][j][r];
    if (t0 == 12345.678f) p.z[0] = t0;
    return;
  }
  int* lab_s = reinterpret_cast<int*>(lds);
  {
    const int r = r0 + tid;
    int y = -1;
    if (r < p.R) {
      const int64_t yl = p.labels[r] - c0;               // rows past R and labels outside this tile's columns: no match
      y = (yl >= 0 && yl < 256 && yl + c0 < p.V) ? static_cast<int>(yl) : -1;
    }
    lab_s[tid] = y;
  }
  __syncthreads();
  const int colw = wc * 128 + l31;                       // column of accumulator tile j = colw + 32 j, relative to c0
  const int64_t prow = static_cast<int64_t>((nt * 2 + wc) * 2 + (l31 >> 4)) * p.R;
  float pen[4];             // 0, or -inf for the zero-filled columns >= V of the last vocabulary tile (x + 0 is exact)
#pragma unroll
  for (int j = 0; j < 4; ++j) pen[j] = (c0 + colw + 32 * j < p.V) ? 0.f : -INFINITY;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float keep_m = 0.f, keep_s = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int rl4 = wr * 128 + i * 32 + 8 * g + 4 * lhi;       // rows rl4 .. rl4 + 3 = accumulator registers 4 g .. 4 g + 3
      const int4 lab4 = *reinterpret_cast<const int4*>(lab_s + rl4);
      float x[4][4], m4[4], s4[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) x[rr][j] = acc[i][j][4 * g + rr] + pen[j];
        m4[rr] = max4_raw(x[rr][0], x[rr][1], x[rr][2], x[rr][3]);
      }
      row16_max4(m4[0], m4[1], m4[2], m4[3]);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float mneg = (m4[rr] == -INFINITY) ? 0.f : -m4[rr] * kLog2e;   // a strip entirely beyond V: (max -inf, sum 0)
        float sx = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) sx += __builtin_amdgcn_exp2f(__builtin_fmaf(x[rr][j], kLog2e, mneg));
        s4[rr] = sx;
      }
      row16_sum4(s4[0], s4[1], s4[2], s4[3]);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = 4 * g + rr;
        keep_m = (l15 == r) ? m4[rr] : keep_m;
        keep_s = (l15 == r) ? s4[rr] : keep_s;
        const int dy = (rr == 0 ? lab4.x : (rr == 1 ? lab4.y : (rr == 2 ? lab4.z : lab4.w))) - colw;
        if ((dy & ~96) == 0) {                                // dy in {0, 32, 64, 96}; select chain: no run-time index into x[]
          const int jy = dy >> 5;
          p.z[r0 + rl4 + rr] = jy == 0 ? x[rr][0] : (jy == 1 ? x[rr][1] : (jy == 2 ? x[rr][2] : x[rr][3]));
        }
      }
    }
    const int row = r0 + wr * 128 + i * 32 + (l15 & 3) + 8 * (l15 >> 2) + 4 * lhi;
    if (row < p.R) {
      p.pm[prow + row] = keep_m;
      p.pl[prow + row] = keep_s;
    }
  }
}

// MODE_GMAX epilogue (exact top-k, first pass): one maximum per (row, 32 consecutive columns) = per accumulator tile,
// written to gmax[row][column / 32] - the layout topk_threshold_kernel / topk_refine_kernel (sim.hip) read.  Zero-filled
// columns >= V count as -inf.  VALU only: 5 DPP steps per accumulator register.
__device__ __forceinline__ void lm_tile_epilogue_gmax(const Lm8Params& p, f32x16 (&acc)[4][4], int r0, int c0, int wr, int wc,
                                                      int tid) {
  const int lane = tid & 63, l31 = lane & 31, lhi = lane >> 5, l15 = lane & 15;
  const int colw = wc * 128 + l31;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float pen = (c0 + colw + 32 * j < p.V) ? 0.f : -INFINITY;
    const int g = (c0 + wc * 128 + 32 * j) >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float keep = -INFINITY;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float x0 = acc[i][j][4 * q + 0] + pen, x1 = acc[i][j][4 * q + 1] + pen;
        float x2 = acc[i][j][4 * q + 2] + pen, x3 = acc[i][j][4 * q + 3] + pen;
        half32_max4(x0, x1, x2, x3);
        keep = (l15 == 4 * q + 0) ? x0 : keep;
        keep = (l15 == 4 * q + 1) ? x1 : keep;
        keep = (l15 == 4 * q + 2) ? x2 : keep;
        keep = (l15 == 4 * q + 3) ? x3 : keep;
      }
      const int row = r0 + wr * 128 + i * 32 + (l15 & 3) + 8 * (l15 >> 2) + 4 * lhi;
      if ((l31 >> 4) == 1 && row < p.R && g < p.ng) p.gmax[static_cast<int64_t>(row) * p.ng + g] = keep;
    }
  }
}

// EPI_DLOGITS: the 256 x 256 logits tile becomes coef_r * (exp(x - lse_r) - [column == label_r]) in registers and leaves as
// bf16 into the chunk workspace out[row][c0 + column] (columns >= V up to out_cols are written as zeros: the workspace is the A
// operand of the d(hidden) contraction).  Reference math: dalm/training/utils/train_utils.py:113-138 differentiated (SURVEY 8a:
// dL/dlogits[b,t,:] = (m_bt / M) (softmax - onehot)); coef carries m_bt / M.  Lanes l and l ^ 1 hold adjacent columns: one DPP
// exchange per pair of accumulator registers lets every lane store one dword (2 bf16) instead of two shorts.
// RAW (EPI_LOGITS): the tile leaves as it is, rounded to bf16 - the logits chunk of the two-contraction training path (round 6:
// logits chunk kept in the Infinity Cache, the fused CE kernel turns it into d(logits) in place, dalm_lm_head_dhidden contracts it).
template <bool RAW>
__device__ __forceinline__ void lm_tile_epilogue_dlogits(const Lm8Params& p, f32x16 (&acc)[4][4], unsigned char* lds, int r0, int c0,
                                                         int wr, int wc, int tid) {
  const int lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  float* lse_s = reinterpret_cast<float*>(lds);              // [256] -lse * log2(e)
  float* cf_s = lse_s + 256;                                  // [256] coef
  int* lab_s = reinterpret_cast<int*>(lse_s + 512);           // [256] label - c0 (or -1)
  if constexpr (!RAW) {
    const int r = r0 + tid;
    float l = 0.f, c = 0.f;
    int y = -1;
    if (r < p.R) {
      l = -p.row_lse[r] * kLog2e;
      c = p.coef[r];
      const int64_t yl = p.labels[r] - p.col_base - c0;
      y = (yl >= 0 && yl < 256 && yl + c0 < p.V) ? static_cast<int>(yl) : -1;
    }
    lse_s[tid] = l; cf_s[tid] = c; lab_s[tid] = y;
  }
  __syncthreads();
  const int colw = wc * 128 + l31;                            // column of accumulator tile j = colw + 32 j, relative to c0
  const bool odd = (lane & 1) != 0;
  unsigned short* outp = static_cast<unsigned short*>(p.out);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int rl4 = wr * 128 + i * 32 + 8 * g + 4 * lhi;     // rows rl4 .. rl4 + 3 = accumulator registers 4 g .. 4 g + 3
      float ls[4] = {0.f, 0.f, 0.f, 0.f}, cs[4] = {0.f, 0.f, 0.f, 0.f};
      int ys[4] = {-1, -1, -1, -1};
      if constexpr (!RAW) {
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + rl4);
        const float4 c4 = *reinterpret_cast<const float4*>(cf_s + rl4);
        const int4 y4 = *reinterpret_cast<const int4*>(lab_s + rl4);
        ls[0] = l4.x; ls[1] = l4.y; ls[2] = l4.z; ls[3] = l4.w;
        cs[0] = c4.x; cs[1] = c4.y; cs[2] = c4.z; cs[3] = c4.w;
        ys[0] = y4.x; ys[1] = y4.y; ys[2] = y4.z; ys[3] = y4.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = colw + 32 * j;
        const bool live = c0 + col < p.V;
        float v[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          if constexpr (RAW) {
            v[rr] = live ? acc[i][j][4 * g + rr] : 0.f;
          } else {
            const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[i][j][4 * g + rr], kLog2e, ls[rr]));
            v[rr] = live ? cs[rr] * (pr - (ys[rr] == col ? 1.f : 0.f)) : 0.f;
          }
        }
        // register pairs (0, 1) and (2, 3): the even lane stores row a (its column and the odd neighbour's), the odd lane row b
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float mine_a = v[2 * h], mine_b = v[2 * h + 1];
          const float send = odd ? mine_a : mine_b;
          const float got = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(send), 0xB1, 0xf, 0xf, false));
          const unsigned int word = odd ? pack_bf16x2(got, mine_b) : pack_bf16x2(mine_a, got);
          const int row = r0 + rl4 + 2 * h + (odd ? 1 : 0);
          const int ce = c0 + (col & ~1);                      // even column of the pair, relative to the chunk
          if (row < p.R && ce < p.out_cols)
            *reinterpret_cast<unsigned int*>(outp + static_cast<int64_t>(row) * p.out_pitch + ce) = word;
        }
      }
    }
  }
}

// EPI_DS3: the similarity backward's dS tile (reference: the autograd of get_nt_xent_loss(S) + get_nt_xent_loss(S.t()) +
// the doc term, dalm/training/utils/train_utils.py:76-88,121-124; closed form SURVEY 8a):
//   dS[i][j] = rc_i e^{S_ij - rl_i} + cc_j e^{S_ij - cl_j} - [j == diag_offset + i] (rc_i + cc_j)
// from the f32 accumulator tile of S, split into bf16 hi / mid / lo thirds (24 significand bits) and written as three images
// side by side in a row of `out` - the A operand of the dA = dS . B contraction, which runs through this same kernel
// (SPLIT3 + EPI_CSTORE).
__device__ __forceinline__ void lm_tile_epilogue_ds3(const Lm8Params& p, f32x16 (&acc)[4][4], unsigned char* lds, int r0, int c0,
                                                     int wr, int wc, int tid) {
  const int lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  float* nrl_s = reinterpret_cast<float*>(lds);               // [256] -row_lse * log2(e)
  float* rc_s = nrl_s + 256;                                   // [256] row_coef
  float* ncl_s = nrl_s + 512;                                  // [256] -col_lse * log2(e)
  float* cc_s = nrl_s + 768;                                   // [256] col_coef
  {
    const int r = r0 + tid, c = c0 + tid;
    float a = 0.f, b = 0.f, cl = 0.f, cc = 0.f;
    if (r < p.R) { a = -p.row_lse[r] * kLog2e; b = p.coef[r]; }
    if (c < p.V) { cl = -p.col_lse[p.col_base + c] * kLog2e; cc = p.col_coef[p.col_base + c]; }
    nrl_s[tid] = a; rc_s[tid] = b; ncl_s[tid] = cl; cc_s[tid] = cc;
  }
  __syncthreads();
  const int colw = wc * 128 + l31;
  const bool odd = (lane & 1) != 0;
  unsigned short* outp = static_cast<unsigned short*>(p.out);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = colw + 32 * j;                             // relative to c0
    const bool live = c0 + col < p.V;
    const float ncl = ncl_s[col], ccj = cc_s[col];
    const int64_t gcol = static_cast<int64_t>(p.col_base) + c0 + col;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rl4 = wr * 128 + i * 32 + 8 * g + 4 * lhi;
        const float4 l4 = *reinterpret_cast<const float4*>(nrl_s + rl4);
        const float4 c4 = *reinterpret_cast<const float4*>(rc_s + rl4);
        const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, cs[4] = {c4.x, c4.y, c4.z, c4.w};
        float v[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float t = acc[i][j][4 * g + rr] * kLog2e;
          float d = cs[rr] * __builtin_amdgcn_exp2f(t + ls[rr]) + ccj * __builtin_amdgcn_exp2f(t + ncl);
          const int row = r0 + rl4 + rr;
          if (gcol == p.diag_offset + row) d -= cs[rr] + ccj;
          v[rr] = (live && row < p.R) ? d : 0.f;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float mine_a = v[2 * h], mine_b = v[2 * h + 1];
          const float send = odd ? mine_a : mine_b;
          const float got = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(send), 0xB1, 0xf, 0xf, false));
          const float x0 = odd ? got : mine_a, x1 = odd ? mine_b : got;      // columns (even, odd) of this lane's row
          const int row = r0 + rl4 + 2 * h + (odd ? 1 : 0);
          const int ce = c0 + (col & ~1);
          if (row < p.R && ce < p.out_cols) {
            // thirds of both columns: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); every difference is exact
            const unsigned int hi = pack_bf16x2(x0, x1);
            const float r0a = x0 - __uint_as_float(hi << 16), r0b = x1 - __uint_as_float(hi & 0xffff0000u);
            const unsigned int mid = pack_bf16x2(r0a, r0b);
            const unsigned int lo = pack_bf16x2(r0a - __uint_as_float(mid << 16), r0b - __uint_as_float(mid & 0xffff0000u));
            unsigned short* dst = outp + static_cast<int64_t>(row) * p.out_pitch + ce;
            *reinterpret_cast<unsigned int*>(dst) = hi;
            *reinterpret_cast<unsigned int*>(dst + p.out_third) = mid;
            *reinterpret_cast<unsigned int*>(dst + 2 * p.out_third) = lo;
          }
        }
      }
    }
  }
}

// EPI_CSTORE: the 256 x 256 tile is stored (or added) as f32 into out[row][c0 + column]: d(hidden) += dlogits_chunk . W_chunk
__device__ __forceinline__ void lm_tile_epilogue_cstore(const Lm8Params& p, f32x16 (&acc)[4][4], int r0, int c0, int wr, int wc,
                                                        int tid) {
  const int lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int col0 = c0 + wc * 128 + l31;
  float* base = static_cast<float*>(p.out) + col0;
  const int pitch = static_cast<int>(p.out_pitch);
  const bool ok0 = col0 < p.out_cols, ok1 = col0 + 32 < p.out_cols, ok2 = col0 + 64 < p.out_cols, ok3 = col0 + 96 < p.out_cols;
  const bool add = p.accumulate != 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = r0 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (row >= p.R) continue;
      float* dst = base + static_cast<int64_t>(row) * pitch;
      float v0 = acc[i][0][r], v1 = acc[i][1][r], v2 = acc[i][2][r], v3 = acc[i][3][r];
      if (add) {
        if (ok0) v0 += dst[0];
        if (ok1) v1 += dst[32];
        if (ok2) v2 += dst[64];
        if (ok3) v3 += dst[96];
      }
      if (ok0) dst[0] = v0;
      if (ok1) dst[32] = v1;
      if (ok2) dst[64] = v2;
      if (ok3) dst[96] = v3;
    }
  }
}

// N3 / N0 / N1 / N2: load pieces issued in k-step 3 (right after the barrier) / 0 / 1 / 2; measured best: 8, 8, 0, 0.
// ABL (measurement only, results are garbage unless 0): 1 = no loads in the loop, 2 = no fragment reads, 32 = no barrier,
// 64 = no epilogue.
// SPLIT3 (the bf16x3 similarity, see rowstats_bf16x3 below): the operands are [rows][3 D] bf16 images holding the (hi, mid,
// lo) bf16 thirds of an f32 matrix side by side; the contraction walks SIX segments of D - the six significant products of
// (hi + mid + lo) x (hi + mid + lo), smallest first - and K tile u reads third segA[u / tpd] of H against third segB[u / tpd]
// of W: the same main loop, only the K offset of a tile is looked up instead of being u * 128.
template <int N3, int N0, int N1, int N2, int ABL = 0, bool SPLIT3 = false, bool GMAX = false, int EPI = EPI_LSE>
__global__ __launch_bounds__(256, 1) void lm_head_lse4w_kernel(const Lm8Params p) {
  static_assert(N3 + N0 + N1 + N2 == 16, "16 load pieces per K tile and wave");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * L8_BUF];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int mt, nt;
  lm_tile_of(p, blockIdx.x, gridDim.x, mt, nt);
  const int r0 = mt * 256, c0 = nt * 256;
  const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.H), 0, static_cast<int>(p.bytesH), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, static_cast<int>(p.bytesW), 0x00020000);

  // load piece q (0..7: rows 32 q .. 32 q + 31 of the A tile, 8..15: of the B tile): thread -> row 32 q + tid / 8, chunk tid % 8
  const int srow = tid >> 3;
  const unsigned schunk = static_cast<unsigned>(((tid & 7) ^ ((srow >> 1) & 7)) * 16);
  unsigned voff[16];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    voff[q] = static_cast<unsigned>(r0 + q * 32 + srow) * p.pitchH + schunk;
    voff[8 + q] = static_cast<unsigned>(c0 + q * 32 + srow) * p.pitchW + schunk;
  }
  const int wave_lds = wave * 1024;
#define L4_DMA(BUF, Q, KSA, KSB) if (!(ABL & 1)) lds_dma16((Q) < 8 ? rsH : rsW, lds + (BUF) * L8_BUF + (Q) * 4096 + wave_lds, voff[Q], (Q) < 8 ? (KSA) : (KSB));

  const int f = (l31 >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koff[kk] = ((kk * 2 + lhi) ^ f) * 16;
  const int abase = (wr * 128 + l31) * 128;
  const int bbase = 32768 + (wc * 128 + l31) * 128;

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = p.K / 64;
  // byte offset of K tile u inside a row of H (SIDE 0) / W (SIDE 1); tiles past the end re-read the last one (no branch)
  //   pairs in contraction order: (m,m) (h,l) (l,h) (h,m) (m,h) (h,h) with h = third 0, m = third 1, l = third 2
  //   third of H per pair, 2 bits each: 1,0,2,0,1,0 -> 0x121;  of W: 1,2,0,1,0,0 -> 0x049
  auto ks = [&](int u, int side) -> int {
    u = min(u, nkt - 1);
    if constexpr (!SPLIT3) {
      return u * 128;
    } else {
      const int sg = static_cast<int>((static_cast<unsigned>(u) * p.tpd_inv) >> 24);
      const int w = u - sg * p.tpd;
      const unsigned third = ((side ? 0x049u : 0x121u) >> (2 * sg)) & 3u;
      return static_cast<int>(third * p.seg_bytes) + w * 128;
    }
  };
  bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];

#define L4_READ(BUF, KK, FA, FB)                                                                                      \
  if (!(ABL & 2)) _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
    FA[i] = *reinterpret_cast<const bf16x8*>(lds + (BUF) * L8_BUF + i * 4096 + abase + koff[KK]);                     \
    FB[i] = *reinterpret_cast<const bf16x8*>(lds + (BUF) * L8_BUF + i * 4096 + bbase + koff[KK]);                     \
  }
#define L4_MFMA(FA, FB)                                                                                               \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                          \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i], FB[j], acc[i][j], 0, 0, 0);
  // 16 MFMAs, 8 fragment reads behind the first 8, ND load pieces behind the last ones
#define L4_SCHED(ND)                                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                \
    if (i < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                     \
    if ((ND) <= 8 ? (i >= 8 && i - 8 < (ND)) : (i >= 16 - (ND))) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   \
  }
#define L4_PIECES(BUF, FIRST, COUNT, KSA, KSB)                                                                        \
  _Pragma("unroll") for (int q = 0; q < 16; ++q) if (q >= (FIRST) && q < (FIRST) + (COUNT)) { L4_DMA(BUF, q, KSA, KSB) }

  // one K tile in buffer BUF (OTH = the other buffer); fragments of its k-step 0 are already in fa0 / fb0
#define L4_TILE(BUF, OTH, t)                                                                                          \
  {                                                                                                                   \
    const int ks1a = ks((t) + 1, 0), ks1b = ks((t) + 1, 1), ks2a = ks((t) + 2, 0), ks2b = ks((t) + 2, 1);             \
    L4_READ(BUF, 1, fa1, fb1) L4_PIECES(OTH, N3, N0, ks1a, ks1b) L4_MFMA(fa0, fb0) L4_SCHED(N0)                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    L4_READ(BUF, 2, fa0, fb0) L4_PIECES(OTH, N3 + N0, N1, ks1a, ks1b) L4_MFMA(fa1, fb1) L4_SCHED(N1)                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    L4_READ(BUF, 3, fa1, fb1) L4_PIECES(OTH, N3 + N0 + N1, N2, ks1a, ks1b) L4_MFMA(fa0, fb0) L4_SCHED(N2)              \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                       \
    if (!(ABL & 32)) __builtin_amdgcn_s_barrier();                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    L4_READ(OTH, 0, fa0, fb0) L4_PIECES(BUF, 0, N3, ks2a, ks2b) L4_MFMA(fa1, fb1) L4_SCHED(N3)                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
  }

  // ---- prologue: tile 0 complete, then the first pieces of tile 1 ----
  _Pragma("unroll") for (int q = 0; q < 16; ++q) lds_dma16(q < 8 ? rsH : rsW, lds + q * 4096 + wave_lds, voff[q], ks(0, q < 8 ? 0 : 1));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (ABL & 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa1[i] = *reinterpret_cast<const bf16x8*>(lds + i * 4096 + abase + koff[1]);
      fb1[i] = *reinterpret_cast<const bf16x8*>(lds + i * 4096 + bbase + koff[1]);
      fa0[i] = *reinterpret_cast<const bf16x8*>(lds + i * 4096 + abase + koff[0]);
      fb0[i] = *reinterpret_cast<const bf16x8*>(lds + i * 4096 + bbase + koff[0]);
    }
  } else
  L4_READ(0, 0, fa0, fb0)
  L4_PIECES(1, 0, N3, ks(1, 0), ks(1, 1))
  __builtin_amdgcn_sched_barrier(0);

  int t = 0;
  for (; t + 1 < nkt; t += 2) {
    L4_TILE(0, 1, t)
    L4_TILE(1, 0, t + 1)
  }
  if (t < nkt) L4_TILE(0, 1, t)
#undef L4_TILE
#undef L4_PIECES
#undef L4_SCHED
#undef L4_MFMA
#undef L4_READ
#undef L4_DMA
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // tail re-reads still target the LDS about to be reused
  __builtin_amdgcn_s_barrier();

  if constexpr (EPI == EPI_DS3) lm_tile_epilogue_ds3(p, acc, lds, r0, c0, wr, wc, tid);
  else if constexpr (EPI == EPI_DLOGITS) lm_tile_epilogue_dlogits<false>(p, acc, lds, r0, c0, wr, wc, tid);
  else if constexpr (EPI == EPI_LOGITS) lm_tile_epilogue_dlogits<true>(p, acc, lds, r0, c0, wr, wc, tid);
  else if constexpr (EPI == EPI_CSTORE) lm_tile_epilogue_cstore(p, acc, r0, c0, wr, wc, tid);
  else if constexpr (GMAX) lm_tile_epilogue_gmax(p, acc, r0, c0, wr, wc, tid);
  else lm_tile_epilogue<ABL>(p, acc, lds, r0, c0, nt, wr, wc, tid);
}

// One workgroup = 16 rows: thread t folds the partials p = t/16, t/16 + 16, .. of row t % 16 (16 consecutive rows per
// load instruction = one 64-byte segment of the [P][R] partial arrays, 8 loads in flight), then the 16 folds of a row are
// combined through LDS in fixed order.  R/16 workgroups: the whole chip takes part (the round-2 form used R/64 = 56).
__global__ __launch_bounds__(256) void lm_head_lse_merge_kernel(const float* __restrict__ pm, const float* __restrict__ pl,
                                                                const float* __restrict__ z,
                                                                const int64_t* __restrict__ labels, int R, int V, int P,
                                                                float* __restrict__ row_lse, float* __restrict__ row_nll,
                                                                float* __restrict__ z_out = nullptr) {
  __shared__ float ms[16][17], ls[16][17];
  const int rl = threadIdx.x & 15, sub = threadIdx.x >> 4;
  const int row = blockIdx.x * 16 + rl;
  const int rr = min(row, R - 1);
  float m = -INFINITY, l = 0.f;
  for (int p0 = sub; p0 < P; p0 += 128) {
    float vm[8], vl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int pp = min(p0 + 16 * u, P - 1);
      vm[u] = pm[static_cast<int64_t>(pp) * R + rr];
      vl[u] = pl[static_cast<int64_t>(pp) * R + rr];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (p0 + 16 * u < P && vm[u] != -INFINITY) {
        const float mn = fmaxf(m, vm[u]);
        l = l * __builtin_amdgcn_exp2f((m - mn) * kLog2e) + vl[u] * __builtin_amdgcn_exp2f((vm[u] - mn) * kLog2e);
        m = mn;
      }
    }
  }
  ms[sub][rl] = m;
  ls[sub][rl] = l;
  __syncthreads();
  if (sub == 0 && row < R) {
    float M = ms[0][rl];
#pragma unroll
    for (int i = 1; i < 16; ++i) M = fmaxf(M, ms[i][rl]);
    float L = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) L += (ms[i][rl] == -INFINITY) ? 0.f : ls[i][rl] * __builtin_amdgcn_exp2f((ms[i][rl] - M) * kLog2e);
    const float lse = M + __logf(L);
    row_lse[row] = lse;
    const int64_t y = labels[row];
    if (row_nll) row_nll[row] = (y < 0) ? 0.f : (y < V ? lse - z[row] : __builtin_nanf(""));
    if (z_out) z_out[row] = z[row];      // the similarity path wants the label's logit itself (the diagonal score)
  }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16x3: the f32-accurate similarity on the bf16 matrix cores.
//   reference: S = matmul(Q, P^T) * scale, get_cosine_sim, dalm/training/utils/train_utils.py:76-77 (rows of log-sum-exp:
//   :80-88); evaluation scores dalm/eval/utils.py:44-68.
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): every difference is exact in f32 and the
// three thirds carry 24 significand bits.  Of the nine products of two such sums the six down to 2^-16 relative are kept
// (hi lo, lo hi and mid mid are the 2^-16 ones; mid lo, lo mid, lo lo <= 2^-24 are dropped); a product of two bf16 values is
// exact in f32, the matrix core accumulates in f32.  Laid out along K - smallest products first - the whole thing is ONE
// bf16 contraction of depth 6 D that runs through the lm_head kernel above, row log-sum-exp epilogue included:
// 6 x the flops of the f32 kernel on a pipe with 16 x its rate.  `scale` is folded into A before the split.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split3_bf16_kernel(const float* __restrict__ X, int rows, int D, float scale,
                                                          unsigned short* __restrict__ out, int64_t* __restrict__ labels,
                                                          int64_t diag_offset) {
  const int per_row = D >> 3;                                   // threads per row: 8 elements each
  const int64_t gid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t row = gid / per_row;
  const int c = static_cast<int>(gid % per_row) * 8;
  if (row >= rows) return;
  if (labels && c == 0) labels[row] = diag_offset + row;
  const float4 a = *reinterpret_cast<const float4*>(X + row * D + c);
  const float4 b = *reinterpret_cast<const float4*>(X + row * D + c + 4);
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  float h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = __fmul_rn(x[e], scale);
    h[e] = bf16_to_f32(f32_to_bf16(v));
    const float r1 = v - h[e];                                  // exact
    m[e] = bf16_to_f32(f32_to_bf16(r1));
    l[e] = (r1 - m[e]);                                         // exact; rounded to bf16 by the pack below
  }
  unsigned short* o = out + row * (3 * static_cast<int64_t>(D)) + c;
  *reinterpret_cast<uint4*>(o) = make_uint4(pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3]), pack_bf16x2(h[4], h[5]), pack_bf16x2(h[6], h[7]));
  *reinterpret_cast<uint4*>(o + D) = make_uint4(pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]), pack_bf16x2(m[4], m[5]), pack_bf16x2(m[6], m[7]));
  *reinterpret_cast<uint4*>(o + 2 * D) = make_uint4(pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3]), pack_bf16x2(l[4], l[5]), pack_bf16x2(l[6], l[7]));
}

// dst[c][r] = src[r][c] for a [rows, cols] bf16 matrix (rows of `ld_src` elements) -> [cols][ld_dst], columns r >= rows of dst
// zero-filled up to ld_dst: a vocabulary chunk of the lm_head weight [Vc, K] becomes the K-contiguous B operand [K, Vc_pad] of the
// d(hidden) contraction.  64 x 64 tiles through LDS, 128-byte rows on both sides.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const unsigned short* __restrict__ src, int rows, int cols, int64_t ld_src,
                                                             unsigned short* __restrict__ dst, int64_t ld_dst) {
  __shared__ unsigned short tile[64][66];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c_in = blockIdx.x * 64 + tx;                      // source column
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int r = blockIdx.y * 64 + ty + 4 * k;               // source row
    tile[ty + 4 * k][tx] = (r < rows && c_in < cols) ? src[static_cast<int64_t>(r) * ld_src + c_in] : static_cast<unsigned short>(0);
  }
  __syncthreads();
  const int r_out = blockIdx.y * 64 + tx;                     // destination column = source row
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int c_out = blockIdx.x * 64 + ty + 4 * k;           // destination row = source column
    if (c_out < cols && r_out < ld_dst) dst[static_cast<int64_t>(c_out) * ld_dst + r_out] = tile[tx][ty + 4 * k];
  }
}

// out[d][third * ncp + j] = third(scale * B[j0 + j][d]) for j < nc, zero for nc <= j < ncp: the K-contiguous (j-contiguous)
// bf16x3 image of a block of rows of B, transposed - the B operand of dA = dS . B.  64 x 64 tiles through LDS.
__global__ __launch_bounds__(256) void split3_transpose_kernel(const float* __restrict__ Bm, int64_t j0, int nc, int D, float scale,
                                                               unsigned short* __restrict__ out, int ncp) {
  __shared__ float tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int d_in = blockIdx.y * 64 + tx;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int j = blockIdx.x * 64 + ty + 4 * k;
    tile[ty + 4 * k][tx] = (j < nc && d_in < D) ? __fmul_rn(Bm[(j0 + j) * static_cast<int64_t>(D) + d_in], scale) : 0.f;
  }
  __syncthreads();
  const int j_out = blockIdx.x * 64 + tx;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int d_out = blockIdx.y * 64 + ty + 4 * k;
    if (d_out >= D || j_out >= ncp) continue;
    const float v = tile[tx][ty + 4 * k];
    const float h = bf16_to_f32(f32_to_bf16(v));
    const float r1 = v - h;
    const float m = bf16_to_f32(f32_to_bf16(r1));
    unsigned short* o = out + static_cast<int64_t>(d_out) * (3 * static_cast<int64_t>(ncp)) + j_out;
    o[0] = f32_to_bf16(v);
    o[ncp] = f32_to_bf16(r1);
    o[2 * static_cast<int64_t>(ncp)] = f32_to_bf16(r1 - m);
  }
}

// dst (bf16) = src (f32) * scale, 8 elements per thread
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int64_t n8) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n8) return;
  const float4 a = *reinterpret_cast<const float4*>(src + i * 8), b = *reinterpret_cast<const float4*>(src + i * 8 + 4);
  *reinterpret_cast<uint4*>(dst + i * 8) = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
}

}  // namespace
}  // namespace dalm

using namespace dalm;

extern "C" size_t dalm_lm_head_lse_workspace_bytes(int64_t R, int64_t V) {
  if (R <= 0 || V <= 0) return 0;
  const int64_t NT = (V + LBN - 1) / LBN;       // 2 partials per 128 columns, or 4 per 256: <= 2 NT + 2 either way
  return static_cast<size_t>((4 * NT + 8) * 2 + 1) * static_cast<size_t>(R) * sizeof(float);
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
  hipStream_t s = as_stream(stream);
  // round-3 kernel (needs 32-bit buffer offsets); DALM_LM_HEAD_GEN=2 selects the round-2 kernel below
  static const char* gen_env = getenv("DALM_LM_HEAD_GEN");
  const int gen = gen_env ? atoi(gen_env) : 3;
  const uint64_t bytesH = static_cast<uint64_t>(R + 256) * K * 2, bytesW = static_cast<uint64_t>(V + 256) * K * 2;
  if (gen >= 3 && bytesH < 0xffffff00ull && bytesW < 0xffffff00ull) {
    Lm8Params q;
    q.H = hidden; q.W = weight; q.labels = labels;
    q.R = static_cast<int>(R); q.V = static_cast<int>(V); q.K = static_cast<int>(K);
    q.MT = static_cast<int>((R + 255) / 256); q.NT = static_cast<int>((V + 255) / 256);
    static const char* xcd_env8 = getenv("DALM_LM_HEAD_XCD");
    q.xcd_order = xcd_env8 ? atoi(xcd_env8) != 0 : 1;
    static const char* gh_env = getenv("DALM_LM_HEAD_GH");
    const int bands = (q.MT + 7) / 8;                                  // bands of <= 8 row tiles, as even as possible
    q.gh = gh_env ? atoi(gh_env) : (q.MT + bands - 1) / bands;
    if (q.gh < 1 || q.gh > q.MT) q.gh = q.MT;
    q.bytesH = static_cast<unsigned>(static_cast<uint64_t>(R) * K * 2);
    q.bytesW = static_cast<unsigned>(static_cast<uint64_t>(V) * K * 2);
    q.pitchH = q.pitchW = static_cast<unsigned>(K * 2);
    q.tpd = 0; q.tpd_inv = 0; q.seg_bytes = 0; q.gmax = nullptr; q.ng = 0;
    float* f8 = static_cast<float*>(ws);
    const int64_t P4 = 4ll * q.NT;
    q.pm = f8; q.pl = f8 + P4 * R; q.z = f8 + 2 * P4 * R;
    const dim3 g4(static_cast<unsigned>(q.MT) * q.NT);
    static const char* var_env = getenv("DALM_LM_HEAD_ABL");          // measurement only (tools/lm_head_ablate.py)
    const int var = var_env ? atoi(var_env) : 0;
    switch (var) {
      case 0: hipLaunchKernelGGL((lm_head_lse4w_kernel<8, 8, 0, 0>), g4, dim3(256), 0, s, q); break;
      case 1: hipLaunchKernelGGL((lm_head_lse4w_kernel<8, 8, 0, 0, 1>), g4, dim3(256), 0, s, q); break;
      case 2: hipLaunchKernelGGL((lm_head_lse4w_kernel<8, 8, 0, 0, 2>), g4, dim3(256), 0, s, q); break;
      case 35: hipLaunchKernelGGL((lm_head_lse4w_kernel<8, 8, 0, 0, 35>), g4, dim3(256), 0, s, q); break;
      case 64: hipLaunchKernelGGL((lm_head_lse4w_kernel<8, 8, 0, 0, 64>), g4, dim3(256), 0, s, q); break;
      case 99: hipLaunchKernelGGL((lm_head_lse4w_kernel<8, 8, 0, 0, 99>), g4, dim3(256), 0, s, q); break;
      case 100: hipLaunchKernelGGL((lm_head_lse4w_kernel<6, 5, 5, 0>), g4, dim3(256), 0, s, q); break;   // load-piece placements
      case 101: hipLaunchKernelGGL((lm_head_lse4w_kernel<4, 4, 4, 4>), g4, dim3(256), 0, s, q); break;
      case 102: hipLaunchKernelGGL((lm_head_lse4w_kernel<12, 4, 0, 0>), g4, dim3(256), 0, s, q); break;
      case 103: hipLaunchKernelGGL((lm_head_lse4w_kernel<0, 8, 8, 0>), g4, dim3(256), 0, s, q); break;
      default: return fail(DALM_E_SHAPE, __func__, "unknown DALM_LM_HEAD_ABL");
    }
    hipLaunchKernelGGL(lm_head_lse_merge_kernel, dim3(static_cast<unsigned>((R + 15) / 16)), dim3(256), 0, s, q.pm, q.pl,
                       q.z, labels, q.R, q.V, static_cast<int>(P4), row_lse, row_nll);
    return check_launch(__func__);
  }
  // 256-row tiles (fewer LDS bytes per MFMA) once they still give every CU several tiles; 128-row tiles below that
  static const char* tm_env = getenv("DALM_LM_HEAD_TM");
  const int64_t NT = (V + LBN - 1) / LBN;
  int tm = (((R + 255) / 256) * NT >= 1024) ? 4 : 2;
  if (tm_env) tm = (atoi(tm_env) == 4) ? 4 : 2;
  const int64_t LBM = 64 * tm, MT = (R + LBM - 1) / LBM;
  DALM_REQUIRE(MT * NT <= 0x7fffffffll, DALM_E_SHAPE, "too many tiles for one launch");
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
  hipLaunchKernelGGL(lm_head_lse_merge_kernel, dim3(static_cast<unsigned>((R + 15) / 16)), dim3(256), 0, s, p.pm, p.pl,
                     p.z, labels, p.R, p.V, static_cast<int>(2 * NT), row_lse, row_nll);
  return check_launch(__func__);
}


// ---- bf16x3 similarity row statistics (see split3_bf16_kernel) ------------------------------------------------------------
namespace {
inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }
struct X3Layout { size_t a3, b3, labels, pm, pl, z, total; int64_t P4; };
inline X3Layout x3_layout(int64_t m, int64_t n, int64_t D) {
  X3Layout L{};
  const int64_t NT = (n + 255) / 256;
  L.P4 = 4 * NT;
  size_t o = 256;                                            // slack for aligning the caller's pointer up to 256 bytes
  L.a3 = o; o += up256(static_cast<size_t>(m) * 3 * D * 2);
  L.b3 = o; o += up256(static_cast<size_t>(n) * 3 * D * 2);
  L.labels = o; o += up256(static_cast<size_t>(m) * 8);
  L.pm = o; o += up256(static_cast<size_t>(L.P4) * m * 4);
  L.pl = o; o += up256(static_cast<size_t>(L.P4) * m * 4);
  L.z = o; o += up256(static_cast<size_t>(m) * 4);
  L.total = o;
  return L;
}
}  // namespace

extern "C" int dalm_sim_rowstats_bf16x3_supported(int64_t m, int64_t n, int64_t D) {
  if (m <= 0 || n <= 0 || D < 64 || D % 64 != 0) return 0;
  if (D / 64 > 1024) return 0;                                                       // the reciprocal of the tile lookup
  if (static_cast<uint64_t>(m + 256) * 3 * D * 2 >= 0xffffff00ull) return 0;           // 32-bit buffer offsets
  if (static_cast<uint64_t>(n + 256) * 3 * D * 2 >= 0xffffff00ull) return 0;
  return 1;
}

extern "C" size_t dalm_sim_rowstats_bf16x3_workspace_bytes(int64_t m, int64_t n, int64_t D) {
  if (!dalm_sim_rowstats_bf16x3_supported(m, n, D)) return 0;
  return x3_layout(m, n, D).total;
}

extern "C" int dalm_sim_rowstats_bf16x3(const float* A, const float* Bm, int64_t m, int64_t n, int64_t D, float scale,
                                        int64_t diag_offset, float* row_lse, float* diag, void* ws, size_t ws_bytes,
                                        dalm_stream_t stream) {
  DALM_REQUIRE(A && Bm && row_lse && diag && ws, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dalm_sim_rowstats_bf16x3_supported(m, n, D), DALM_E_SHAPE,
               "bf16x3 similarity needs D % 64 == 0 and operand images below 4 GB (dalm_sim_rowstats_bf16x3_supported)");
  DALM_REQUIRE(diag_offset >= 0 && diag_offset + m <= n, DALM_E_SHAPE, "diag_offset + m must be <= n");
  DALM_REQUIRE(reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(Bm) % 16 == 0, DALM_E_ALIGN,
               "A / B must be 16-byte aligned");
  const X3Layout L = x3_layout(m, n, D);
  DALM_REQUIRE(ws_bytes >= L.total, DALM_E_WORKSPACE, "workspace too small");
  hipStream_t s = as_stream(stream);
  unsigned char* base = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(ws) + 255) / 256 * 256) - 256;
  // (offsets in L start at 256: base + L.x is 256-byte aligned and inside the caller's buffer)
  auto* a3 = reinterpret_cast<unsigned short*>(base + L.a3);
  auto* b3 = reinterpret_cast<unsigned short*>(base + L.b3);
  auto* labels = reinterpret_cast<int64_t*>(base + L.labels);
  const int per_row = static_cast<int>(D / 8);
  hipLaunchKernelGGL(split3_bf16_kernel, dim3(static_cast<unsigned>((m * per_row + 255) / 256)), dim3(256), 0, s, A,
                     static_cast<int>(m), static_cast<int>(D), scale, a3, labels, diag_offset);
  hipLaunchKernelGGL(split3_bf16_kernel, dim3(static_cast<unsigned>((n * per_row + 255) / 256)), dim3(256), 0, s, Bm,
                     static_cast<int>(n), static_cast<int>(D), 1.0f, b3, static_cast<int64_t*>(nullptr), static_cast<int64_t>(0));
  Lm8Params q;
  q.H = a3; q.W = b3; q.labels = labels;
  q.R = static_cast<int>(m); q.V = static_cast<int>(n); q.K = static_cast<int>(6 * D);
  q.MT = static_cast<int>((m + 255) / 256); q.NT = static_cast<int>((n + 255) / 256);
  q.xcd_order = 1;
  const int bands = (q.MT + 7) / 8;
  q.gh = (q.MT + bands - 1) / bands;
  if (q.gh < 1 || q.gh > q.MT) q.gh = q.MT;
  q.pitchH = q.pitchW = static_cast<unsigned>(3 * D * 2);
  q.bytesH = static_cast<unsigned>(static_cast<uint64_t>(m) * 3 * D * 2);
  q.bytesW = static_cast<unsigned>(static_cast<uint64_t>(n) * 3 * D * 2);
  q.tpd = static_cast<int>(D / 64);
  q.tpd_inv = static_cast<unsigned>(((1u << 24) + q.tpd - 1) / q.tpd);
  q.seg_bytes = static_cast<unsigned>(D * 2);
  q.gmax = nullptr; q.ng = 0;
  q.pm = reinterpret_cast<float*>(base + L.pm); q.pl = reinterpret_cast<float*>(base + L.pl);
  q.z = reinterpret_cast<float*>(base + L.z);
  DALM_REQUIRE(static_cast<int64_t>(q.MT) * q.NT <= 0x7fffffffll, DALM_E_SHAPE, "too many tiles for one launch");
  hipLaunchKernelGGL((lm_head_lse4w_kernel<8, 8, 0, 0, 0, true>), dim3(static_cast<unsigned>(q.MT) * q.NT), dim3(256), 0, s, q);
  hipLaunchKernelGGL(lm_head_lse_merge_kernel, dim3(static_cast<unsigned>((m + 15) / 16)), dim3(256), 0, s, q.pm, q.pl, q.z,
                     labels, q.R, q.V, static_cast<int>(L.P4), row_lse, static_cast<float*>(nullptr), diag);
  return check_launch(__func__);
}


// ---- bf16x3 group maxima for the exact top-k (first pass of dalm_sim_topk): gmax[m][ng], ng groups of 32 columns ----
extern "C" size_t dalm_x3_group_max_workspace_bytes(int64_t m, int64_t n, int64_t D) {
  if (!dalm_sim_rowstats_bf16x3_supported(m, n, D)) return 0;
  return 256 + up256(static_cast<size_t>(m) * 3 * D * 2) + up256(static_cast<size_t>(n) * 3 * D * 2);
}

extern "C" int dalm_x3_group_max(const float* A, const float* Bm, int64_t m, int64_t n, int64_t D, float scale, float* gmax,
                                 int64_t ng, void* ws, size_t ws_bytes, dalm_stream_t stream) {
  DALM_REQUIRE(A && Bm && gmax && ws, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dalm_sim_rowstats_bf16x3_supported(m, n, D), DALM_E_SHAPE, "shape outside the bf16x3 form");
  DALM_REQUIRE(ws_bytes >= dalm_x3_group_max_workspace_bytes(m, n, D), DALM_E_WORKSPACE, "workspace too small");
  DALM_REQUIRE(ng > 0 && ng <= 0x7fffffffll, DALM_E_SHAPE, "need ng > 0");
  hipStream_t s = as_stream(stream);
  unsigned char* base = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(ws) + 255) / 256 * 256);
  auto* a3 = reinterpret_cast<unsigned short*>(base);
  auto* b3 = reinterpret_cast<unsigned short*>(base + up256(static_cast<size_t>(m) * 3 * D * 2));
  const int per_row = static_cast<int>(D / 8);
  hipLaunchKernelGGL(split3_bf16_kernel, dim3(static_cast<unsigned>((m * per_row + 255) / 256)), dim3(256), 0, s, A,
                     static_cast<int>(m), static_cast<int>(D), scale, a3, static_cast<int64_t*>(nullptr), static_cast<int64_t>(0));
  hipLaunchKernelGGL(split3_bf16_kernel, dim3(static_cast<unsigned>((n * per_row + 255) / 256)), dim3(256), 0, s, Bm,
                     static_cast<int>(n), static_cast<int>(D), 1.0f, b3, static_cast<int64_t*>(nullptr), static_cast<int64_t>(0));
  Lm8Params q{};
  q.H = a3; q.W = b3; q.labels = nullptr;
  q.R = static_cast<int>(m); q.V = static_cast<int>(n); q.K = static_cast<int>(6 * D);
  q.MT = static_cast<int>((m + 255) / 256); q.NT = static_cast<int>((n + 255) / 256);
  q.xcd_order = 1;
  const int bands = (q.MT + 7) / 8;
  q.gh = (q.MT + bands - 1) / bands;
  if (q.gh < 1 || q.gh > q.MT) q.gh = q.MT;
  q.pitchH = q.pitchW = static_cast<unsigned>(3 * D * 2);
  q.bytesH = static_cast<unsigned>(static_cast<uint64_t>(m) * 3 * D * 2);
  q.bytesW = static_cast<unsigned>(static_cast<uint64_t>(n) * 3 * D * 2);
  q.tpd = static_cast<int>(D / 64);
  q.tpd_inv = static_cast<unsigned>(((1u << 24) + q.tpd - 1) / q.tpd);
  q.seg_bytes = static_cast<unsigned>(D * 2);
  q.gmax = gmax; q.ng = static_cast<int>(ng);
  DALM_REQUIRE(static_cast<int64_t>(q.MT) * q.NT <= 0x7fffffffll, DALM_E_SHAPE, "too many tiles for one launch");
  hipLaunchKernelGGL((lm_head_lse4w_kernel<8, 8, 0, 0, 0, true, true>), dim3(static_cast<unsigned>(q.MT) * q.NT), dim3(256), 0, s, q);
  return check_launch(__func__);
}


// ---- round 5: the backward of the fused head -----------------------------------------------------------------
namespace {
inline void lm8_geometry(Lm8Params& q, int64_t R, int64_t V) {
  q.R = static_cast<int>(R); q.V = static_cast<int>(V);
  q.MT = static_cast<int>((R + 255) / 256); q.NT = static_cast<int>((V + 255) / 256);
  static const char* xcd_env8 = getenv("DALM_LM_HEAD_XCD");
  q.xcd_order = xcd_env8 ? atoi(xcd_env8) != 0 : 1;
  const int bands = (q.MT + 7) / 8;
  q.gh = (q.MT + bands - 1) / bands;
  if (q.gh < 1 || q.gh > q.MT) q.gh = q.MT;
  q.tpd = 0; q.tpd_inv = 0; q.seg_bytes = 0; q.gmax = nullptr; q.ng = 0;
  q.pm = q.pl = q.z = nullptr;
}
}  // namespace

extern "C" int dalm_lm_head_dlogits(const void* hidden, const void* weight_chunk, const int64_t* labels, const float* row_lse,
                                    const float* coef, int64_t R, int64_t Vc, int64_t K, int64_t col_base, void* dl,
                                    int64_t pitch, dalm_stream_t stream) {
  DALM_REQUIRE(hidden && weight_chunk && labels && row_lse && coef && dl, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(R > 0 && Vc > 0 && K > 0 && K % 64 == 0 && col_base >= 0, DALM_E_SHAPE, "need R, Vc > 0 and K a positive multiple of 64");
  DALM_REQUIRE(pitch >= Vc && pitch % 64 == 0 && pitch <= (Vc + 255) / 256 * 256, DALM_E_SHAPE,
               "pitch must be a multiple of 64 in [Vc, round_up(Vc, 256)]");
  DALM_REQUIRE(reinterpret_cast<uintptr_t>(hidden) % 16 == 0 && reinterpret_cast<uintptr_t>(weight_chunk) % 16 == 0 &&
               reinterpret_cast<uintptr_t>(dl) % 4 == 0, DALM_E_ALIGN, "hidden / weight must be 16-byte aligned");
  const uint64_t bytesH = static_cast<uint64_t>(R + 256) * K * 2, bytesW = static_cast<uint64_t>(Vc + 256) * K * 2;
  DALM_REQUIRE(bytesH < 0xffffff00ull && bytesW < 0xffffff00ull, DALM_E_SHAPE, "operands above 4 GB: use smaller chunks");
  Lm8Params q;
  q.H = hidden; q.W = weight_chunk; q.labels = labels; q.K = static_cast<int>(K);
  lm8_geometry(q, R, Vc);
  q.bytesH = static_cast<unsigned>(static_cast<uint64_t>(R) * K * 2);
  q.bytesW = static_cast<unsigned>(static_cast<uint64_t>(Vc) * K * 2);
  q.pitchH = q.pitchW = static_cast<unsigned>(K * 2);
  q.row_lse = row_lse; q.coef = coef; q.col_base = static_cast<int>(col_base);
  q.out = dl; q.out_pitch = pitch; q.out_cols = static_cast<int>(pitch); q.accumulate = 0;
  DALM_REQUIRE(static_cast<int64_t>(q.MT) * q.NT <= 0x7fffffffll, DALM_E_SHAPE, "too many tiles for one launch");
  hipLaunchKernelGGL((lm_head_lse4w_kernel<8, 8, 0, 0, 0, false, false, EPI_DLOGITS>), dim3(static_cast<unsigned>(q.MT) * q.NT),
                     dim3(256), 0, as_stream(stream), q);
  return check_launch(__func__);
}

extern "C" int dalm_lm_head_logits(const void* hidden, const void* weight, int64_t R, int64_t V, int64_t K, void* logits, int64_t pitch,
                                   dalm_stream_t stream) {
  DALM_REQUIRE(hidden && weight && logits, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(R > 0 && V > 0 && K > 0 && K % 64 == 0, DALM_E_SHAPE, "need R, V > 0 and K a positive multiple of 64");
  DALM_REQUIRE(pitch >= V && pitch % 2 == 0, DALM_E_SHAPE, "pitch must be even and at least V");
  DALM_REQUIRE(reinterpret_cast<uintptr_t>(hidden) % 16 == 0 && reinterpret_cast<uintptr_t>(weight) % 16 == 0 &&
               reinterpret_cast<uintptr_t>(logits) % 4 == 0, DALM_E_ALIGN, "hidden / weight must be 16-byte aligned");
  const uint64_t bytesH = static_cast<uint64_t>(R + 256) * K * 2, bytesW = static_cast<uint64_t>(V + 256) * K * 2;
  DALM_REQUIRE(bytesH < 0xffffff00ull && bytesW < 0xffffff00ull, DALM_E_SHAPE, "operands above 4 GB: use smaller chunks");
  Lm8Params q;
  q.H = hidden; q.W = weight; q.labels = nullptr; q.K = static_cast<int>(K);
  lm8_geometry(q, R, V);
  q.bytesH = static_cast<unsigned>(static_cast<uint64_t>(R) * K * 2);
  q.bytesW = static_cast<unsigned>(static_cast<uint64_t>(V) * K * 2);
  q.pitchH = q.pitchW = static_cast<unsigned>(K * 2);
  q.row_lse = nullptr; q.coef = nullptr; q.col_base = 0;
  q.out = logits; q.out_pitch = pitch; q.out_cols = static_cast<int>(V % 2 ? V + 1 : V) <= pitch ? static_cast<int>(V % 2 ? V + 1 : V) : static_cast<int>(pitch);
  q.accumulate = 0;
  DALM_REQUIRE(static_cast<int64_t>(q.MT) * q.NT <= 0x7fffffffll, DALM_E_SHAPE, "too many tiles for one launch");
  hipLaunchKernelGGL((lm_head_lse4w_kernel<8, 8, 0, 0, 0, false, false, EPI_LOGITS>), dim3(static_cast<unsigned>(q.MT) * q.NT),
                     dim3(256), 0, as_stream(stream), q);
  return check_launch(__func__);
}

extern "C" int dalm_lm_head_dhidden(const void* dl, const void* wt, int64_t R, int64_t Vp, int64_t K, float* dh, int accumulate,
                                    dalm_stream_t stream) {
  DALM_REQUIRE(dl && wt && dh, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(R > 0 && Vp > 0 && K > 0 && Vp % 64 == 0, DALM_E_SHAPE, "need R, K > 0 and a chunk width that is a multiple of 64");
  DALM_REQUIRE(reinterpret_cast<uintptr_t>(dl) % 16 == 0 && reinterpret_cast<uintptr_t>(wt) % 16 == 0, DALM_E_ALIGN,
               "dl / wt must be 16-byte aligned");
  const uint64_t bytesH = static_cast<uint64_t>(R + 256) * Vp * 2, bytesW = static_cast<uint64_t>(K + 256) * Vp * 2;
  DALM_REQUIRE(bytesH < 0xffffff00ull && bytesW < 0xffffff00ull, DALM_E_SHAPE, "operands above 4 GB: use smaller chunks");
  Lm8Params q;
  q.H = dl; q.W = wt; q.labels = nullptr; q.K = static_cast<int>(Vp);          // the contraction runs over the chunk's vocabulary
  lm8_geometry(q, R, K);
  q.bytesH = static_cast<unsigned>(static_cast<uint64_t>(R) * Vp * 2);
  q.bytesW = static_cast<unsigned>(static_cast<uint64_t>(K) * Vp * 2);
  q.pitchH = q.pitchW = static_cast<unsigned>(Vp * 2);
  q.row_lse = nullptr; q.coef = nullptr; q.col_base = 0;
  q.out = dh; q.out_pitch = K; q.out_cols = static_cast<int>(K); q.accumulate = accumulate;
  DALM_REQUIRE(static_cast<int64_t>(q.MT) * q.NT <= 0x7fffffffll, DALM_E_SHAPE, "too many tiles for one launch");
  hipLaunchKernelGGL((lm_head_lse4w_kernel<8, 8, 0, 0, 0, false, false, EPI_CSTORE>), dim3(static_cast<unsigned>(q.MT) * q.NT),
                     dim3(256), 0, as_stream(stream), q);
  return check_launch(__func__);
}

extern "C" int dalm_transpose_bf16(const void* src, int64_t rows, int64_t cols, int64_t ld_src, void* dst, int64_t ld_dst,
                                   dalm_stream_t stream) {
  DALM_REQUIRE(src && dst, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows && rows <= 0x7fffffffll && cols <= 0x7fffffffll,
               DALM_E_SHAPE, "need rows, cols > 0, ld_src >= cols, ld_dst >= rows");
  const dim3 grid(static_cast<unsigned>((cols + 63) / 64), static_cast<unsigned>((ld_dst + 63) / 64));
  DALM_REQUIRE(grid.y <= 65535, DALM_E_SHAPE, "too many rows for one launch");
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, as_stream(stream), static_cast<const unsigned short*>(src),
                     static_cast<int>(rows), static_cast<int>(cols), ld_src, static_cast<unsigned short*>(dst), ld_dst);
  return check_launch(__func__);
}

extern "C" int dalm_f32_to_bf16(const float* src, void* dst, int64_t n, dalm_stream_t stream) {
  DALM_REQUIRE(src && dst, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(n > 0 && n % 8 == 0, DALM_E_SHAPE, "need a positive multiple of 8 elements");
  DALM_REQUIRE(reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0, DALM_E_ALIGN,
               "src / dst must be 16-byte aligned");
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(static_cast<unsigned>((n / 8 + 255) / 256)), dim3(256), 0, as_stream(stream), src,
                     static_cast<unsigned short*>(dst), n / 8);
  return check_launch(__func__);
}


// ---- round 5: the similarity BACKWARD on the bf16 pipe ---------------------------------------------------
//   reference: the autograd of get_cosine_sim + get_nt_xent_loss (x2) + the doc term, dalm/training/utils/train_utils.py:76-88
// dA = scale * dS . B with dS rebuilt from S = scale * A . B^T: both contractions as bf16x3 (six significant products, f32
// accumulation) through the lm_head core - per block of `nc` columns of S: (1) S tile -> dS thirds image (EPI_DS3), (2) the
// transposed thirds image of the block's rows of B, (3) dA (+)= dS_blk . B_blk (SPLIT3 + EPI_CSTORE).  24 m n D bf16 flops
// instead of 6 m n D on the f32 pipe (1/16 of the rate).
namespace {
// Columns of S per block (a multiple of 256); the dS image is m x 3 x block x 2 bytes.  Larger blocks = fewer, fuller launches
// (16384^2, profiles/r05_sim_grad_x3.txt: block 2048 163 TF f32-equivalent, 4096 179, 8192 188): as large as a 1 GiB dS image
// allows, between 2048 and 8192.  DALM_X3_GRAD_BLOCK overrides (measurement).
inline int64_t x3_grad_block(int64_t m) {
  static const int64_t forced = [] { const char* e = getenv("DALM_X3_GRAD_BLOCK"); return e ? atoll(e) : 0ll; }();
  int64_t b = forced > 0 ? forced : (int64_t(1) << 30) / (6 * (m > 0 ? m : 1));
  if (forced <= 0) { if (b > 8192) b = 8192; if (b < 2048) b = 2048; }
  if (b < 256) b = 256;
  return b / 256 * 256;
}
struct X3GradLayout { size_t a3, b3, ds3, bt3, total; int64_t ncp; };
inline X3GradLayout x3_grad_layout(int64_t m, int64_t n, int64_t D) {
  X3GradLayout L{};
  const int64_t blk = x3_grad_block(m);
  L.ncp = n < blk ? (n + 255) / 256 * 256 : blk;
  size_t o = 256;
  L.a3 = o; o += up256(static_cast<size_t>(m) * 3 * D * 2);
  L.b3 = o; o += up256(static_cast<size_t>(n) * 3 * D * 2);
  L.ds3 = o; o += up256(static_cast<size_t>(m) * 3 * L.ncp * 2);
  L.bt3 = o; o += up256(static_cast<size_t>(D) * 3 * L.ncp * 2);
  L.total = o;
  return L;
}
}  // namespace

extern "C" int dalm_sim_grad_bf16x3_supported(int64_t m, int64_t n, int64_t D) {
  if (!dalm_sim_rowstats_bf16x3_supported(m, n, D)) return 0;
  const int64_t blk = x3_grad_block(m);
  const int64_t ncp = n < blk ? (n + 255) / 256 * 256 : blk;
  if (static_cast<uint64_t>(m + 256) * 3 * ncp * 2 >= 0xffffff00ull) return 0;       // 32-bit buffer offsets of the dS image
  if (static_cast<uint64_t>(D + 256) * 3 * ncp * 2 >= 0xffffff00ull) return 0;
  return 1;
}

extern "C" size_t dalm_sim_grad_bf16x3_workspace_bytes(int64_t m, int64_t n, int64_t D) {
  if (!dalm_sim_grad_bf16x3_supported(m, n, D)) return 0;
  return x3_grad_layout(m, n, D).total;
}

extern "C" int dalm_sim_grad_bf16x3(const float* A, const float* Bm, int64_t m, int64_t n, int64_t D, float scale,
                                    int64_t diag_offset, const float* row_coef, const float* row_lse, const float* col_coef,
                                    const float* col_lse, float* dA, void* ws, size_t ws_bytes, dalm_stream_t stream) {
  DALM_REQUIRE(A && Bm && row_coef && row_lse && col_coef && col_lse && dA && ws, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dalm_sim_grad_bf16x3_supported(m, n, D), DALM_E_SHAPE,
               "bf16x3 similarity backward needs D % 64 == 0 and operand images below 4 GB (dalm_sim_grad_bf16x3_supported)");
  DALM_REQUIRE(diag_offset >= 0 && diag_offset + m <= n, DALM_E_SHAPE, "diag_offset + m must be <= n");
  DALM_REQUIRE(reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(Bm) % 16 == 0, DALM_E_ALIGN,
               "A / B must be 16-byte aligned");
  const X3GradLayout L = x3_grad_layout(m, n, D);
  DALM_REQUIRE(ws_bytes >= L.total, DALM_E_WORKSPACE, "workspace too small");
  hipStream_t s = as_stream(stream);
  unsigned char* base = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(ws) + 255) / 256 * 256) - 256;
  auto* a3 = reinterpret_cast<unsigned short*>(base + L.a3);
  auto* b3 = reinterpret_cast<unsigned short*>(base + L.b3);
  auto* ds3 = reinterpret_cast<unsigned short*>(base + L.ds3);
  auto* bt3 = reinterpret_cast<unsigned short*>(base + L.bt3);
  const int per_row = static_cast<int>(D / 8);
  hipLaunchKernelGGL(split3_bf16_kernel, dim3(static_cast<unsigned>((m * per_row + 255) / 256)), dim3(256), 0, s, A,
                     static_cast<int>(m), static_cast<int>(D), scale, a3, static_cast<int64_t*>(nullptr), static_cast<int64_t>(0));
  hipLaunchKernelGGL(split3_bf16_kernel, dim3(static_cast<unsigned>((n * per_row + 255) / 256)), dim3(256), 0, s, Bm,
                     static_cast<int>(n), static_cast<int>(D), 1.0f, b3, static_cast<int64_t*>(nullptr), static_cast<int64_t>(0));
  for (int64_t j0 = 0; j0 < n; j0 += L.ncp) {
    const int64_t nc = n - j0 < L.ncp ? n - j0 : L.ncp;
    {  // (1) S tile -> dS thirds
      Lm8Params q;
      q.H = a3; q.W = b3 + j0 * 3 * D; q.labels = nullptr; q.K = static_cast<int>(6 * D);
      lm8_geometry(q, m, nc);
      q.pitchH = q.pitchW = static_cast<unsigned>(3 * D * 2);
      q.bytesH = static_cast<unsigned>(static_cast<uint64_t>(m) * 3 * D * 2);
      q.bytesW = static_cast<unsigned>(static_cast<uint64_t>(nc) * 3 * D * 2);
      q.tpd = static_cast<int>(D / 64);
      q.tpd_inv = static_cast<unsigned>(((1u << 24) + q.tpd - 1) / q.tpd);
      q.seg_bytes = static_cast<unsigned>(D * 2);
      q.row_lse = row_lse; q.coef = row_coef; q.col_lse = col_lse; q.col_coef = col_coef;
      q.col_base = static_cast<int>(j0); q.diag_offset = diag_offset;
      q.out = ds3; q.out_pitch = 3 * L.ncp; q.out_third = L.ncp; q.out_cols = static_cast<int>(L.ncp); q.accumulate = 0;
      hipLaunchKernelGGL((lm_head_lse4w_kernel<8, 8, 0, 0, 0, true, false, EPI_DS3>), dim3(static_cast<unsigned>(q.MT) * q.NT),
                         dim3(256), 0, s, q);
    }
    hipLaunchKernelGGL(split3_transpose_kernel, dim3(static_cast<unsigned>(L.ncp / 64), static_cast<unsigned>((D + 63) / 64)),
                       dim3(256), 0, s, Bm, j0, static_cast<int>(nc), static_cast<int>(D), scale, bt3, static_cast<int>(L.ncp));
    {  // (3) dA (+)= dS_blk . B_blk
      Lm8Params q;
      q.H = ds3; q.W = bt3; q.labels = nullptr; q.K = static_cast<int>(6 * L.ncp);
      lm8_geometry(q, m, D);
      q.pitchH = q.pitchW = static_cast<unsigned>(3 * L.ncp * 2);
      q.bytesH = static_cast<unsigned>(static_cast<uint64_t>(m) * 3 * L.ncp * 2);
      q.bytesW = static_cast<unsigned>(static_cast<uint64_t>(D) * 3 * L.ncp * 2);
      q.tpd = static_cast<int>(L.ncp / 64);
      q.tpd_inv = static_cast<unsigned>(((1u << 24) + q.tpd - 1) / q.tpd);
      q.seg_bytes = static_cast<unsigned>(L.ncp * 2);
      q.row_lse = nullptr; q.coef = nullptr; q.col_base = 0;
      q.out = dA; q.out_pitch = D; q.out_cols = static_cast<int>(D); q.accumulate = j0 > 0;
      hipLaunchKernelGGL((lm_head_lse4w_kernel<8, 8, 0, 0, 0, true, false, EPI_CSTORE>), dim3(static_cast<unsigned>(q.MT) * q.NT),
                         dim3(256), 0, s, q);
    }
  }
  return check_launch(__func__);
}
