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
// Round 3: the same contraction as a 256 x 256 x 64 tile per 8-wave workgroup on the direct-to-LDS path
// (buffer_load_dwordx4 ... lds), two 64 KB LDS buffers, FOUR PHASES per K tile with the load queue never drained.
//
// Workgroup = 8 waves as 2 (rows) x 4 (vocabulary): a wave owns 128 x 64 outputs = 8 accumulator tiles of 32 x 32
// (128 VGPRs), walked as four quadrants of 64 rows x 32 columns; a phase = one quadrant x K = 64 = 8 MFMAs (256 cycles
// of the SIMD's matrix pipe).  One block per CU, two waves per SIMD: the waves of the second row half run ONE BARRIER
// behind the first half, so that on every SIMD one wave multiplies while its partner reads fragments / issues loads.
//
// A K tile of an operand is kept as two 16 KB half-tiles of 128 rows x 64 k: half 0 holds the FIRST sub-block of every
// wave (rows wr*128 + 0..63 / columns wc*64 + 0..31), half 1 the second - so a half-tile is read in exactly one phase
// (A.h0, B.h0: phase 1; B.h1: phase 2; A.h1: phase 3; the b0 fragments stay in registers for phase 4) and its LDS region
// can be refilled two phases later while the rest of the tile is still in use:
//     phase 1 (tile t): refill B.h1 of tile t+1      phase 3: refill A.h0 of tile t+2
//     phase 2         : refill A.h1 of tile t+1      phase 4: refill B.h0 of tile t+2
// Every half-tile is issued >= 5 phases before its first read; each phase ends its issue with s_waitcnt vmcnt(8): the four
// most recent half-tiles (2 loads per thread each) stay in flight ACROSS the barriers, everything older has landed, and a
// half-tile is first read one phase after the wait that covers it (the other waves' wait is ordered by the barrier between).
//
// LDS image of a half-tile: row-major 128-byte rows, 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7): the 16
// lanes a ds_read_b128 services together (16 different rows, one k-chunk) then hit 16 different 16-byte slots of the 256-byte
// bank row.  The LDS-DMA writes lane-linear, so the permutation is applied to the per-lane SOURCE offset (within one 128-byte
// line: still full-line requests) and again on the read.  Rows / vocabulary entries past the end read as zero through the
// buffer descriptor's range check (no clamping, no branches); zero-padded vocabulary columns are masked in the epilogue.
struct Lm8Params {
  const void* H;             // [R, K] bf16
  const void* W;             // [V, K] bf16
  const int64_t* labels;     // [R]
  int R, V, K, MT, NT, xcd_order;
  int gh;                    // row tiles per group of the tile order (see lm_tile_of)
  unsigned bytesH, bytesW;
  float* pm;                 // [4*NT, R]
  float* pl;                 // [4*NT, R]
  float* z;                  // [R]
};

constexpr int L8_HALF = 16384, L8_BUF = 65536;

// Tile order.  Workgroup L runs on XCD L % 8 (own 4 MB L2): every XCD gets a contiguous run of the ORDERED tile list, and
// the list walks the tiles in bands of gh row tiles - within a band vocabulary tile by vocabulary tile, the band's row tiles
// innermost.  The 32 tiles an XCD runs at a time are then ~gh row panels x 32/gh vocabulary panels (7 + 4.6 instead of
// 14 + 2.3 panels of 2 MB for 32 tiles at 14 row tiles): every panel a CU pulls into the L2 is shared by 4-7 neighbours.
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

// Reductions over the 16 lanes of a DPP row, all lanes receive the result: xor 1, xor 2 (quad_perm), then the two mirrors.
// Pure VALU - the first version of these epilogues reduced over 32 lanes with __shfl_xor = 10 ds_bpermute round trips per
// row, each waited for: 640 per lane and tile, ~30 % of the kernel's time.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  return fmaxf(v, dpp_f32<0x140>(v));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  return v + dpp_f32<0x140>(v);
}

// Four independent 16-lane butterflies at once, written out: (1) fmaxf() lowers to a canonicalising v_max x, x in front of
// every v_max and keeps the DPP move separate (3 instructions + s_nop 1 per step); (2) a VALU result needs two wait states
// before a DPP instruction may read it - with four chains interleaved the next step of a chain is 3 instructions behind its
// producer and no s_nop is needed.  v_max_f32 returns the other operand for a NaN, as fmaxf does.
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
// max of up to four finite-or--inf values without the canonicalising self-max fmaxf() puts in front of every operand
__device__ __forceinline__ float max4_raw(float a, float b, float c, float d) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
  return r;
}
__device__ __forceinline__ float max2_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// one direct-to-LDS piece: 64 lanes x 16 bytes -> dst + 16 * lane (dst wave-uniform); its own function so that the host pass,
// which has no such builtin, never sees it inside a kernel body
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* dst, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)dst, 16, static_cast<int>(voff), soff, 0, 0);
}

// ABL: measurement-only ablations (results are garbage unless 0): 1 = no loads in the loop, 2 = no fragment reads in the loop,
// 4 = no stagger between the row halves, 8 = no s_setprio, 16 = no MFMA
template <int ABL>
__global__ __launch_bounds__(512, 2) void lm_head_lse8_kernel(const Lm8Params p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * L8_BUF];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  unsigned wgid = blockIdx.x;
  if (p.xcd_order) {
    const unsigned nwg = gridDim.x, L = blockIdx.x;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7u, xcd = L & 7u;
    wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (L >> 3);
  }
  const int mt = static_cast<int>(wgid) % p.MT, nt = static_cast<int>(wgid) / p.MT;
  const int r0 = mt * 256, c0 = nt * 256;

  const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.H), 0, static_cast<int>(p.bytesH), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, static_cast<int>(p.bytesW), 0x00020000);

  // ---- staging geometry: LDS position (round j, thread tid) -> source row / chunk ----
  // voff[X][j]: byte offset of this lane's 16 bytes of half-tile X (0: A.h0, 1: A.h1, 2: B.h0, 3: B.h1), k = 0
  unsigned voff[4][2];
  const unsigned rowbytes = static_cast<unsigned>(p.K) * 2u;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int lin = j * 512 + tid, lr = lin >> 3, pc = lin & 7;
    const int c = pc ^ ((lr >> 1) & 7);
#pragma unroll
    for (int sblk = 0; sblk < 2; ++sblk) {
      const unsigned ra = static_cast<unsigned>(r0 + (lr >> 6) * 128 + sblk * 64 + (lr & 63));
      const unsigned cb = static_cast<unsigned>(c0 + (lr >> 5) * 64 + sblk * 32 + (lr & 31));
      voff[sblk][j] = ra * rowbytes + c * 16;
      voff[2 + sblk][j] = cb * rowbytes + c * 16;
    }
  }
  const int wave_lds = wave * 1024;
  auto stage = [&](auto X, auto BUFc, int ksoff, bool in_loop = true) {
    constexpr int XX = decltype(X)::value, BB = decltype(BUFc)::value;
    if ((ABL & 1) && in_loop) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      lds_void_t* dst = (lds_void_t*)(lds + BB * L8_BUF + XX * L8_HALF + j * 8192 + wave_lds);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(XX < 2 ? rsH : rsW, dst, 16, static_cast<int>(voff[XX][j]), ksoff, 0, 0);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // ---- fragment read geometry ----
  const int f = (l31 >> 1) & 7;
  int koff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) koff[kk] = ((kk * 2 + lhi) ^ f) * 16;
  const int arow = (wr * 64 + l31) * 128;     // + rb * 4096
  const int brow = (wc * 32 + l31) * 128;

  f32x16 acc[2][2][2];                        // [row sub-block][32-row block][column sub-block]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;

  const int nkt = p.K / 64;
  auto ks = [&](int u) { return min(u, nkt - 1) * 128; };   // byte offset of K tile u (tail issues re-read the last tile)

  // ---- prologue: tile 0 complete + the phase-1 halves of tile 1 ----
  stage(I0{}, I0{}, ks(0), false);
  stage(I2{}, I0{}, ks(0), false);
  stage(I3{}, I0{}, ks(0), false);
  stage(I1{}, I0{}, ks(0), false);
  stage(I0{}, I1{}, ks(1), false);
  stage(I2{}, I1{}, ks(1), false);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (!(ABL & 4) && wr == 1) __builtin_amdgcn_s_barrier();    // second row half: one barrier behind

  bf16x8 a[2][4], b0[4], b1[4];
  if (ABL & 2) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) a[rb][kk] = *reinterpret_cast<const bf16x8*>(lds + rb * 4096 + arow + koff[kk]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      b0[kk] = *reinterpret_cast<const bf16x8*>(lds + 2 * L8_HALF + brow + koff[kk]);
      b1[kk] = *reinterpret_cast<const bf16x8*>(lds + 2 * L8_HALF + brow + koff[kk] + 64);
    }
  }

#define L8_READ_A(BUF, SBLK)                                                                                         \
  if (!(ABL & 2)) _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)    \
      a[rb][kk] = *reinterpret_cast<const bf16x8*>(lds + (BUF) * L8_BUF + (SBLK) * L8_HALF + rb * 4096 + arow + koff[kk]);
#define L8_READ_B(BUF, SBLK, DST)                                                                                    \
  if (!(ABL & 2)) _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                   \
      DST[kk] = *reinterpret_cast<const bf16x8*>(lds + (BUF) * L8_BUF + (2 + (SBLK)) * L8_HALF + brow + koff[kk]);
#define L8_SYNC_COMPUTE(SA, BFRAG, CB)                                                                               \
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                                   \
  __builtin_amdgcn_s_barrier();                                                                                      \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                                                 \
  if (!(ABL & 8)) __builtin_amdgcn_s_setprio(1);                                                                     \
  if (!(ABL & 16)) _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)   \
      acc[SA][rb][CB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rb][kk], BFRAG[kk], acc[SA][rb][CB], 0, 0, 0);     \
  if (!(ABL & 8)) __builtin_amdgcn_s_setprio(0);                                                                     \
  __builtin_amdgcn_sched_barrier(0);                                                                                 \
  __builtin_amdgcn_s_barrier();

  // one K tile = four phases; BUF is the LDS buffer of tile t, OTH the other one
#define L8_TILE(BUF, OTH, t)                                                                                         \
  {                                                                                                                  \
    L8_READ_B(BUF, 0, b0) __builtin_amdgcn_sched_barrier(0); L8_READ_A(BUF, 0)                                        \
    stage(I3{}, std::integral_constant<int, OTH>{}, ks((t) + 1));                                                    \
    L8_SYNC_COMPUTE(0, b0, 0)                                                                                        \
    L8_READ_B(BUF, 1, b1)                                                                                            \
    stage(I1{}, std::integral_constant<int, OTH>{}, ks((t) + 1));                                                    \
    L8_SYNC_COMPUTE(0, b1, 1)                                                                                        \
    L8_READ_A(BUF, 1)                                                                                                \
    stage(I0{}, std::integral_constant<int, BUF>{}, ks((t) + 2));                                                    \
    L8_SYNC_COMPUTE(1, b1, 1)                                                                                        \
    stage(I2{}, std::integral_constant<int, BUF>{}, ks((t) + 2));                                                    \
    L8_SYNC_COMPUTE(1, b0, 0)                                                                                        \
  }

  int t = 0;
  for (; t + 1 < nkt; t += 2) {
    L8_TILE(0, 1, t)
    L8_TILE(1, 0, t + 1)
  }
  if (t < nkt) L8_TILE(0, 1, t)
#undef L8_TILE
#undef L8_SYNC_COMPUTE
#undef L8_READ_A
#undef L8_READ_B
  if (!(ABL & 4) && wr == 0) __builtin_amdgcn_s_barrier();    // the first half waits for the staggered one
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tail re-reads still target the LDS about to be reused
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: labels through the (now free) LDS, the tile reduced in registers ----
  int* lab_s = reinterpret_cast<int*>(lds);
  if (tid < 256) {
    const int r = r0 + tid;
    int y = -1;
    if (r < p.R) {
      const int64_t yl = p.labels[r] - c0;               // rows past R and labels outside this tile's columns: no match
      y = (yl >= 0 && yl < 256 && yl + c0 < p.V) ? static_cast<int>(yl) : -1;
    }
    lab_s[tid] = y;
  }
  __syncthreads();
  const int colA = c0 + wc * 64 + l31, colB = colA + 32;
  const bool okA = colA < p.V, okB = colB < p.V;
  const int64_t prow = static_cast<int64_t>(nt * 4 + wc) * p.R;
#pragma unroll
  for (int sa = 0; sa < 2; ++sa) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = wr * 128 + sa * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const int row = r0 + rl;
        const float x0 = okA ? acc[sa][rb][0][r] : -INFINITY;
        const float x1 = okB ? acc[sa][rb][1][r] : -INFINITY;
        float m = fmaxf(x0, x1);
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        const float mref = (m == -INFINITY) ? 0.f : m;
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
}

// Epilogue of the kernels whose waves own 128 rows x 32 NJ columns (2 x 256/(32 NJ) waves): labels through the (free) LDS, the
// tile reduced in registers.
template <int ABL, int NJ>
__device__ __forceinline__ void lm_tile_epilogue(const Lm8Params& p, f32x16 (&acc)[4][NJ], unsigned char* lds, int r0, int c0,
                                                 int nt, int wr, int wc, int tid) {
  constexpr int WN = 8 / NJ;            // waves along the vocabulary
  const int lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  if (ABL & 64) {   // measurement only: keep the accumulators alive, skip the reduction
    float t0 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t0 += acc[i][j][r];
    if (t0 == 12345.678f) p.z[0] = t0;
    return;
  }
  int* lab_s = reinterpret_cast<int*>(lds);
  if (tid < 256) {
    const int r = r0 + tid;
    int y = -1;
    if (r < p.R) {
      const int64_t yl = p.labels[r] - c0;               // rows past R and labels outside this tile's columns: no match
      y = (yl >= 0 && yl < 256 && yl + c0 < p.V) ? static_cast<int>(yl) : -1;
    }
    lab_s[tid] = y;
  }
  __syncthreads();
  // A 16-lane DPP row holds 16 of the 32 columns of every 32 x 32 tile: it reduces ITS 16 NJ columns of a row to one
  // (max, sum exp) partial - 2 partials per row and wave, no exchange between the two 16-lane rows of a half-wave.
  // After the butterflies every lane of the row holds the result, lane k keeps the one of accumulator register k and
  // the 16 rows of a 32 x 32 tile leave in ONE store instruction.
  const int colw = wc * (32 * NJ) + l31;                 // column of accumulator tile j = colw + 32 j, relative to c0
  const int l15 = lane & 15;
  const int64_t prow = static_cast<int64_t>((nt * WN + wc) * 2 + (l31 >> 4)) * p.R;
  float pen[NJ];            // 0, or -inf for the zero-filled columns >= V of the last vocabulary tile (x + 0 is exact)
#pragma unroll
  for (int j = 0; j < NJ; ++j) pen[j] = (c0 + colw + 32 * j < p.V) ? 0.f : -INFINITY;
  {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float keep_m = 0.f, keep_s = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rl4 = wr * 128 + i * 32 + 8 * g + 4 * lhi;     // rows rl4 .. rl4 + 3 = accumulator registers 4 g .. 4 g + 3
        const int4 lab4 = *reinterpret_cast<const int4*>(lab_s + rl4);
        float x[4][4], m4[4], s4[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
          for (int j = 0; j < 4; ++j) x[rr][j] = -INFINITY;
#pragma unroll
          for (int j = 0; j < NJ; ++j) x[rr][j] = acc[i][j][4 * g + rr] + pen[j];
          m4[rr] = NJ == 4 ? max4_raw(x[rr][0], x[rr][1], x[rr][2], x[rr][3]) : max2_raw(x[rr][0], x[rr][1]);
        }
        row16_max4(m4[0], m4[1], m4[2], m4[3]);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float mneg = (m4[rr] == -INFINITY) ? 0.f : -m4[rr] * kLog2e;   // a strip entirely beyond V: (max -inf, sum 0)
          float sx = 0.f;
#pragma unroll
          for (int j = 0; j < NJ; ++j) sx += __builtin_amdgcn_exp2f(__builtin_fmaf(x[rr][j], kLog2e, mneg));
          s4[rr] = sx;
        }
        row16_sum4(s4[0], s4[1], s4[2], s4[3]);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int r = 4 * g + rr;
          keep_m = (l15 == r) ? m4[rr] : keep_m;
          keep_s = (l15 == r) ? s4[rr] : keep_s;
          // lab_s holds label - c0 when the label falls into this tile's 256 columns, else -1 (no lane matches)
          const int dy = (rr == 0 ? lab4.x : (rr == 1 ? lab4.y : (rr == 2 ? lab4.z : lab4.w))) - colw;
          if ((dy & ~((NJ - 1) * 32)) == 0) {                 // dy in {0, 32, .., 32 (NJ-1)}; select chain: no run-time index into x[]
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
}

// ------------------------------------------------------------------------------------------------------------------
// Round 3, second structure: ONE wave per SIMD.  What the 8-wave kernel above showed on the MI355X (tools/lm_head_ablate.py):
// with two barriers per 8 MFMAs even the MFMA-and-barriers-only skeleton stops at ~1.2 PF/s (an s_barrier hand-off costs
// ~150 cycles and nothing multiplies meanwhile), so the sync count per MFMA has to come down by an order of magnitude.
//
// Workgroup = 4 waves as 2 x 2, the whole 512-register file per wave: a wave owns 128 x 128 outputs = 16 accumulator tiles
// of 32 x 32 (256 accumulation registers).  A K tile (64 deep) is walked in four k-steps of 16 MFMAs (512 matrix-pipe cycles);
// the 8 fragments of the NEXT k-step are read from LDS and 4-6 direct-to-LDS load pieces of a later K tile are issued in the
// shadow of the current k-step's MFMAs (explicit sched_group_barrier interleave).  ONE barrier per K tile, placed where the
// wave has already pulled the tile's last fragments into registers:
//     k-step 0..2 of tile t : multiply | read k-step +1 of tile t            | issue loads of tile t+1 (other LDS buffer)
//     s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier          -> tile t+1 has landed for everybody, tile t's buffer is free
//     k-step 3 of tile t    : multiply | read k-step 0 of tile t+1           | issue loads of tile t+2 (tile t's buffer)
// 0.5 LDS fragment reads per MFMA (0.75 above), half the LDS read traffic per CU.  Same swizzled LDS image, same zero-fill
// through the buffer descriptor, same in-register epilogue as above.
template <int N3, int N0, int N1, int N2, int ABL = 0>
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
  const unsigned rowbytes = static_cast<unsigned>(p.K) * 2u;
  const int srow = tid >> 3;
  const unsigned schunk = static_cast<unsigned>(((tid & 7) ^ ((srow >> 1) & 7)) * 16);
  unsigned voff[16];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    voff[q] = static_cast<unsigned>(r0 + q * 32 + srow) * rowbytes + schunk;
    voff[8 + q] = static_cast<unsigned>(c0 + q * 32 + srow) * rowbytes + schunk;
  }
  const int wave_lds = wave * 1024;
#define L4_DMA(BUF, Q, KSOFF) if (!(ABL & 1)) lds_dma16((Q) < 8 ? rsH : rsW, lds + (BUF) * L8_BUF + (Q) * 4096 + wave_lds, voff[Q], KSOFF);

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
  auto ks = [&](int u) { return min(u, nkt - 1) * 128; };
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
#define L4_PIECES(BUF, FIRST, COUNT, KSOFF)                                                                           \
  _Pragma("unroll") for (int q = 0; q < 16; ++q) if (q >= (FIRST) && q < (FIRST) + (COUNT)) { L4_DMA(BUF, q, KSOFF) }

  // one K tile in buffer BUF (OTH = the other buffer); fragments of its k-step 0 are already in fa0 / fb0
#define L4_TILE(BUF, OTH, t)                                                                                          \
  {                                                                                                                   \
    const int ks1 = ks((t) + 1), ks2 = ks((t) + 2);                                                                   \
    L4_READ(BUF, 1, fa1, fb1) L4_PIECES(OTH, N3, N0, ks1) L4_MFMA(fa0, fb0) L4_SCHED(N0)                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    L4_READ(BUF, 2, fa0, fb0) L4_PIECES(OTH, N3 + N0, N1, ks1) L4_MFMA(fa1, fb1) L4_SCHED(N1)                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    L4_READ(BUF, 3, fa1, fb1) L4_PIECES(OTH, N3 + N0 + N1, N2, ks1) L4_MFMA(fa0, fb0) L4_SCHED(N2)                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                       \
    if (!(ABL & 32)) __builtin_amdgcn_s_barrier();                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    L4_READ(OTH, 0, fa0, fb0) L4_PIECES(BUF, 0, N3, ks2) L4_MFMA(fa1, fb1) L4_SCHED(N3)                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
  }

  // ---- prologue: tile 0 complete, then the first pieces of tile 1 ----
  _Pragma("unroll") for (int q = 0; q < 16; ++q) lds_dma16(q < 8 ? rsH : rsW, lds + q * 4096 + wave_lds, voff[q], ks(0));
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
  L4_PIECES(1, 0, N3, ks(1))
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

  lm_tile_epilogue<ABL, 4>(p, acc, lds, r0, c0, nt, wr, wc, tid);
}

// ------------------------------------------------------------------------------------------------------------------
// Round 3, third structure: the 4-wave kernel with a RING of four 32-deep LDS stages (4 x 32 KB) instead of two 64-deep
// buffers.  Measured on the kernel above: the fragment reads and the barrier cost ~5 %, the loads 25 % - the wave stalled in
// s_waitcnt vmcnt(0) once per K tile, because a tile's last load pieces had only ~1000 cycles to arrive (L2 hit rate 50 %,
// the rest comes from the Infinity Cache / HBM).  Here a stage is issued THREE stages (~3000 matrix-pipe cycles) before it is
// read, and the wait is counted: s_waitcnt vmcnt(16) leaves the two younger stages in flight.
//     k-step 0 of stage s : multiply | read k-step 1 of stage s   | issue the second half of stage s+3
//     s_waitcnt vmcnt(16) lgkmcnt(0); s_barrier      -> stage s+1 has landed for everybody, the slot of stage s is free
//     k-step 1 of stage s : multiply | read k-step 0 of stage s+1 | issue the first half of stage s+4 (slot of stage s)
// LDS image of a stage: [256 rows][64 bytes] per operand, 16-byte chunk c of row r at chunk c ^ ((r >> 2) & 3): a
// ds_read_b128's 16-lane group (16 rows, one chunk) then covers all 16 slots of the 256-byte bank row.
constexpr int LR_STAGE = 32768, LR_OPER = 16384;

// WN = waves along the vocabulary: 2 -> 4 waves of 128 x 128 (one per SIMD, 512 registers each); 4 -> 8 waves of 128 x 64,
// TWO per SIMD, free-running between the one barrier per stage: while one wave of a SIMD sits in the issue of a load piece
// (~50 cycles each, the cost the one-wave-per-SIMD form cannot hide) its partner keeps the matrix pipe busy.
template <int WN, int NA, int ABL>
__global__ __launch_bounds__(128 * WN, WN / 2) void lm_head_ring_kernel(const Lm8Params p) {
  constexpr int NT_ = 128 * WN;                 // threads
  constexpr int NJ = 8 / WN;                    // 32-column MFMA tiles per wave
  constexpr int NP = 32768 / (NT_ * 16);        // load pieces per stage and wave (half of them A, half B)
  constexpr int RPP = NT_ / 4;                  // rows of a stage one piece-round covers
  static_assert(NA >= 0 && NA <= NP, "load pieces per stage and wave");
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * LR_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  unsigned wgid = blockIdx.x;
  if (p.xcd_order) {
    const unsigned nwg = gridDim.x, L = blockIdx.x;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7u, xcd = L & 7u;
    wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (L >> 3);
  }
  const int mt = static_cast<int>(wgid) % p.MT, nt = static_cast<int>(wgid) / p.MT;
  const int r0 = mt * 256, c0 = nt * 256;
  const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.H), 0, static_cast<int>(p.bytesH), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, static_cast<int>(p.bytesW), 0x00020000);

  // load piece q (first half: rows RPP q .. of the A stage, second half: of the B stage): thread -> row RPP q + tid / 4, chunk tid % 4
  const unsigned rowbytes = static_cast<unsigned>(p.K) * 2u;
  const int srow = tid >> 2;
  const unsigned schunk = static_cast<unsigned>(((tid & 3) ^ ((srow >> 2) & 3)) * 16);
  unsigned voff[NP];
#pragma unroll
  for (int q = 0; q < NP / 2; ++q) {
    const int rl0 = p.xcd_order == 2 ? 0 : r0, cl0 = p.xcd_order == 2 ? 0 : c0;   // 2: measurement only, every tile loads tile (0, 0)
    voff[q] = static_cast<unsigned>(rl0 + q * RPP + srow) * rowbytes + schunk;
    voff[NP / 2 + q] = static_cast<unsigned>(cl0 + q * RPP + srow) * rowbytes + schunk;
  }
  const int wave_lds = wave * 1024;
#define LR_DMA(SLOT, Q, KSOFF) if (!(ABL & 1)) lds_dma16((Q) < NP / 2 ? rsH : rsW, lds + (SLOT) * LR_STAGE + (Q) * (NT_ * 16) + wave_lds, voff[Q], KSOFF);

  const int f = (l31 >> 2) & 3;
  int koff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) koff[kk] = ((kk * 2 + lhi) ^ f) * 16;
  const int abase = (wr * 128 + l31) * 64;
  const int bbase = LR_OPER + (wc * (32 * NJ) + l31) * 64;

  f32x16 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int ns = p.K / 32;
  auto ks = [&](int u) { return min(u, ns - 1) * 64; };      // byte offset of stage u (tail issues re-read the last stage)
  bf16x8 fa0[4], fb0[NJ], fa1[4], fb1[NJ];

#define LR_READ(SLOT, KK, FA, FB)                                                                                     \
  if (!(ABL & 2)) _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
    FA[i] = *reinterpret_cast<const bf16x8*>(lds + (SLOT) * LR_STAGE + i * 2048 + abase + koff[KK]);                  \
    if (i < NJ) FB[i] = *reinterpret_cast<const bf16x8*>(lds + (SLOT) * LR_STAGE + i * 2048 + bbase + koff[KK]);      \
  }
#define LR_MFMA(FA, FB)                                                                                               \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j)                         \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i], FB[j], acc[i][j], 0, 0, 0);
#define LR_SCHED(ND)                                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 4 * NJ; ++i) {                                                                \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                \
    if (i < 4 + NJ) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                \
    if (NJ == 4 ? (i >= 8 && i - 8 < (ND)) : (i >= 4 * NJ - (ND))) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); \
  }
#define LR_PIECES(SLOT, FIRST, COUNT, KSOFF)                                                                          \
  _Pragma("unroll") for (int q = 0; q < NP; ++q) if (q >= (FIRST) && q < (FIRST) + (COUNT)) { LR_DMA(SLOT, q, KSOFF) }

  // stage s in slot SL; fragments of its k-step 0 are in fa0 / fb0.  P3 = slot of stage s+3 (= SL - 1), N1 = slot of s+1
#define LR_STAGE_STEP(SL, s)                                                                                          \
  {                                                                                                                   \
    constexpr int P3 = ((SL) + 3) & 3, N1 = ((SL) + 1) & 3;                                                           \
    const int ks3 = ks((s) + 3), ks4 = ks((s) + 4);                                                                   \
    LR_READ(SL, 1, fa1, fb1) LR_PIECES(P3, NA, NP - NA, ks3) LR_MFMA(fa0, fb0) LR_SCHED(NP - NA)                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    if constexpr (NP == 8) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");                               \
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                                  \
    if (!(ABL & 32)) __builtin_amdgcn_s_barrier();                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    LR_READ(N1, 0, fa0, fb0) LR_PIECES(SL, 0, NA, ks4) LR_MFMA(fa1, fb1) LR_SCHED(NA)                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
  }

  // ---- prologue: stages 0, 1, 2 complete and the first pieces of stage 3 ----
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int q = 0; q < NP; ++q) lds_dma16(q < NP / 2 ? rsH : rsW, lds + u * LR_STAGE + q * (NT_ * 16) + wave_lds, voff[q], ks(u));
  if constexpr (NP == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // stage 0 has landed
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (ABL & 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa1[i] = *reinterpret_cast<const bf16x8*>(lds + i * 2048 + abase + koff[1]);
      fa0[i] = *reinterpret_cast<const bf16x8*>(lds + i * 2048 + abase + koff[0]);
      if (i < NJ) {
        fb1[i] = *reinterpret_cast<const bf16x8*>(lds + i * 2048 + bbase + koff[1]);
        fb0[i] = *reinterpret_cast<const bf16x8*>(lds + i * 2048 + bbase + koff[0]);
      }
    }
  } else
  LR_READ(0, 0, fa0, fb0)
  LR_PIECES(3, 0, NA, ks(3))
  __builtin_amdgcn_sched_barrier(0);

  int s = 0;
  for (; s + 3 < ns; s += 4) {
    LR_STAGE_STEP(0, s)
    LR_STAGE_STEP(1, s + 1)
    LR_STAGE_STEP(2, s + 2)
    LR_STAGE_STEP(3, s + 3)
  }
  if (s < ns) LR_STAGE_STEP(0, s)
  if (s + 1 < ns) LR_STAGE_STEP(1, s + 1)
  if (s + 2 < ns) LR_STAGE_STEP(2, s + 2)
#undef LR_STAGE_STEP
#undef LR_PIECES
#undef LR_SCHED
#undef LR_MFMA
#undef LR_READ
#undef LR_DMA
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // tail re-reads still target the LDS about to be reused
  __builtin_amdgcn_s_barrier();
  lm_tile_epilogue<ABL, NJ>(p, acc, lds, r0, c0, nt, wr, wc, tid);
}

// One workgroup = 16 rows: thread t folds the partials p = t/16, t/16 + 16, .. of row t % 16 (16 consecutive rows per
// load instruction = one 64-byte segment of the [P][R] partial arrays, 8 loads in flight), then the 16 folds of a row are
// combined through LDS in fixed order.  R/16 workgroups: the whole chip takes part (the round-2 form used R/64 = 56).
__global__ __launch_bounds__(256) void lm_head_lse_merge_kernel(const float* __restrict__ pm, const float* __restrict__ pl,
                                                                const float* __restrict__ z,
                                                                const int64_t* __restrict__ labels, int R, int V, int P,
                                                                float* __restrict__ row_lse, float* __restrict__ row_nll) {
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
    row_nll[row] = (y < 0) ? 0.f : (y < V ? lse - z[row] : __builtin_nanf(""));
  }
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
  // round-3 kernel: 256 x 256 tiles, direct-to-LDS loads, 4 phases per K tile (needs 32-bit buffer offsets)
  static const char* gen_env = getenv("DALM_LM_HEAD_GEN");
  const int gen = gen_env ? atoi(gen_env) : 4;
  const uint64_t bytesH = static_cast<uint64_t>(R + 256) * K * 2, bytesW = static_cast<uint64_t>(V + 256) * K * 2;
  if (gen >= 3 && bytesH < 0xffffff00ull && bytesW < 0xffffff00ull) {
    Lm8Params q;
    q.H = hidden; q.W = weight; q.labels = labels;
    q.R = static_cast<int>(R); q.V = static_cast<int>(V); q.K = static_cast<int>(K);
    q.MT = static_cast<int>((R + 255) / 256); q.NT = static_cast<int>((V + 255) / 256);
    static const char* xcd_env8 = getenv("DALM_LM_HEAD_XCD");
    q.xcd_order = xcd_env8 ? atoi(xcd_env8) : 1;
    static const char* gh_env = getenv("DALM_LM_HEAD_GH");
    const int bands = (q.MT + 7) / 8;                                  // bands of <= 8 row tiles, as even as possible
    q.gh = gh_env ? atoi(gh_env) : (q.MT + bands - 1) / bands;
    if (q.gh < 1 || q.gh > q.MT) q.gh = q.MT;
    q.bytesH = static_cast<unsigned>(static_cast<uint64_t>(R) * K * 2);
    q.bytesW = static_cast<unsigned>(static_cast<uint64_t>(V) * K * 2);
    float* f8 = static_cast<float*>(ws);
    const int64_t P8 = 4ll * q.NT;
    q.pm = f8; q.pl = f8 + P8 * R; q.z = f8 + 2 * P8 * R;
    if (gen >= 4) {
      const int64_t P4 = 4ll * q.NT;
      q.pm = f8; q.pl = f8 + P4 * R; q.z = f8 + 2 * P4 * R;
      static const char* var_env = getenv("DALM_LM_HEAD_PIECES");
      const int var = var_env ? atoi(var_env) : 0;
      const dim3 g4(static_cast<unsigned>(q.MT) * q.NT);
      switch (var) {
        case 10: hipLaunchKernelGGL((lm_head_ring_kernel<2, 4, 0>), g4, dim3(256), 0, s, q); break;    // ring of four 32-deep stages
        case 111: hipLaunchKernelGGL((lm_head_ring_kernel<2, 4, 1>), g4, dim3(256), 0, s, q); break;   // ablations
        case 112: hipLaunchKernelGGL((lm_head_ring_kernel<2, 4, 2>), g4, dim3(256), 0, s, q); break;
        case 174: hipLaunchKernelGGL((lm_head_ring_kernel<2, 4, 64>), g4, dim3(256), 0, s, q); break;
        case 20: case 21: case 22: case 211: case 212: case 274: case 275: {                          // 8 waves, two per SIMD
          const int64_t P8w = 8ll * q.NT;
          q.pm = f8; q.pl = f8 + P8w * R; q.z = f8 + 2 * P8w * R;
          if (var == 20) hipLaunchKernelGGL((lm_head_ring_kernel<4, 2, 0>), g4, dim3(512), 0, s, q);
          else if (var == 21) hipLaunchKernelGGL((lm_head_ring_kernel<4, 4, 0>), g4, dim3(512), 0, s, q);
          else if (var == 22) hipLaunchKernelGGL((lm_head_ring_kernel<4, 0, 0>), g4, dim3(512), 0, s, q);
          else if (var == 211) hipLaunchKernelGGL((lm_head_ring_kernel<4, 2, 1>), g4, dim3(512), 0, s, q);
          else if (var == 212) hipLaunchKernelGGL((lm_head_ring_kernel<4, 2, 2>), g4, dim3(512), 0, s, q);
          else if (var == 274) hipLaunchKernelGGL((lm_head_ring_kernel<4, 2, 64>), g4, dim3(512), 0, s, q);
          else hipLaunchKernelGGL((lm_head_ring_kernel<4, 2, 67>), g4, dim3(512), 0, s, q);
          hipLaunchKernelGGL(lm_head_lse_merge_kernel, dim3(static_cast<unsigned>((R + 15) / 16)), dim3(256), 0, s, q.pm, q.pl,
                             q.z, labels, q.R, q.V, static_cast<int>(P8w), row_lse, row_nll);
          return check_launch(__func__);
        }
        case 0: hipLaunchKernelGGL((lm_head_lse4w_kernel<6, 5, 5, 0>), g4, dim3(256), 0, s, q); break;
        case 1: hipLaunchKernelGGL((lm_head_lse4w_kernel<4, 4, 4, 4>), g4, dim3(256), 0, s, q); break;
        case 2: hipLaunchKernelGGL((lm_head_lse4w_kernel<8, 8, 0, 0>), g4, dim3(256), 0, s, q); break;
        case 3: hipLaunchKernelGGL((lm_head_lse4w_kernel<4, 6, 6, 0>), g4, dim3(256), 0, s, q); break;
        case 4: hipLaunchKernelGGL((lm_head_lse4w_kernel<16, 0, 0, 0>), g4, dim3(256), 0, s, q); break;
        case 5: hipLaunchKernelGGL((lm_head_lse4w_kernel<12, 4, 0, 0>), g4, dim3(256), 0, s, q); break;
        case 6: hipLaunchKernelGGL((lm_head_lse4w_kernel<0, 8, 8, 0>), g4, dim3(256), 0, s, q); break;
        case 101: hipLaunchKernelGGL((lm_head_lse4w_kernel<6, 5, 5, 0, 1>), g4, dim3(256), 0, s, q); break;   // ablations
        case 102: hipLaunchKernelGGL((lm_head_lse4w_kernel<6, 5, 5, 0, 2>), g4, dim3(256), 0, s, q); break;
        case 103: hipLaunchKernelGGL((lm_head_lse4w_kernel<6, 5, 5, 0, 3>), g4, dim3(256), 0, s, q); break;
        case 135: hipLaunchKernelGGL((lm_head_lse4w_kernel<6, 5, 5, 0, 35>), g4, dim3(256), 0, s, q); break;
        case 199: hipLaunchKernelGGL((lm_head_lse4w_kernel<6, 5, 5, 0, 99>), g4, dim3(256), 0, s, q); break;
        case 164: hipLaunchKernelGGL((lm_head_lse4w_kernel<6, 5, 5, 0, 64>), g4, dim3(256), 0, s, q); break;
        default: return fail(DALM_E_SHAPE, __func__, "unknown DALM_LM_HEAD_PIECES");
      }
      hipLaunchKernelGGL(lm_head_lse_merge_kernel, dim3(static_cast<unsigned>((R + 15) / 16)), dim3(256), 0, s, q.pm, q.pl,
                         q.z, labels, q.R, q.V, static_cast<int>(P4), row_lse, row_nll);
      return check_launch(__func__);
    }
    static const char* abl_env = getenv("DALM_LM_HEAD_ABL");   // measurement only (tools/lm_head_ablate.py)
    const int abl = abl_env ? atoi(abl_env) : 0;
    const dim3 g8(static_cast<unsigned>(q.MT) * q.NT);
    switch (abl) {
      case 0: hipLaunchKernelGGL(lm_head_lse8_kernel<0>, g8, dim3(512), 0, s, q); break;
      case 1: hipLaunchKernelGGL(lm_head_lse8_kernel<1>, g8, dim3(512), 0, s, q); break;
      case 2: hipLaunchKernelGGL(lm_head_lse8_kernel<2>, g8, dim3(512), 0, s, q); break;
      case 3: hipLaunchKernelGGL(lm_head_lse8_kernel<3>, g8, dim3(512), 0, s, q); break;
      case 4: hipLaunchKernelGGL(lm_head_lse8_kernel<4>, g8, dim3(512), 0, s, q); break;
      case 8: hipLaunchKernelGGL(lm_head_lse8_kernel<8>, g8, dim3(512), 0, s, q); break;
      case 16: hipLaunchKernelGGL(lm_head_lse8_kernel<16>, g8, dim3(512), 0, s, q); break;
      case 18: hipLaunchKernelGGL(lm_head_lse8_kernel<18>, g8, dim3(512), 0, s, q); break;
      case 17: hipLaunchKernelGGL(lm_head_lse8_kernel<17>, g8, dim3(512), 0, s, q); break;
      case 19: hipLaunchKernelGGL(lm_head_lse8_kernel<19>, g8, dim3(512), 0, s, q); break;
      default: return fail(DALM_E_SHAPE, __func__, "unknown DALM_LM_HEAD_ABL");
    }
    hipLaunchKernelGGL(lm_head_lse_merge_kernel, dim3(static_cast<unsigned>((R + 15) / 16)), dim3(256), 0, s, q.pm, q.pl,
                       q.z, labels, q.R, q.V, static_cast<int>(P8), row_lse, row_nll);
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
