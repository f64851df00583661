// K5-K7: marginalised causal-LM cross-entropy for gfx950.
//
// Stands in for compute_marginalized_loss_from_logits / marginalize_log_probs /
// get_nll (dalm/training/utils/train_utils.py:91-138).  The reference makes
// ~6 full [B,Tg-1,V] copies (log_softmax, per-sample cat, stack, gather ...);
// here every vocabulary row is read from HBM exactly once, held in registers
// (V = 32000 f32 -> 62.5 floats/lane in a 512-thread block), reduced with
// wave shuffles + one LDS hop, and - optionally - turned into its gradient and
// written back in the same pass.  HBM-bound by construction:
//   forward  : R*V*el bytes read                       (R = B*(Tg-1) rows)
//   fwd+grad : R*V*el read + R*V*el written            (reference: ~12x that)
//
// Row addressing uses 16-byte aligned windows: a row may start at any element
// offset; the first/last 16-byte slot is masked on load and stored with scalar
// writes, all interior slots are single dwordx4 accesses.
#include <cstdlib>

#include "common.hpp"

namespace dalm {
namespace {

struct bf16_t { unsigned short v; };

template <typename T> struct Elt;
template <> struct Elt<float> {
  static constexpr int VEC = 4;
  __device__ static __forceinline__ void load(const float* p, float (&x)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
  }
  __device__ static __forceinline__ void store(float* p, const float (&x)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
  }
  __device__ static __forceinline__ float get(const float* p) { return *p; }
  __device__ static __forceinline__ void put(float* p, float v) { *p = v; }
  // streaming (non-temporal) forms: every logit is read once / every gradient written once
  __device__ static __forceinline__ void load_nt(const float* p, float (&x)[4]) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
  }
  __device__ static __forceinline__ void store_nt(float* p, const float (&x)[4]) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 v;
    v.x = x[0]; v.y = x[1]; v.z = x[2]; v.w = x[3];
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  }
};
template <> struct Elt<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ void load(const bf16_t* p, float (&x)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[2 * i] = __uint_as_float(w[i] << 16);
      x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float (&x)[8]) {
    unsigned int w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = pack_bf16x2(x[2 * i], x[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  __device__ static __forceinline__ float get(const bf16_t* p) { return bf16_to_f32(p->v); }
  __device__ static __forceinline__ void put(bf16_t* p, float v) { p->v = f32_to_bf16(v); }
  __device__ static __forceinline__ void load_nt(const bf16_t* p, float (&x)[8]) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[2 * i] = __uint_as_float(w[i] << 16);
      x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void store_nt(bf16_t* p, const float (&x)[8]) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    v.x = pack_bf16x2(x[0], x[1]); v.y = pack_bf16x2(x[2], x[3]);
    v.z = pack_bf16x2(x[4], x[5]); v.w = pack_bf16x2(x[6], x[7]);
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
  }
};

// Geometry of one vocabulary row seen through 16-byte aligned slots.
template <typename T>
struct RowWin {
  const T* abase;  // 16-byte aligned address at or before the row start
  int lead;        // elements of the first slot that belong to the previous row
  int nslots;      // slots covering [lead, lead+V)
  int V;
  __device__ __forceinline__ RowWin(const T* row, int V_) : V(V_) {
    constexpr int VEC = Elt<T>::VEC;
    const uintptr_t a = reinterpret_cast<uintptr_t>(row);
    lead = static_cast<int>((a & 15u) / sizeof(T));
    abase = row - lead;
    nslots = (lead + V + VEC - 1) / VEC;
  }
  __device__ __forceinline__ bool partial(int slot) const {
    constexpr int VEC = Elt<T>::VEC;
    return (slot == 0 && lead != 0) || (slot == nslots - 1 && ((lead + V) % VEC) != 0);
  }
};

// Write `fill` over one full row (used for masked rows and the t = Tg-1 slot).
// NTFILL: non-temporal stores (the zero rows of a padded batch are a pure write stream: 110 of the 295 MB written per
// launch at the bench's masks); measured against cached stores in profiles/history/r04_ce_row_order.txt.
template <typename T, int BS, bool NTFILL = false>
__device__ __forceinline__ void fill_row(T* row, int V, float fill) {
  constexpr int VEC = Elt<T>::VEC;
  RowWin<T> w(row, V);
  T* abase = const_cast<T*>(w.abase);
  float z[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) z[e] = fill;
  for (int slot = threadIdx.x; slot < w.nslots; slot += BS) {
    if (!w.partial(slot)) {
      if constexpr (NTFILL) Elt<T>::store_nt(abase + static_cast<int64_t>(slot) * VEC, z);
      else Elt<T>::store(abase + static_cast<int64_t>(slot) * VEC, z);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int idx = slot * VEC + e - w.lead;
        if (idx >= 0 && idx < V) Elt<T>::put(row + idx, fill);
      }
    }
  }
}

// Workgroup -> (b,t) row.  Left-padded samples put a run of write-only rows (padding: the gradient is zeros) in front
// of a run of read+write rows; with row = blockIdx.x the dispatcher issues them in that order.  `order` permutes which
// row a workgroup takes (a bijection on [0, B*Tg)), so that both kinds are co-resident on a CU:
//   0  identity;   -1  sample-interleaved (b = blk % B, t = blk / B);   s > 0  t = (i * s) % Tg inside a sample, gcd(s,Tg)=1
__device__ __forceinline__ int64_t map_row(unsigned blk, int Tg, int order) {
  if (order == 0) return blk;
  if (order < 0) {
    const unsigned B = gridDim.x / static_cast<unsigned>(Tg);
    return static_cast<int64_t>(blk % B) * Tg + blk / B;
  }
  const unsigned b = blk / static_cast<unsigned>(Tg), i = blk % static_cast<unsigned>(Tg);
  return static_cast<int64_t>(b) * Tg + (static_cast<uint64_t>(i) * static_cast<unsigned>(order)) % static_cast<unsigned>(Tg);
}

// ---------------------------------------------------------------------------
// Register-resident row kernel: one block = one (b,t) row.
// ALIGNED: every row starts on a 16-byte boundary and V % VEC == 0 (decided on
// the host) - no edge-slot logic at all; addresses are uniform base + 32-bit
// lane offset so the loads/stores use the saddr form and no address VGPRs.
// ---------------------------------------------------------------------------

template <typename T, int BS, int SLOTS, bool WRITE_GRAD, bool ALIGNED>
__global__ __launch_bounds__(BS, (SLOTS * Elt<T>::VEC > 64 ? 3 : 4)) void marg_ce_row_kernel(
    const T* __restrict__ logits, int64_t stride_b, int64_t stride_t,
    const int64_t* __restrict__ ids, const int64_t* __restrict__ mask, int Tg, int V,
    const float* __restrict__ stats, float* __restrict__ row_lse, float* __restrict__ row_nll,
    T* dlogits, int order) {
  constexpr int VEC = Elt<T>::VEC;
  __shared__ float red[BS / kWave];
  const int64_t row = map_row(blockIdx.x, Tg, order);
  const int b = static_cast<int>(row / Tg), t = static_cast<int>(row % Tg);
  const int tid = threadIdx.x;
  const int64_t off = b * stride_b + t * stride_t;
  const bool last = (t == Tg - 1);
  const int64_t mi = last ? 0 : mask[static_cast<int64_t>(b) * Tg + t + 1];
  const float M = stats[0];

  if (last || mi == 0) {
    if (tid == 0) { row_lse[row] = 0.f; row_nll[row] = 0.f; }
    if constexpr (WRITE_GRAD) {
      // reference: masked rows get 0 * (1/M); with M == 0 that is NaN (0*inf)
      const float fill = (!last && M == 0.f) ? __builtin_nanf("") : 0.f;
      fill_row<T, BS>(dlogits + off, V, fill);
    }
    return;
  }

  const T* xrow = logits + off;
  const int64_t y = ids[static_cast<int64_t>(b) * Tg + t + 1];
  int lead = 0, nslots = V / VEC;
  if constexpr (!ALIGNED) {
    RowWin<T> w(xrow, V);
    lead = w.lead; nslots = w.nslots;
  }
  const bool tail_partial = !ALIGNED && ((lead + V) % VEC) != 0;
  const char* abase = reinterpret_cast<const char*>(xrow - lead);  // wave-uniform

  // ---- single HBM pass: the whole row lands in registers -------------------
  float x[SLOTS][VEC];
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const int slot = k * BS + tid;
    if (slot < nslots) {
      Elt<T>::load_nt(reinterpret_cast<const T*>(abase + static_cast<unsigned>(slot) * 16u), x[k]);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) x[k][e] = -INFINITY;
    }
  }
  // every lane reads the label logit (one broadcast request per wave): lane 0 needs it for the
  // NLL, the lane owning the label's slot for the gradient patch
  const float xy = (y >= 0 && y < V) ? Elt<T>::get(xrow + y) : __builtin_nanf("");

  float tmax = -INFINITY;
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    if constexpr (!ALIGNED) {
      const int slot = k * BS + tid;
      if ((slot == 0 && lead != 0) || (slot == nslots - 1 && tail_partial)) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const int idx = slot * VEC + e - lead;
          if (idx < 0 || idx >= V) x[k][e] = -INFINITY;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) tmax = fmaxf(tmax, x[k][e]);
  }
  const float m = block_max<BS>(tmax, red);
  const float mneg = -m * kLog2e;

  float tsum = 0.f;
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      x[k][e] = __builtin_amdgcn_exp2f(fmaf(x[k][e], kLog2e, mneg));  // exp(x - m); -inf -> 0
      tsum += x[k][e];
    }
  }
  const float l = block_sum<BS>(tsum, red);
  const float lse = m + __logf(l);
  const float mval = static_cast<float>(mi);
  if (tid == 0) {
    row_lse[row] = lse;
    row_nll[row] = mval * (lse - xy);
  }

  if constexpr (WRITE_GRAD) {
    // dL/dlogits = (m/M) (softmax - onehot(y))       [upstream grad = 1]
    // The row is stored as coef*softmax with plain vector stores; the one label entry is then
    // patched by the lane that just stored it (same lane, same address: program order), which
    // keeps ~3 predicated VALU ops per element out of the store loop.
    T* grow = dlogits + off;
    char* gbase = reinterpret_cast<char*>(grow - lead);  // wave-uniform
    const float coef = mval / M;
    const float inv = coef / l;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int slot = k * BS + tid;
      if (slot >= nslots) continue;
#pragma unroll
      for (int e = 0; e < VEC; ++e) x[k][e] *= inv;
      bool part = false;
      if constexpr (!ALIGNED) part = (slot == 0 && lead != 0) || (slot == nslots - 1 && tail_partial);
      if (!part) {
        Elt<T>::store_nt(reinterpret_cast<T*>(gbase + static_cast<unsigned>(slot) * 16u), x[k]);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const int idx = slot * VEC + e - lead;
          if (idx >= 0 && idx < V) Elt<T>::put(grow + idx, x[k][e]);
        }
      }
    }
    if (y >= 0 && y < V) {
      const int slot_y = (static_cast<int>(y) + lead) / VEC;
      if (tid == slot_y % BS) {  // the lane that owns (and has just stored) the label's slot
        const float py = __builtin_amdgcn_exp2f(fmaf(xy, kLog2e, mneg)) * inv;  // same ops as the row pass
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // its vector stores are acknowledged first
        Elt<T>::put(grow + y, py - coef);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// bf16 rows kept PACKED in registers (4 VGPRs per 8 logits): 32 data VGPRs per lane for a 32768-wide
// row in a 512-thread block -> <= 64 VGPRs, 8 waves/SIMD, four rows resident per CU.  With that many
// independent rows per CU the read phase of one row overlaps the write phase of another, which the
// unpacked variant (2 rows/CU, time = T_read + T_write) could not do.  exp() is recomputed in the
// gradient phase instead of being kept in f32 registers (VALU is far from the limit here).
// ---------------------------------------------------------------------------
// NT: cache policy of the streams - bit 0 = non-temporal loads, bit 1 = non-temporal stores, bit 2 = non-temporal zero fill
// of the rows without loss.
template <int BS, int SLOTS, bool WRITE_GRAD, bool ALIGNED, int NT = 0>
__global__ __launch_bounds__(BS, 8) void marg_ce_row_bf16_kernel(
    const bf16_t* __restrict__ logits, int64_t stride_b, int64_t stride_t,
    const int64_t* __restrict__ ids, const int64_t* __restrict__ mask, int Tg, int V,
    const float* __restrict__ stats, float* __restrict__ row_lse, float* __restrict__ row_nll,
    bf16_t* dlogits, int order) {
  using T = bf16_t;
  constexpr int VEC = 8;
  __shared__ float red[BS / kWave];
  const int64_t row = map_row(blockIdx.x, Tg, order);
  const int b = static_cast<int>(row / Tg), t = static_cast<int>(row % Tg);
  const int tid = threadIdx.x;
  const int64_t off = b * stride_b + t * stride_t;
  const bool last = (t == Tg - 1);
  const int64_t mi = last ? 0 : mask[static_cast<int64_t>(b) * Tg + t + 1];
  const float M = stats[0];
  if (last || mi == 0) {
    if (tid == 0) { row_lse[row] = 0.f; row_nll[row] = 0.f; }
    if constexpr (WRITE_GRAD) {
      const float fill = (!last && M == 0.f) ? __builtin_nanf("") : 0.f;
      fill_row<T, BS, (NT & 4) != 0>(dlogits + off, V, fill);
    }
    return;
  }
  const T* xrow = logits + off;
  const int64_t y = ids[static_cast<int64_t>(b) * Tg + t + 1];
  int lead = 0, nslots = V / VEC;
  if constexpr (!ALIGNED) {
    RowWin<T> w(xrow, V);
    lead = w.lead; nslots = w.nslots;
  }
  const bool tail_partial = !ALIGNED && ((lead + V) % VEC) != 0;
  const char* abase = reinterpret_cast<const char*>(xrow - lead);  // wave-uniform

  constexpr unsigned kNegInf2 = 0xff80ff80u;  // two bf16 -inf
  uint4 raw[SLOTS];
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const int slot = k * BS + tid;
    if (slot < nslots) {
      const uint4* src = reinterpret_cast<const uint4*>(abase + static_cast<unsigned>(slot) * 16u);
      if constexpr (NT & 1) {  // streaming policy: every logit is read exactly once
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src));
        raw[k] = make_uint4(t.x, t.y, t.z, t.w);
      } else {
        raw[k] = *src;
      }
    } else {
      raw[k] = make_uint4(kNegInf2, kNegInf2, kNegInf2, kNegInf2);
    }
  }
  const float xy = (y >= 0 && y < V) ? bf16_to_f32(xrow[y].v) : __builtin_nanf("");

  if constexpr (!ALIGNED) {  // mask the elements of edge slots that belong to neighbouring rows
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int slot = k * BS + tid;
      if (slot < nslots && ((slot == 0 && lead != 0) || (slot == nslots - 1 && tail_partial))) {
        unsigned w[4] = {raw[k].x, raw[k].y, raw[k].z, raw[k].w};
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const int idx = slot * VEC + e - lead;
          if (idx < 0 || idx >= V) w[e >> 1] = (e & 1) ? ((w[e >> 1] & 0x0000ffffu) | 0xff800000u)
                                                        : ((w[e >> 1] & 0xffff0000u) | 0x0000ff80u);
        }
        raw[k] = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
  }

  float tmax = -INFINITY;
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const unsigned w[4] = {raw[k].x, raw[k].y, raw[k].z, raw[k].w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      tmax = fmaxf(tmax, fmaxf(__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)));
    __builtin_amdgcn_sched_barrier(0);
  }
  const float m = block_max<BS>(tmax, red);
  const float mneg = -m * kLog2e;
  // make raw[] opaque between the passes: otherwise the unpacked f32 values are CSE'd across the
  // three passes and stay live (64 extra VGPRs), which is exactly what packing is meant to avoid
#define DALM_LAUNDER_RAW()                                                                          \
  _Pragma("unroll") for (int k = 0; k < SLOTS; ++k)                                                 \
      asm volatile("" : "+v"(raw[k].x), "+v"(raw[k].y), "+v"(raw[k].z), "+v"(raw[k].w))
  DALM_LAUNDER_RAW();
  float tsum = 0.f;
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const unsigned w[4] = {raw[k].x, raw[k].y, raw[k].z, raw[k].w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tsum += __builtin_amdgcn_exp2f(fmaf(__uint_as_float(w[i] << 16), kLog2e, mneg));
      tsum += __builtin_amdgcn_exp2f(fmaf(__uint_as_float(w[i] & 0xffff0000u), kLog2e, mneg));
    }
    __builtin_amdgcn_sched_barrier(0);  // one slot at a time: keeps the live set at raw[] + a few temporaries
  }
  const float l = block_sum<BS>(tsum, red);
  const float lse = m + __logf(l);
  const float mval = static_cast<float>(mi);
  if (tid == 0) {
    row_lse[row] = lse;
    row_nll[row] = mval * (lse - xy);
  }
  if constexpr (WRITE_GRAD) {
    DALM_LAUNDER_RAW();
    T* grow = dlogits + off;
    char* gbase = reinterpret_cast<char*>(grow - lead);
    const float coef = mval / M;
    // softmax = exp(x - lse): a different exponent offset than the sum pass on purpose - with the
    // same expression the compiler CSEs the two passes and keeps all 64 exps live (+64 VGPRs)
    const float nlse = -lse * kLog2e;
    const bool y_ok = y >= 0 && y < V;
    const int slot_y = y_ok ? (static_cast<int>(y) + lead) / VEC : -1, e_y = y_ok ? (static_cast<int>(y) + lead) % VEC : 0;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int slot = k * BS + tid;
      if (slot >= nslots) continue;
      const unsigned w[4] = {raw[k].x, raw[k].y, raw[k].z, raw[k].w};
      float g[VEC];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        g[2 * i] = __builtin_amdgcn_exp2f(fmaf(__uint_as_float(w[i] << 16), kLog2e, nlse)) * coef;
        g[2 * i + 1] = __builtin_amdgcn_exp2f(fmaf(__uint_as_float(w[i] & 0xffff0000u), kLog2e, nlse)) * coef;
      }
      if (slot == slot_y) {   // the label's element: softmax - 1 (same bits as a separate store of py - coef would write)
#pragma unroll
        for (int e = 0; e < VEC; ++e) g[e] = (e == e_y) ? g[e] - coef : g[e];
      }
      bool part = false;
      if constexpr (!ALIGNED) part = (slot == 0 && lead != 0) || (slot == nslots - 1 && tail_partial);
      if (!part) {
        if constexpr (NT & 2) {
          typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
          u32x4 o;
          o.x = pack_bf16x2(g[0], g[1]); o.y = pack_bf16x2(g[2], g[3]);
          o.z = pack_bf16x2(g[4], g[5]); o.w = pack_bf16x2(g[6], g[7]);
          __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(gbase + static_cast<unsigned>(slot) * 16u));
        } else {
          Elt<T>::store(reinterpret_cast<T*>(gbase + static_cast<unsigned>(slot) * 16u), g);
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const int idx = slot * VEC + e - lead;
          if (idx >= 0 && idx < V) Elt<T>::put(grow + idx, g[e]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
#undef DALM_LAUNDER_RAW

// ---------------------------------------------------------------------------
// Streaming fallback for rows that do not fit the register file (V > 65k f32):
// online (max,sum) pass, then an L2-served second pass for the gradient.
// ---------------------------------------------------------------------------
template <typename T, int BS, bool WRITE_GRAD>
__global__ __launch_bounds__(BS) void marg_ce_stream_kernel(
    const T* __restrict__ logits, int64_t stride_b, int64_t stride_t,
    const int64_t* __restrict__ ids, const int64_t* __restrict__ mask, int Tg, int V,
    const float* __restrict__ stats, float* __restrict__ row_lse, float* __restrict__ row_nll,
    T* dlogits, int order) {
  constexpr int VEC = Elt<T>::VEC;
  __shared__ float red[BS / kWave];
  const int64_t row = map_row(blockIdx.x, Tg, order);
  const int b = static_cast<int>(row / Tg), t = static_cast<int>(row % Tg);
  const int tid = threadIdx.x;
  const int64_t off = b * stride_b + t * stride_t;
  const bool last = (t == Tg - 1);
  const int64_t mi = last ? 0 : mask[static_cast<int64_t>(b) * Tg + t + 1];
  const float M = stats[0];
  if (last || mi == 0) {
    if (tid == 0) { row_lse[row] = 0.f; row_nll[row] = 0.f; }
    if constexpr (WRITE_GRAD) {
      const float fill = (!last && M == 0.f) ? __builtin_nanf("") : 0.f;
      fill_row<T, BS>(dlogits + off, V, fill);
    }
    return;
  }
  const T* xrow = logits + off;
  const int64_t y = ids[static_cast<int64_t>(b) * Tg + t + 1];
  RowWin<T> w(xrow, V);
  // label logit first: with dlogits aliasing logits (in-place mode) other waves may already be storing the
  // gradient of this row by the time thread 0 gets past the block reductions below
  float xy = 0.f;
  if (tid == 0) xy = (y >= 0 && y < V) ? Elt<T>::get(xrow + y) : __builtin_nanf("");

  float tm = -INFINITY, tl = 0.f;
  for (int slot = tid; slot < w.nslots; slot += BS) {
    float v[VEC];
    Elt<T>::load(w.abase + static_cast<int64_t>(slot) * VEC, v);
    if (w.partial(slot)) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int idx = slot * VEC + e - w.lead;
        if (idx < 0 || idx >= V) v[e] = -INFINITY;
      }
    }
    float vm = v[0];
#pragma unroll
    for (int e = 1; e < VEC; ++e) vm = fmaxf(vm, v[e]);
    const float mn = fmaxf(tm, vm);
    float acc = (tm == -INFINITY) ? 0.f : tl * fast_exp(tm - mn);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc += fast_exp(v[e] - mn);
    tm = mn; tl = acc;
  }
  const float m = block_max<BS>(tm, red);
  const float l = block_sum<BS>((tm == -INFINITY) ? 0.f : tl * fast_exp(tm - m), red);
  const float lse = m + __logf(l);
  const float mval = static_cast<float>(mi);
  if (tid == 0) {
    row_lse[row] = lse;
    row_nll[row] = mval * (lse - xy);
  }
  if constexpr (WRITE_GRAD) {
    T* grow = dlogits + off;
    T* gbase = grow - w.lead;
    const float coef = mval / M;
    // in-place safe only if every thread re-reads exactly the slots it writes: it does.
    for (int slot = tid; slot < w.nslots; slot += BS) {
      float v[VEC];
      Elt<T>::load(w.abase + static_cast<int64_t>(slot) * VEC, v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int idx = slot * VEC + e - w.lead;
        v[e] = coef * fast_exp(v[e] - lse) - ((idx == y) ? coef : 0.f);
      }
      if (!w.partial(slot)) {
        Elt<T>::store(gbase + static_cast<int64_t>(slot) * VEC, v);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const int idx = slot * VEC + e - w.lead;
          if (idx >= 0 && idx < V) Elt<T>::put(grow + idx, v[e]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Stand-alone backward from the saved row_lse (one read + one write of [R,V]).
// ---------------------------------------------------------------------------
template <typename T, int BS>
__global__ __launch_bounds__(BS) void marg_ce_bwd_kernel(
    const T* __restrict__ logits, int64_t stride_b, int64_t stride_t,
    const int64_t* __restrict__ ids, const int64_t* __restrict__ mask, int Tg, int V,
    const float* __restrict__ stats, const float* __restrict__ row_lse,
    const float* __restrict__ gscale, T* dlogits, const float* __restrict__ row_w) {
  constexpr int VEC = Elt<T>::VEC;
  const int64_t row = blockIdx.x;
  const int b = static_cast<int>(row / Tg), t = static_cast<int>(row % Tg);
  const int tid = threadIdx.x;
  const int64_t off = b * stride_b + t * stride_t;
  const bool last = (t == Tg - 1);
  const int64_t mi = last ? 0 : mask[static_cast<int64_t>(b) * Tg + t + 1];
  const float M = stats[0];
  const float g = gscale ? gscale[0] : 1.f;
  if (last || mi == 0) {
    const float fill = (!last && M == 0.f) ? __builtin_nanf("") : 0.f;
    fill_row<T, BS>(dlogits + off, V, fill);
    return;
  }
  const T* xrow = logits + off;
  const int64_t y = ids[static_cast<int64_t>(b) * Tg + t + 1];
  RowWin<T> w(xrow, V);
  T* grow = dlogits + off;
  T* gbase = grow - w.lead;
  // row_w (k retrieved contexts, dalm_marg_ce_finalize_topk): the row's own weight -dL/d(label log-prob) replaces m/M
  const float coef = row_w ? g * row_w[row] : g * static_cast<float>(mi) / M;
  const float nlse = -row_lse[row] * kLog2e;
  for (int slot = tid; slot < w.nslots; slot += BS) {
    float v[VEC];
    Elt<T>::load_nt(w.abase + static_cast<int64_t>(slot) * VEC, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int idx = slot * VEC + e - w.lead;
      v[e] = coef * __builtin_amdgcn_exp2f(fmaf(v[e], kLog2e, nlse)) - ((idx == y) ? coef : 0.f);
    }
    if (!w.partial(slot)) {
      Elt<T>::store_nt(gbase + static_cast<int64_t>(slot) * VEC, v);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int idx = slot * VEC + e - w.lead;
        if (idx >= 0 && idx < V) Elt<T>::put(grow + idx, v[e]);
      }
    }
  }
}

// ---- prep: M = sum mask[:,1:], N_b = sum_t mask[b,t+1] [t >= qlen_b-1] -----
// stage 1: one wave per sample -> Mb[b], Nb[b]; stage 2: one block -> stats[0] = M.
__global__ __launch_bounds__(64) void ce_prep_rows_kernel(const int64_t* __restrict__ mask,
                                                          const int64_t* __restrict__ qlen, int Tg,
                                                          float* __restrict__ Mb, float* __restrict__ Nb) {
  const int b = blockIdx.x;
  // python slice semantics of lp[:qlen-1] / lp[qlen-1:] over the Tg-1 shifted rows
  // (train_utils.py:100-103): a negative start counts from the end.
  int64_t cut = qlen ? qlen[b] - 1 : static_cast<int64_t>(Tg);
  if (cut < 0) cut = (cut + (Tg - 1) < 0) ? 0 : cut + (Tg - 1);
  float ms = 0.f, ns = 0.f;
  for (int t = threadIdx.x; t < Tg - 1; t += 64) {
    const float mv = static_cast<float>(mask[static_cast<int64_t>(b) * Tg + t + 1]);
    ms += mv;
    if (static_cast<int64_t>(t) >= cut) ns += mv;
  }
  ms = wave_sum(ms);
  ns = wave_sum(ns);
  if (threadIdx.x == 0) { Mb[b] = ms; Nb[b] = ns; }
}
__global__ __launch_bounds__(256) void ce_prep_total_kernel(const float* __restrict__ Mb, int B,
                                                            float* __restrict__ stats) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) s += Mb[i];
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) { stats[0] = s; stats[1] = static_cast<float>(B); }
}

// ---- finalize: deterministic reduction to L_gen -----------------------------
__global__ __launch_bounds__(1024) void ce_finalize_kernel(const float* __restrict__ row_nll, int64_t R,
                                                           const float* __restrict__ Nb,
                                                           const float* __restrict__ doc_lp, int B,
                                                           const float* __restrict__ stats,
                                                           float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < R; i += 1024) s += row_nll[i];
  if (doc_lp)
    for (int i = threadIdx.x; i < B; i += 1024) s -= Nb[i] * doc_lp[i];
  s = block_sum<1024>(s, red);
  if (threadIdx.x == 0) out[0] = s / stats[0];
}

// ---- finalize with k retrieved contexts per sample (RAG-token marginalisation; the reference is k = 1) ----
// One workgroup walks the samples in order (B k Tg is a few thousand values): deterministic, no atomics.
//   part A (rows before the cut of sequence (b,c)):  sum_c sum_{t < cut_bc} row_nll[b,c,t] / k
//   part B (answer row j of sample b):                -logsumexp_c( doc_lp[b,c] - row_nll[b,c,cut_bc + j] )
__global__ __launch_bounds__(256) void ce_finalize_topk_kernel(const float* __restrict__ row_nll, int B, int k, int Tg,
                                                               const int64_t* __restrict__ cut,
                                                               const float* __restrict__ Nb,
                                                               const float* __restrict__ doc_lp,
                                                               const float* __restrict__ stats, float* __restrict__ out,
                                                               float* __restrict__ weights) {
  __shared__ float red[4];
  const float M = stats[0], invk = 1.f / static_cast<float>(k);
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    const int nans = static_cast<int>(Nb[b]);
    for (int c = 0; c < k; ++c) {
      const int64_t base = (static_cast<int64_t>(b) * k + c) * Tg;
      const int cu = static_cast<int>(min<int64_t>(max<int64_t>(cut[b * k + c], 0), Tg));
      for (int t = threadIdx.x; t < Tg; t += 256) {
        if (t < cu) s += row_nll[base + t] * invk;
        if (weights) weights[base + t] = (t < cu) ? invk / M : 0.f;
      }
    }
    // the answer rows below overwrite entries the loop above has just zeroed - from OTHER threads (t = cut + j belongs to
    // thread (cut + j) % 256 above and to thread j % 256 below): without this barrier a wave still in the first loop could
    // zero a weight another wave had already written
    if (weights) __syncthreads();
    for (int j = threadIdx.x; j < nans; j += 256) {
      float mx = -INFINITY;
      for (int c = 0; c < k; ++c) {
        const int t = static_cast<int>(cut[b * k + c]) + j;
        const float v = (t >= 0 && t < Tg) ? doc_lp[b * k + c] - row_nll[(static_cast<int64_t>(b) * k + c) * Tg + t] : -INFINITY;
        mx = fmaxf(mx, v);
      }
      if (mx == -INFINITY) continue;   // answer row j lies outside every sequence (inconsistent cut / Nb): no term, no weight
      float se = 0.f;
      for (int c = 0; c < k; ++c) {
        const int t = static_cast<int>(cut[b * k + c]) + j;
        if (t >= 0 && t < Tg) se += __expf(doc_lp[b * k + c] - row_nll[(static_cast<int64_t>(b) * k + c) * Tg + t] - mx);
      }
      s -= mx + __logf(se);
      if (weights)
        for (int c = 0; c < k; ++c) {
          const int t = static_cast<int>(cut[b * k + c]) + j;
          if (t >= 0 && t < Tg) {
            const int64_t at = (static_cast<int64_t>(b) * k + c) * Tg + t;
            weights[at] = __expf(doc_lp[b * k + c] - row_nll[at] - mx) / se / M;
          }
        }
    }
  }
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) out[0] = s / M;
}

template <typename T>
__global__ __launch_bounds__(256) void scale_inplace_kernel(T* x, int64_t n, const float* __restrict__ gscale) {
  constexpr int VEC = Elt<T>::VEC;
  const float g = gscale[0];
  if (g == 1.f) return;  // uniform: the common loss.backward() case costs one tiny launch
  const int64_t nvec = n / VEC;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < nvec; i += gridDim.x * 256ll) {
    float v[VEC];
    Elt<T>::load(x + i * VEC, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] *= g;
    Elt<T>::store(x + i * VEC, v);
  }
  if (blockIdx.x == 0)
    for (int64_t i = nvec * VEC + threadIdx.x; i < n; i += 256) Elt<T>::put(x + i, Elt<T>::get(x + i) * g);
}

__global__ __launch_bounds__(256) void gather_nll_kernel(const float* __restrict__ lp,
                                                         const int64_t* __restrict__ labels, int64_t R,
                                                         int64_t V, float* __restrict__ out) {
  const int64_t r = blockIdx.x * 256ll + threadIdx.x;
  if (r >= R) return;
  const int64_t y = labels[r];
  out[r] = (y >= 0 && y < V) ? -lp[r * V + y] : __builtin_nanf("");
}

__global__ __launch_bounds__(256) void marginalize_rows_kernel(const float* __restrict__ lp, int64_t T,
                                                               int64_t V, const float* __restrict__ doc_lp,
                                                               int64_t qlen, const int64_t* __restrict__ qlen_dev,
                                                               float* __restrict__ out) {
  const int64_t n = T * V;
  const float d = doc_lp[0];
  if (qlen_dev) qlen = qlen_dev[0];   // the length stays on the device: no host round trip per sample
  // python slice semantics of lp[:qlen-1] / lp[qlen-1:] (train_utils.py:100-103)
  int64_t cut = qlen - 1;
  if (cut < 0) cut = (cut + T < 0) ? 0 : cut + T;
  if (cut > T) cut = T;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
    const int64_t t = i / V;
    out[i] = lp[i] + ((t >= cut) ? d : 0.f);
  }
}

constexpr int kDefaultRowOrder = 0;     // decided by measurement, see profiles/history/r04_ce_row_order.txt
constexpr bool kDefaultNtFill = false;

template <typename T, bool GRAD, bool ALIGNED>
void launch_fwd2(const T* logits, int64_t B, int64_t Tg, int64_t V, int64_t sb, int64_t st,
                 const int64_t* ids, const int64_t* mask, const float* stats, float* row_lse,
                 float* row_nll, T* dlogits, hipStream_t s) {
  constexpr int VEC = Elt<T>::VEC;
  const int64_t need = ALIGNED ? V / VEC : (V + 2 * (VEC - 1)) / VEC;  // slots incl. worst-case lead
  const dim3 grid(static_cast<unsigned>(B * Tg));
  const int Tgi = static_cast<int>(Tg), Vi = static_cast<int>(V);
  constexpr int S_BIG = 64 / VEC;    // 64 floats per lane: <=128 VGPRs, 4 waves/SIMD
  constexpr int S_SMALL = 16 / VEC;  // 16 floats per lane
  // A/B knobs of the write stream (profiles/history/r04_ce_row_order.txt): DALM_CE_ORDER = 0 identity | i sample-interleaved |
  // <s> stride inside a sample (made coprime to Tg here); DALM_CE_FILL = c cached zero fill | n non-temporal
  static const char* order_env = getenv("DALM_CE_ORDER");
  static const char* fill_env = getenv("DALM_CE_FILL");
  int order = kDefaultRowOrder;
  if (order_env) order = (order_env[0] == 'i') ? -1 : atoi(order_env);
  if (order > 0) {
    order %= Tgi;
    auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
    while (order > 1 && gcd(order, Tgi) != 1) ++order;
    if (order <= 1) order = 0;
  }
  if (!GRAD) order = 0;   // a forward-only launch writes nothing: nothing to interleave
  const bool nt_fill = fill_env ? fill_env[0] == 'n' : kDefaultNtFill;
#define DALM_CE_ARGS logits, sb, st, ids, mask, Tgi, Vi, stats, row_lse, row_nll, dlogits, order
  // tuning knob for A/B runs (tools/kernel_bench.py): DALM_CE_VARIANT=stream | t256
  static const char* variant = getenv("DALM_CE_VARIANT");
  constexpr int S_WIDE = 128 / VEC;  // 128 floats per lane, 4 waves per row (fewer barrier participants)
  // forward-only bf16 rows: the online streaming kernel wins (measured 63 vs 76 us at V=32000, 108 vs
  // 171 us at V=65024): 8 waves/SIMD of independent 16-byte streams, no register-resident phases
  const bool prefer_stream = !GRAD && sizeof(T) == 2 && V >= 8192;
  if ((variant && variant[0] == 's') || (!variant && prefer_stream))
    hipLaunchKernelGGL((marg_ce_stream_kernel<T, 1024, GRAD>), grid, dim3(1024), 0, s, DALM_CE_ARGS);
  else if (variant && variant[0] == 't' && need <= 256 * S_WIDE)
    hipLaunchKernelGGL((marg_ce_row_kernel<T, 256, S_WIDE, GRAD, ALIGNED>), grid, dim3(256), 0, s, DALM_CE_ARGS);
  else if (sizeof(T) == 2 && !(variant && variant[0] == 'u') && need > 256 * S_SMALL && need <= 1024 * 8) {
    if constexpr (sizeof(T) == 2) {  // packed-register bf16 rows (see marg_ce_row_bf16_kernel)
      // cache policy of the read / write streams (DALM_CE_VARIANT=c cached, l nt loads only, w nt stores only, n both)
      int nt = 3;
      if (variant && variant[0] == 'c') nt = 0;
      else if (variant && variant[0] == 'l') nt = 1;
      else if (variant && variant[0] == 'w') nt = 2;
      if (nt == 3 && nt_fill) nt = 7;
#define DALM_CE_BF16(BSZ, P) hipLaunchKernelGGL((marg_ce_row_bf16_kernel<BSZ, 8, GRAD, ALIGNED, P>), grid, dim3(BSZ), 0, s, DALM_CE_ARGS)
      static const bool wide = getenv("DALM_CE_BS") && atoi(getenv("DALM_CE_BS")) == 1024;   // A/B knob: 16 waves per row
      if (need <= 512 * 4)
        hipLaunchKernelGGL((marg_ce_row_bf16_kernel<512, 4, GRAD, ALIGNED>), grid, dim3(512), 0, s, DALM_CE_ARGS);
      else if (wide && need <= 1024 * 4) {
        if (nt == 3) hipLaunchKernelGGL((marg_ce_row_bf16_kernel<1024, 4, GRAD, ALIGNED, 3>), grid, dim3(1024), 0, s, DALM_CE_ARGS);
        else hipLaunchKernelGGL((marg_ce_row_bf16_kernel<1024, 4, GRAD, ALIGNED, 0>), grid, dim3(1024), 0, s, DALM_CE_ARGS);
      } else if (need <= 512 * 8) {
        if (nt == 7) DALM_CE_BF16(512, 7); else if (nt == 3) DALM_CE_BF16(512, 3); else if (nt == 2) DALM_CE_BF16(512, 2); else if (nt == 1) DALM_CE_BF16(512, 1); else DALM_CE_BF16(512, 0);
      } else {
        if (nt == 7) DALM_CE_BF16(1024, 7); else if (nt == 3) DALM_CE_BF16(1024, 3); else if (nt == 2) DALM_CE_BF16(1024, 2); else if (nt == 1) DALM_CE_BF16(1024, 1); else DALM_CE_BF16(1024, 0);
      }
#undef DALM_CE_BF16
    }
  } else if (need <= 256 * S_SMALL)
    hipLaunchKernelGGL((marg_ce_row_kernel<T, 256, S_SMALL, GRAD, ALIGNED>), grid, dim3(256), 0, s, DALM_CE_ARGS);
  else if (need <= 256 * S_BIG)
    hipLaunchKernelGGL((marg_ce_row_kernel<T, 256, S_BIG, GRAD, ALIGNED>), grid, dim3(256), 0, s, DALM_CE_ARGS);
  else if (need <= 512 * S_BIG)
    hipLaunchKernelGGL((marg_ce_row_kernel<T, 512, S_BIG, GRAD, ALIGNED>), grid, dim3(512), 0, s, DALM_CE_ARGS);
  else if (need <= 1024 * S_BIG)
    hipLaunchKernelGGL((marg_ce_row_kernel<T, 1024, S_BIG, GRAD, ALIGNED>), grid, dim3(1024), 0, s, DALM_CE_ARGS);
  else
    hipLaunchKernelGGL((marg_ce_stream_kernel<T, 1024, GRAD>), grid, dim3(1024), 0, s, DALM_CE_ARGS);
#undef DALM_CE_ARGS
}

template <typename T, bool GRAD>
int launch_fwd(const T* logits, int64_t B, int64_t Tg, int64_t V, int64_t sb, int64_t st,
               const int64_t* ids, const int64_t* mask, const float* stats, float* row_lse,
               float* row_nll, T* dlogits, hipStream_t s) {
  constexpr int VEC = Elt<T>::VEC;
  const bool aligned = (reinterpret_cast<uintptr_t>(logits) % 16 == 0) && (V % VEC == 0) &&
                       (sb % VEC == 0) && (st % VEC == 0) &&
                       (!GRAD || reinterpret_cast<uintptr_t>(dlogits) % 16 == 0);
  if (aligned) launch_fwd2<T, GRAD, true>(logits, B, Tg, V, sb, st, ids, mask, stats, row_lse, row_nll, dlogits, s);
  else launch_fwd2<T, GRAD, false>(logits, B, Tg, V, sb, st, ids, mask, stats, row_lse, row_nll, dlogits, s);
  return 0;
}

int check_common(const void* logits, int dtype, int64_t B, int64_t Tg, int64_t V, int64_t sb, int64_t st,
                 const int64_t* ids, const int64_t* mask, const float* stats, const char* fn) {
  if (!logits || !ids || !mask || !stats) return fail(DALM_E_NULL, fn, "null pointer argument");
  if (dtype != DALM_F32 && dtype != DALM_BF16) return fail(DALM_E_DTYPE, fn, "dtype must be DALM_F32 or DALM_BF16");
  if (B <= 0 || Tg < 2 || V <= 0) return fail(DALM_E_SHAPE, fn, "need B>0, Tg>=2, V>0");
  if (st < V || sb < Tg * st) return fail(DALM_E_SHAPE, fn, "strides must describe non-overlapping rows");
  if (B * Tg > 0x7fffffffll || V > 0x3fffffffll) return fail(DALM_E_SHAPE, fn, "B*Tg or V too large");
  const size_t es = (dtype == DALM_F32) ? 4 : 2;
  if (reinterpret_cast<uintptr_t>(logits) % es) return fail(DALM_E_ALIGN, fn, "logits not element-aligned");
  return 0;
}

}  // namespace
}  // namespace dalm

using namespace dalm;

extern "C" int dalm_marg_ce_prep(const int64_t* mask, const int64_t* qlen, int64_t B, int64_t Tg,
                                 float* stats, float* Nb, float* Mb, dalm_stream_t stream) {
  DALM_REQUIRE(mask && stats && Nb && Mb, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(B > 0 && Tg >= 2 && B <= 0x7fffffffll && Tg <= 0x7fffffffll, DALM_E_SHAPE, "need B>0, Tg>=2");
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(ce_prep_rows_kernel, dim3(static_cast<unsigned>(B)), dim3(64), 0, s, mask, qlen,
                     static_cast<int>(Tg), Mb, Nb);
  hipLaunchKernelGGL(ce_prep_total_kernel, dim3(1), dim3(256), 0, s, Mb, static_cast<int>(B), stats);
  return check_launch(__func__);
}

extern "C" int dalm_marg_ce_fwd(const void* logits, int dtype, int64_t B, int64_t Tg, int64_t V,
                                int64_t stride_b, int64_t stride_t, const int64_t* ids,
                                const int64_t* mask, const float* stats, float* row_lse, float* row_nll,
                                void* dlogits, dalm_stream_t stream) {
  if (int e = check_common(logits, dtype, B, Tg, V, stride_b, stride_t, ids, mask, stats, __func__)) return e;
  DALM_REQUIRE(row_lse && row_nll, DALM_E_NULL, "row_lse/row_nll are required");
  hipStream_t s = as_stream(stream);
  if (dtype == DALM_F32) {
    auto* x = static_cast<const float*>(logits);
    auto* g = static_cast<float*>(dlogits);
    if (g) launch_fwd<float, true>(x, B, Tg, V, stride_b, stride_t, ids, mask, stats, row_lse, row_nll, g, s);
    else launch_fwd<float, false>(x, B, Tg, V, stride_b, stride_t, ids, mask, stats, row_lse, row_nll, g, s);
  } else {
    auto* x = static_cast<const bf16_t*>(logits);
    auto* g = static_cast<bf16_t*>(dlogits);
    if (g) launch_fwd<bf16_t, true>(x, B, Tg, V, stride_b, stride_t, ids, mask, stats, row_lse, row_nll, g, s);
    else launch_fwd<bf16_t, false>(x, B, Tg, V, stride_b, stride_t, ids, mask, stats, row_lse, row_nll, g, s);
  }
  return check_launch(__func__);
}

namespace {
int marg_ce_bwd_impl(const void* logits, int dtype, int64_t B, int64_t Tg, int64_t V, int64_t stride_b, int64_t stride_t,
                     const int64_t* ids, const int64_t* mask, const float* stats, const float* row_lse, const float* gscale,
                     const float* row_w, void* dlogits, dalm_stream_t stream, const char* fn) {
  if (int e = check_common(logits, dtype, B, Tg, V, stride_b, stride_t, ids, mask, stats, fn)) return e;
  if (!(row_lse && dlogits)) return fail(DALM_E_NULL, fn, "row_lse/dlogits are required");
  hipStream_t s = as_stream(stream);
  const dim3 grid(static_cast<unsigned>(B * Tg));
  const int Tgi = static_cast<int>(Tg), Vi = static_cast<int>(V);
  if (dtype == DALM_F32)
    hipLaunchKernelGGL((marg_ce_bwd_kernel<float, 256>), grid, dim3(256), 0, s,
                       static_cast<const float*>(logits), stride_b, stride_t, ids, mask, Tgi, Vi, stats,
                       row_lse, gscale, static_cast<float*>(dlogits), row_w);
  else
    hipLaunchKernelGGL((marg_ce_bwd_kernel<bf16_t, 256>), grid, dim3(256), 0, s,
                       static_cast<const bf16_t*>(logits), stride_b, stride_t, ids, mask, Tgi, Vi, stats,
                       row_lse, gscale, static_cast<bf16_t*>(dlogits), row_w);
  return check_launch(fn);
}
}  // namespace

extern "C" int dalm_marg_ce_bwd(const void* logits, int dtype, int64_t B, int64_t Tg, int64_t V,
                                int64_t stride_b, int64_t stride_t, const int64_t* ids,
                                const int64_t* mask, const float* stats, const float* row_lse,
                                const float* gscale, void* dlogits, dalm_stream_t stream) {
  return marg_ce_bwd_impl(logits, dtype, B, Tg, V, stride_b, stride_t, ids, mask, stats, row_lse, gscale, nullptr, dlogits,
                          stream, __func__);
}

extern "C" int dalm_marg_ce_bwd_weighted(const void* logits, int dtype, int64_t B, int64_t Tg, int64_t V,
                                         int64_t stride_b, int64_t stride_t, const int64_t* ids,
                                         const int64_t* mask, const float* stats, const float* row_lse,
                                         const float* gscale, const float* row_weight, void* dlogits,
                                         dalm_stream_t stream) {
  DALM_REQUIRE(row_weight, DALM_E_NULL, "row_weight is required (dalm_marg_ce_bwd is the scalar m / M form)");
  return marg_ce_bwd_impl(logits, dtype, B, Tg, V, stride_b, stride_t, ids, mask, stats, row_lse, gscale, row_weight, dlogits,
                          stream, __func__);
}

extern "C" int dalm_scale_inplace(void* x, int dtype, int64_t n, const float* gscale, dalm_stream_t stream) {
  DALM_REQUIRE(x && gscale, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(n >= 0, DALM_E_SHAPE, "n must be >= 0");
  DALM_REQUIRE(dtype == DALM_F32 || dtype == DALM_BF16, DALM_E_DTYPE, "bad dtype");
  DALM_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0, DALM_E_ALIGN, "x must be 16-byte aligned");
  if (n == 0) return 0;
  hipStream_t s = as_stream(stream);
  const int64_t per = (dtype == DALM_F32) ? 4 : 8;
  int64_t blocks = (n / per + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  if (dtype == DALM_F32)
    hipLaunchKernelGGL(scale_inplace_kernel<float>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                       static_cast<float*>(x), n, gscale);
  else
    hipLaunchKernelGGL(scale_inplace_kernel<bf16_t>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s,
                       static_cast<bf16_t*>(x), n, gscale);
  return check_launch(__func__);
}

extern "C" int dalm_marg_ce_finalize(const float* row_nll, int64_t num_rows, const float* Nb,
                                     const float* doc_lp, int64_t B, const float* stats, float* out,
                                     dalm_stream_t stream) {
  DALM_REQUIRE(row_nll && stats && out, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(!doc_lp || Nb, DALM_E_NULL, "Nb is required when doc_lp is given");
  DALM_REQUIRE(num_rows > 0 && B > 0, DALM_E_SHAPE, "need num_rows>0, B>0");
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1024), 0, as_stream(stream), row_nll, num_rows, Nb,
                     doc_lp, static_cast<int>(B), stats, out);
  return check_launch(__func__);
}

extern "C" int dalm_marg_ce_finalize_topk(const float* row_nll, int64_t B, int64_t k, int64_t Tg, const int64_t* cut,
                                          const float* Nb, const float* doc_lp, const float* stats, float* out,
                                          float* weights, dalm_stream_t stream) {
  DALM_REQUIRE(row_nll && Nb && doc_lp && stats && out, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(B > 0 && k > 0 && Tg > 0 && B * k * Tg < (1ll << 31), DALM_E_SHAPE, "need B, k, Tg > 0");
  if (k == 1 && !weights)      // the reference's case: the SAME kernel and reduction order as dalm_marg_ce_finalize, bit for bit
    return dalm_marg_ce_finalize(row_nll, B * Tg, Nb, doc_lp, B, stats, out, stream);
  DALM_REQUIRE(cut, DALM_E_NULL, "cut is required for k > 1 (or when the weights are asked for)");
  hipLaunchKernelGGL(ce_finalize_topk_kernel, dim3(1), dim3(256), 0, as_stream(stream), row_nll, static_cast<int>(B),
                     static_cast<int>(k), static_cast<int>(Tg), cut, Nb, doc_lp, stats, out, weights);
  return check_launch(__func__);
}

extern "C" int dalm_gather_nll(const float* lp, const int64_t* labels, int64_t R, int64_t V, float* out,
                               dalm_stream_t stream) {
  DALM_REQUIRE(lp && labels && out, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(R >= 0 && V > 0, DALM_E_SHAPE, "need R>=0, V>0");
  if (R == 0) return 0;
  hipLaunchKernelGGL(gather_nll_kernel, dim3(static_cast<unsigned>((R + 255) / 256)), dim3(256), 0,
                     as_stream(stream), lp, labels, R, V, out);
  return check_launch(__func__);
}

extern "C" int dalm_marginalize_rows(const float* lp, int64_t T, int64_t V, const float* doc_lp,
                                     int64_t qlen, float* out, dalm_stream_t stream) {
  DALM_REQUIRE(lp && doc_lp && out, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(T >= 0 && V > 0, DALM_E_SHAPE, "need T>=0, V>0");
  if (T == 0) return 0;
  int64_t blocks = (T * V + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(marginalize_rows_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     as_stream(stream), lp, T, V, doc_lp, qlen, static_cast<const int64_t*>(nullptr), out);
  return check_launch(__func__);
}

extern "C" int dalm_marginalize_rows_dev(const float* lp, int64_t T, int64_t V, const float* doc_lp,
                                         const int64_t* qlen_dev, float* out, dalm_stream_t stream) {
  DALM_REQUIRE(lp && doc_lp && qlen_dev && out, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(T >= 0 && V > 0, DALM_E_SHAPE, "need T>=0, V>0");
  if (T == 0) return 0;
  int64_t blocks = (T * V + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(marginalize_rows_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                     as_stream(stream), lp, T, V, doc_lp, static_cast<int64_t>(0), qlen_dev, out);
  return check_launch(__func__);
}

// ---------------------------------------------------------------------------------------------------------------------
// k retrieved contexts per sample, end to end (round 4; the reference is k = 1: train_utils.py:123-124, and only muses about
// more at train_rage2e.py:461-462).  Scores of the k contexts of query b and their log-softmax over the k (RAG-token:
// p(c | q_b) = softmax_c(scale q_b . P[b,c])), and the closed-form backward from the per-row weights that
// dalm_marg_ce_finalize_topk returns (w = -dL/d(label log-prob), 1/M included):
//     dL/d doc_lp[b,c] = -sum_{j < N_b} w[b, c, cut_bc + j]           (the posterior mass of context c over the answer rows)
//     dL/d s[b,c]      = g_bc - p_bc sum_c' g_bc'                      (through the log-softmax over the k contexts)
//     dq_b = scale sum_c ds_bc P[b,c]        dP[b,c] = scale ds_bc q_b
// One workgroup per sample; k <= 64.
// ---------------------------------------------------------------------------------------------------------------------
namespace dalm {
namespace {

constexpr int kMaxCtx = 64;

__global__ __launch_bounds__(256) void doc_scores_topk_fwd_kernel(const float* __restrict__ q, const float* __restrict__ P,
                                                                  int k, int D, float scale, float* __restrict__ s_out,
                                                                  float* __restrict__ doc_lp) {
  __shared__ float red[4];
  __shared__ float sc[kMaxCtx];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* qb = q + static_cast<int64_t>(b) * D;
  for (int c = 0; c < k; ++c) {
    const float* pc = P + (static_cast<int64_t>(b) * k + c) * D;
    float a = 0.f;
    for (int d = tid; d < D; d += 256) a = fmaf(qb[d], pc[d], a);
    a = block_sum<256>(a, red);
    if (tid == 0) sc[c] = __fmul_rn(scale, a);
  }
  __syncthreads();
  if (tid == 0) {
    float mx = -INFINITY;
    for (int c = 0; c < k; ++c) mx = fmaxf(mx, sc[c]);
    float se = 0.f;
    for (int c = 0; c < k; ++c) se += __expf(sc[c] - mx);
    const float lse = mx + __logf(se);
    for (int c = 0; c < k; ++c) {
      s_out[b * k + c] = sc[c];
      doc_lp[b * k + c] = sc[c] - lse;
    }
  }
}

__global__ __launch_bounds__(256) void doc_scores_topk_bwd_kernel(const float* __restrict__ q, const float* __restrict__ P,
                                                                  int k, int D, int Tg, float scale,
                                                                  const float* __restrict__ doc_lp,
                                                                  const float* __restrict__ weights,
                                                                  const int64_t* __restrict__ cut,
                                                                  const float* __restrict__ Nb,
                                                                  const float* __restrict__ gscale, float* __restrict__ dq,
                                                                  float* __restrict__ dP, float* __restrict__ ds_out) {
  __shared__ float gs[kMaxCtx], ds[kMaxCtx];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nans = static_cast<int>(Nb[b]);
  const float g0 = gscale ? gscale[0] : 1.f;
  for (int c = wave; c < k; c += 4) {                        // one wave per context: fixed order inside the wave
    const int64_t base = (static_cast<int64_t>(b) * k + c) * Tg;
    const int cu = static_cast<int>(cut[b * k + c]);
    float a = 0.f;
    for (int j = lane; j < nans; j += 64) {
      const int t = cu + j;
      if (t >= 0 && t < Tg) a += weights[base + t];
    }
    a = wave_sum(a);
    if (lane == 0) gs[c] = -g0 * a;
  }
  __syncthreads();
  if (tid == 0) {
    float G = 0.f;
    for (int c = 0; c < k; ++c) G += gs[c];
    for (int c = 0; c < k; ++c) {
      ds[c] = gs[c] - __expf(doc_lp[b * k + c]) * G;
      if (ds_out) ds_out[b * k + c] = ds[c];
    }
  }
  __syncthreads();
  const float* qb = q + static_cast<int64_t>(b) * D;
  for (int d = tid; d < D; d += 256) {
    float a = 0.f;
    const float qv = qb[d];
    for (int c = 0; c < k; ++c) {
      const int64_t at = (static_cast<int64_t>(b) * k + c) * D + d;
      a = fmaf(ds[c], P[at], a);
      if (dP) dP[at] = scale * ds[c] * qv;
    }
    if (dq) dq[static_cast<int64_t>(b) * D + d] = scale * a;
  }
}

}  // namespace
}  // namespace dalm

extern "C" int dalm_doc_scores_topk_fwd(const float* q, const float* P, int64_t B, int64_t k, int64_t D, float scale,
                                        float* scores, float* doc_lp, dalm_stream_t stream) {
  DALM_REQUIRE(q && P && scores && doc_lp, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(B > 0 && k > 0 && k <= kMaxCtx && D > 0 && B <= 0x7fffffffll && D <= 0x7fffffffll, DALM_E_SHAPE,
               "need B > 0, 0 < k <= 64, D > 0");
  hipLaunchKernelGGL(doc_scores_topk_fwd_kernel, dim3(static_cast<unsigned>(B)), dim3(256), 0, as_stream(stream), q, P,
                     static_cast<int>(k), static_cast<int>(D), scale, scores, doc_lp);
  return check_launch(__func__);
}

extern "C" int dalm_doc_scores_topk_bwd(const float* q, const float* P, int64_t B, int64_t k, int64_t D, int64_t Tg,
                                        float scale, const float* doc_lp, const float* weights, const int64_t* cut,
                                        const float* Nb, const float* gscale, float* dq, float* dP, float* dscores,
                                        dalm_stream_t stream) {
  DALM_REQUIRE(q && P && doc_lp && weights && cut && Nb, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dq || dP || dscores, DALM_E_NULL, "at least one output is required");
  DALM_REQUIRE(B > 0 && k > 0 && k <= kMaxCtx && D > 0 && Tg > 0 && B <= 0x7fffffffll && D <= 0x7fffffffll &&
               B * k * Tg < (1ll << 31), DALM_E_SHAPE, "need B > 0, 0 < k <= 64, D > 0, Tg > 0");
  hipLaunchKernelGGL(doc_scores_topk_bwd_kernel, dim3(static_cast<unsigned>(B)), dim3(256), 0, as_stream(stream), q, P,
                     static_cast<int>(k), static_cast<int>(D), static_cast<int>(Tg), scale, doc_lp, weights, cut, Nb, gscale,
                     dq, dP, dscores);
  return check_launch(__func__);
}
