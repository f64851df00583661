// K1: masked mean-pool + L2 normalise of bi-encoder token states (gfx950).
//
// Stands in for AutoModelForRagE2E.mean_pooling + F.normalize
//   dalm/models/rag_e2e_base_model.py:95-97,108-111
//   dalm/models/retriever_only_base_model.py:60-68
// The reference materialises three [B,T,D] temporaries (mask.expand().float(),
// the product, the sum); here each unmasked token row is read once with 16-byte
// lane accesses and padded tokens are never touched.  HBM-bound:
//   forward : n_tok*D*eh bytes read (n_tok = unmasked tokens) + B*D*4 written
//   backward: B*T*D*eh bytes written
#include "common.hpp"

namespace dalm {
namespace {

struct bf16_t { unsigned short v; };
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

template <typename T> struct HV;
template <> struct HV<float> {
  static constexpr int VEC = 4;
  __device__ static __forceinline__ void load(const float* p, int nvalid, bool vec, float (&x)[4]) {
    if (nvalid >= 4 && vec) {
      // cached loads on purpose: the token states were just written by the encoder's last layer and sit in
      // L2 / Infinity Cache (non-temporal loads measured 20 % slower here, unlike the CE kernels)
      const float4 v = *reinterpret_cast<const float4*>(p);
      x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = (e < nvalid) ? p[e] : 0.f;
    }
  }
  template <bool NTS = false>
  __device__ static __forceinline__ void store(float* p, int nvalid, bool vec, const float (&x)[4]) {
    if (nvalid >= 4 && vec) {
      if (NTS) {
        v4f v = {x[0], x[1], x[2], x[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
      } else {
        *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nvalid) p[e] = x[e];
    }
  }
};
template <> struct HV<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ void load(const bf16_t* p, int nvalid, bool vec, float (&x)[8]) {
    if (nvalid >= 8 && vec) {
      const uint4 v = *reinterpret_cast<const uint4*>(p);
      const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x[2 * i] = __uint_as_float(w[i] << 16);
        x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = (e < nvalid) ? bf16_to_f32(p[e].v) : 0.f;
    }
  }
  template <bool NTS = false>
  __device__ static __forceinline__ void store(bf16_t* p, int nvalid, bool vec, const float (&x)[8]) {
    if (nvalid >= 8 && vec) {
      unsigned int w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        w[i] = pack_bf16x2(x[2 * i], x[2 * i + 1]);
      if (NTS) {
        v4u v = {w[0], w[1], w[2], w[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<v4u*>(p));
      } else {
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (e < nvalid) p[e].v = f32_to_bf16(x[e]);
    }
  }
};

// grid (DC, B); 4 waves share one 64*VEC-wide d-chunk and split the tokens.
template <typename T>
__global__ __launch_bounds__(256) void pool_sum_kernel(const T* __restrict__ h, const int64_t* __restrict__ mask,
                                                       int Tn, int D, int vec_ok, float* __restrict__ emb,
                                                       float* __restrict__ inv_count) {
  constexpr int VEC = HV<T>::VEC;
  __shared__ float part[4][64 * VEC];
  __shared__ float cnt_part[4];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = (blockIdx.x * 64 + lane) * VEC;
  int nvalid = D - d;
  nvalid = nvalid < 0 ? 0 : (nvalid > VEC ? VEC : nvalid);
  const T* hb = h + static_cast<int64_t>(b) * Tn * D + d;
  const int64_t* mb = mask + static_cast<int64_t>(b) * Tn;

  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  float cnt = 0.f;
  // tokens wave, wave+4, ... ; two in flight per iteration
  for (int t = wave; t < Tn; t += 8) {
    const int t2 = t + 4;
    const int64_t m0 = mb[t];
    const int64_t m1 = (t2 < Tn) ? mb[t2] : 0;
    float x0[VEC], x1[VEC];
    if (m0 != 0 && nvalid > 0) HV<T>::load(hb + static_cast<int64_t>(t) * D, nvalid, vec_ok, x0);
    if (m1 != 0 && nvalid > 0) HV<T>::load(hb + static_cast<int64_t>(t2) * D, nvalid, vec_ok, x1);
    const float f0 = static_cast<float>(m0), f1 = static_cast<float>(m1);
    cnt += f0 + f1;
    if (m0 != 0 && nvalid > 0) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = fmaf(f0, x0[e], acc[e]);
    }
    if (m1 != 0 && nvalid > 0) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = fmaf(f1, x1[e], acc[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) part[wave][lane * VEC + e] = acc[e];
  if (lane == 0) cnt_part[wave] = cnt;
  __syncthreads();
  if (wave == 0) {
    const float c = fmaxf(cnt_part[0] + cnt_part[1] + cnt_part[2] + cnt_part[3], 1e-9f);
    const float ic = 1.f / c;
    if (blockIdx.x == 0 && lane == 0) inv_count[b] = ic;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int i = lane * VEC + e;
      const float s = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
      if (e < nvalid) emb[static_cast<int64_t>(b) * D + d + e] = s / c;
    }
  }
}

// one block per sample: |u| and (optionally) e = u / max(|u|, 1e-12) in place
__global__ __launch_bounds__(256) void l2norm_rows_kernel(float* __restrict__ emb, int D, int normalize,
                                                          float* __restrict__ norm) {
  __shared__ float red[4];
  float* row = emb + static_cast<int64_t>(blockIdx.x) * D;
  float ss = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) ss = fmaf(row[i], row[i], ss);
  ss = block_sum<256>(ss, red);
  const float nrm = sqrtf(ss);
  if (threadIdx.x == 0) norm[blockIdx.x] = nrm;
  if (normalize) {
    const float denom = fmaxf(nrm, 1e-12f);
    for (int i = threadIdx.x; i < D; i += 256) row[i] = row[i] / denom;
  }
}

// ---------------------------------------------------------------------------------------------------
// Fused forward: one 1024-thread workgroup walks the unmasked tokens of ONE sample (or of one token slice of
// it, grid.x = TZ) across the full embedding width, so the mean, |u| and the normalisation happen in the
// same launch (the two-launch form above costs ~20 us at batch 18 for ~2 us of data movement).
//   TPR threads cover one token row (VEC elements each, NCH d-chunks per thread when D > TPR*VEC),
//   G = NT/TPR token groups stride through the tokens, 4-8 token rows in flight per group.
//   NT = 1024 threads when the batch is small (one workgroup has to keep a CU's memory pipe full on its own);
//   NT = 256 for large batches: with >= 2 samples per CU, 1024-thread workgroups run in ceil(B/512) rounds and the
//   reduction tail of each round leaves the memory pipe idle (B = 1200: 3 rounds for 2.3 rounds of work).
//   TZ == 1: sums are combined through LDS in fixed group order, then u, |u|, e are written.
//   TZ  > 1: raw partial sums go to part[b][z][D] (+ counts) and pool_finish_kernel completes the sample;
//            used when B alone cannot occupy the chip (B * bytes per sample is large but B < ~128).
// ---------------------------------------------------------------------------------------------------
template <typename T, int NCH, int NT>
__global__ __launch_bounds__(NT) void pool_fused_kernel(const T* __restrict__ h, const int64_t* __restrict__ mask,
                                                          int Tn, int D, int tpr_log2, int vec_ok, int normalize,
                                                          float* __restrict__ emb, float* __restrict__ norm,
                                                          float* __restrict__ inv_count, float* __restrict__ part,
                                                          float* __restrict__ part_cnt) {
  constexpr int VEC = HV<T>::VEC;
  constexpr int TIF = (NCH == 1) ? 8 : 4;   // token rows in flight per group on the fast path (128 B per thread)
  extern __shared__ float lds[];       // [G][Dp] partial sums, then reduction scratch
  __shared__ float red[16];
  __shared__ float cnt_s[16];
  const int b = blockIdx.y, z = blockIdx.x, TZ = gridDim.x;
  const int tid = threadIdx.x;
  const int TPR = 1 << tpr_log2, G = NT >> tpr_log2;
  const int g = tid >> tpr_log2, c = tid & (TPR - 1);
  const int per = (Tn + TZ - 1) / TZ;
  const int t_lo = z * per, t_hi = min(Tn, t_lo + per);
  const T* hb = h + static_cast<int64_t>(b) * Tn * D;
  const int64_t* mb = mask + static_cast<int64_t>(b) * Tn;

  float acc[NCH][VEC];
#pragma unroll
  for (int q = 0; q < NCH; ++q)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[q][e] = 0.f;
  float cnt = 0.f;
  if (vec_ok) {
    // Fast path (16-byte aligned rows, D % VEC == 0): UNCONDITIONAL vector loads - a predicated load compiles to a
    // branch and the compiler then waits with vmcnt(0), which serialises the four rows in flight.  A padded token
    // is redirected to the slice's first row instead (an L1/L2 hit, no HBM traffic) and its value discarded.
    const int t_anchor = min(t_lo + g, Tn - 1);
    for (int t0 = t_lo + g; t0 < t_hi; t0 += TIF * G) {
      int64_t m[TIF];
#pragma unroll
      for (int u = 0; u < TIF; ++u) {
        const int t = t0 + u * G;
        const int64_t mv = mb[min(t, t_hi - 1)];
        m[u] = (t < t_hi) ? mv : 0;
      }
#pragma unroll
      for (int q = 0; q < NCH; ++q) {
        const int d = (q * TPR + c) * VEC;
        const bool dok = d < D;
        const int dd = dok ? d : 0;
        float x[TIF][VEC];
#pragma unroll
        for (int u = 0; u < TIF; ++u) {
          const int tt = (m[u] != 0) ? (t0 + u * G) : t_anchor;
          HV<T>::load(hb + static_cast<int64_t>(tt) * D + dd, VEC, true, x[u]);
        }
#pragma unroll
        for (int u = 0; u < TIF; ++u) {
          const float f = static_cast<float>(m[u]);
          const bool use = (m[u] != 0) && dok;
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[q][e] += use ? f * x[u][e] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < TIF; ++u) cnt += static_cast<float>(m[u]);
    }
  } else {
    for (int t0 = t_lo + g; t0 < t_hi; t0 += 4 * G) {
      int64_t m[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) m[u] = (t0 + u * G < t_hi) ? mb[t0 + u * G] : 0;
#pragma unroll
      for (int q = 0; q < NCH; ++q) {
        const int d = (q * TPR + c) * VEC;
        int nvalid = D - d;
        nvalid = nvalid < 0 ? 0 : (nvalid > VEC ? VEC : nvalid);
        float x[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (m[u] != 0 && nvalid > 0) HV<T>::load(hb + static_cast<int64_t>(t0 + u * G) * D + d, nvalid, false, x[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (m[u] != 0 && nvalid > 0) {
            const float f = static_cast<float>(m[u]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[q][e] = fmaf(f, x[u][e], acc[q][e]);
          }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) cnt += static_cast<float>(m[u]);
    }
  }
  // ---- combine the G token groups (fixed order) ----
  const int Dp = TPR * VEC * NCH;
#pragma unroll
  for (int q = 0; q < NCH; ++q)
#pragma unroll
    for (int e = 0; e < VEC; ++e) lds[g * Dp + (q * TPR + c) * VEC + e] = acc[q][e];
  if (c == 0) cnt_s[g] = cnt;
  __syncthreads();
  float total = 0.f;
  for (int i = 0; i < G; ++i) total += cnt_s[i];
  float ss = 0.f;
  float u[NCH][VEC];
  if (g == 0) {
#pragma unroll
    for (int q = 0; q < NCH; ++q)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int i = (q * TPR + c) * VEC + e;
        float s = 0.f;
        for (int k = 0; k < G; ++k) s += lds[k * Dp + i];
        u[q][e] = s;
      }
  }
  if (TZ > 1) {
    if (g == 0) {
      float* pr = part + (static_cast<int64_t>(b) * TZ + z) * D;
#pragma unroll
      for (int q = 0; q < NCH; ++q)
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const int i = (q * TPR + c) * VEC + e;
          if (i < D) pr[i] = u[q][e];
        }
      if (c == 0) part_cnt[b * TZ + z] = total;
    }
    return;
  }
  const float cden = fmaxf(total, 1e-9f);
  if (g == 0) {
#pragma unroll
    for (int q = 0; q < NCH; ++q)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int i = (q * TPR + c) * VEC + e;
        u[q][e] = (i < D) ? u[q][e] / cden : 0.f;
        ss = fmaf(u[q][e], u[q][e], ss);
      }
  }
  ss = block_sum<NT>(ss, red);
  const float nrm = sqrtf(ss);
  if (tid == 0) { norm[b] = nrm; inv_count[b] = 1.f / cden; }
  if (g == 0) {
    const float denom = fmaxf(nrm, 1e-12f);
#pragma unroll
    for (int q = 0; q < NCH; ++q)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int i = (q * TPR + c) * VEC + e;
        if (i < D) emb[static_cast<int64_t>(b) * D + i] = normalize ? u[q][e] / denom : u[q][e];
      }
  }
}

// completes a sample from TZ partial sums (fixed z order): one block per sample
__global__ __launch_bounds__(256) void pool_finish_kernel(const float* __restrict__ part,
                                                          const float* __restrict__ part_cnt, int TZ, int D,
                                                          int normalize, float* __restrict__ emb,
                                                          float* __restrict__ norm, float* __restrict__ inv_count) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  float total = 0.f;
  for (int z = 0; z < TZ; ++z) total += part_cnt[b * TZ + z];
  const float cden = fmaxf(total, 1e-9f);
  float* row = emb + static_cast<int64_t>(b) * D;
  float ss = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) {
    float pv[32];   // TZ <= 32: one batch of loads (a `for z < TZ` loop of loads is a serial latency chain)
#pragma unroll
    for (int z = 0; z < 32; ++z) pv[z] = part[(static_cast<int64_t>(b) * TZ + min(z, TZ - 1)) * D + i];
    float s = 0.f;
#pragma unroll
    for (int z = 0; z < 32; ++z) s += (z < TZ) ? pv[z] : 0.f;
    s /= cden;
    row[i] = s;   // re-read below by the same thread only
    ss = fmaf(s, s, ss);
  }
  ss = block_sum<256>(ss, red);
  const float nrm = sqrtf(ss);
  if (threadIdx.x == 0) { norm[b] = nrm; inv_count[b] = 1.f / cden; }
  if (normalize) {
    const float denom = fmaxf(nrm, 1e-12f);
    for (int i = threadIdx.x; i < D; i += 256) row[i] = row[i] / denom;
  }
}

// grid (DC, B, TZ).  dh[b,t,:] = mask[b,t] * inv_count[b] * du[b,:], with
//   du = d_emb                                  (normalize == 0)
//   du = (d_emb - e (e . d_emb)) / |u|           (|u| >= 1e-12)
//   du = d_emb / 1e-12                           (|u| <  1e-12: clamp_min branch)
template <typename T, bool NTS>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ d_emb,
                                                       const float* __restrict__ emb,
                                                       const float* __restrict__ norm,
                                                       const float* __restrict__ inv_count,
                                                       const int64_t* __restrict__ mask, int Tn, int D,
                                                       int normalize, int vec_ok, T* __restrict__ dh) {
  constexpr int VEC = HV<T>::VEC;
  __shared__ float red[4];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = (blockIdx.x * 64 + lane) * VEC;
  int nvalid = D - d;
  nvalid = nvalid < 0 ? 0 : (nvalid > VEC ? VEC : nvalid);
  const float* de = d_emb + static_cast<int64_t>(b) * D;
  const float* eb = emb + static_cast<int64_t>(b) * D;

  float g[VEC];
  const float ic = inv_count[b];
  if (normalize) {
    float dot = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) dot = fmaf(eb[i], de[i], dot);
    dot = block_sum<256>(dot, red);
    const float nrm = norm[b];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float v = 0.f;
      if (e < nvalid) v = (nrm >= 1e-12f) ? (de[d + e] - eb[d + e] * dot) / nrm : de[d + e] / 1e-12f;
      g[e] = v * ic;
    }
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] = (e < nvalid) ? de[d + e] * ic : 0.f;
  }
  if (nvalid == 0) return;
  T* hb = dh + static_cast<int64_t>(b) * Tn * D + d;
  const int64_t* mb = mask + static_cast<int64_t>(b) * Tn;
  const int tz = gridDim.z, z = blockIdx.z;
  const int per = (Tn + tz - 1) / tz;
  const int t_end = min(Tn, (z + 1) * per);
  for (int t = z * per + wave; t < t_end; t += 4) {
    const float f = static_cast<float>(mb[t]);
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = f * g[e];
    HV<T>::template store<NTS>(hb + static_cast<int64_t>(t) * D, nvalid, vec_ok, o);
  }
}


// Row-major backward for large batches: grid (TZ, B), 256 threads.  TPR threads cover one token row (NCH d-chunks each), the
// G = 256/TPR token groups write CONSECUTIVE rows, so a workgroup streams whole contiguous rows (2-4 KB per step at D = 1024)
// and rebuilds du once per sample slice instead of once per 64*VEC-wide d-chunk.  Needs 16-byte aligned rows (vec_ok).
template <typename T, int NCH, bool NTS>
__global__ __launch_bounds__(256) void pool_bwd_rows_kernel(const float* __restrict__ d_emb,
                                                            const float* __restrict__ emb,
                                                            const float* __restrict__ norm,
                                                            const float* __restrict__ inv_count,
                                                            const int64_t* __restrict__ mask, int Tn, int D,
                                                            int normalize, int tpr_log2, T* __restrict__ dh) {
  constexpr int VEC = HV<T>::VEC;
  __shared__ float red[4];
  const int b = blockIdx.y, z = blockIdx.x, TZ = gridDim.x, tid = threadIdx.x;
  const int TPR = 1 << tpr_log2, G = 256 >> tpr_log2;
  const int g = tid >> tpr_log2, c = tid & (TPR - 1);
  const float* de = d_emb + static_cast<int64_t>(b) * D;
  const float* eb = emb + static_cast<int64_t>(b) * D;
  const float ic = inv_count[b];
  float dot = 0.f, nrm = 1.f;
  if (normalize) {
    for (int i = tid; i < D; i += 256) dot = fmaf(eb[i], de[i], dot);
    dot = block_sum<256>(dot, red);
    nrm = norm[b];
  }
  float gv[NCH][VEC];
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const int d = (q * TPR + c) * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float v = 0.f;
      if (d + e < D) {
        if (!normalize) v = de[d + e];
        else v = (nrm >= 1e-12f) ? (de[d + e] - eb[d + e] * dot) / nrm : de[d + e] / 1e-12f;
      }
      gv[q][e] = v * ic;
    }
  }
  const int per = (Tn + TZ - 1) / TZ;
  const int t_lo = z * per, t_hi = min(Tn, t_lo + per);
  T* hb = dh + static_cast<int64_t>(b) * Tn * D;
  const int64_t* mb = mask + static_cast<int64_t>(b) * Tn;
  for (int t0 = t_lo + g; t0 < t_hi; t0 += 4 * G) {
    float f[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) f[u] = static_cast<float>(mb[min(t0 + u * G, t_hi - 1)]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * G;
      if (t < t_hi) {
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
          const int d = (q * TPR + c) * VEC;
          if (d < D) {
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = f[u] * gv[q][e];
            HV<T>::template store<NTS>(hb + static_cast<int64_t>(t) * D + d, VEC, true, o);
          }
        }
      }
    }
  }
}

}  // namespace
}  // namespace dalm

using namespace dalm;

namespace dalm {
namespace {
// Geometry of the fused forward: TPR threads per token row (power of two), NCH chunks per thread, TZ token slices.
struct PoolPlan { int tpr_log2, nch, tz, nt; bool fused; };
inline PoolPlan pool_plan(int64_t B, int64_t T, int64_t D, int vec) {
  PoolPlan p{0, 1, 1, 1024, false};
  const int64_t lanes = (D + vec - 1) / vec;
  int lg = 6;                                  // at least one wave per row
  while ((1ll << lg) < lanes && lg < 10) ++lg;
  const int64_t tpr = 1ll << lg;
  const int64_t nch = (lanes + tpr - 1) / tpr;
  const int64_t lds_bytes = (1024 / tpr) * tpr * vec * nch * 4;
  if (nch > 4 || lds_bytes > 64 * 1024) return p;   // very wide rows: two-launch kernels
  p.tpr_log2 = lg; p.nch = static_cast<int>(nch); p.fused = true;
  // token slices: only when the batch alone leaves most CUs idle AND a sample is big enough to matter
  const int64_t groups = 1024 / tpr;
  int64_t tz = 1;
  // one CU streams ~50-80 GB/s on its own: slice when a sample is more than a few microseconds of that
  static const int64_t slice_bytes = [] { const char* e = getenv("DALM_POOL_SLICE_BYTES"); return e ? atoll(e) : (1ll << 20); }();
  if (B < 64 && T * D * (16 / vec) >= slice_bytes) {
    tz = (192 + B - 1) / B;
    const int64_t tz_max = (T + 4 * groups - 1) / (4 * groups);
    if (tz > tz_max) tz = tz_max;
    if (tz > 32) tz = 32;
    if (tz < 1) tz = 1;
  }
  p.tz = static_cast<int>(tz);
  static const int nt_env = [] { const char* e = getenv("DALM_POOL_NT"); return e ? atoi(e) : 0; }();
  if (tz == 1 && tpr <= 256 && (nt_env == 256 || (nt_env == 0 && B >= 512))) p.nt = 256;
  return p;
}
template <typename T>
void launch_pool_fused(const PoolPlan& pl, const T* h, const int64_t* mask, int B, int Tn, int D, int vok, int normalize,
                       float* emb, float* norm, float* inv_count, float* part, float* part_cnt, hipStream_t s) {
  const int vec = HV<T>::VEC;
  const size_t lds = static_cast<size_t>(pl.nt) * vec * pl.nch * sizeof(float);
  const dim3 grid(static_cast<unsigned>(pl.tz), static_cast<unsigned>(B));
#define DALM_POOL_LAUNCH(N, NT) \
  hipLaunchKernelGGL((pool_fused_kernel<T, N, NT>), grid, dim3(NT), lds, s, h, mask, Tn, D, pl.tpr_log2, vok, normalize, \
                     emb, norm, inv_count, part, part_cnt)
  if (pl.nt == 256) {
    switch (pl.nch) {
      case 1: DALM_POOL_LAUNCH(1, 256); break;
      case 2: DALM_POOL_LAUNCH(2, 256); break;
      case 3: DALM_POOL_LAUNCH(3, 256); break;
      default: DALM_POOL_LAUNCH(4, 256); break;
    }
  } else {
    switch (pl.nch) {
      case 1: DALM_POOL_LAUNCH(1, 1024); break;
      case 2: DALM_POOL_LAUNCH(2, 1024); break;
      case 3: DALM_POOL_LAUNCH(3, 1024); break;
      default: DALM_POOL_LAUNCH(4, 1024); break;
    }
  }
#undef DALM_POOL_LAUNCH
  if (pl.tz > 1)
    hipLaunchKernelGGL(pool_finish_kernel, dim3(static_cast<unsigned>(B)), dim3(256), 0, s, part, part_cnt, pl.tz, D,
                       normalize, emb, norm, inv_count);
}
}  // namespace
}  // namespace dalm

extern "C" size_t dalm_pool_l2norm_fwd_workspace_bytes(int64_t B, int64_t T, int64_t D, int dtype) {
  if (B <= 0 || T <= 0 || D <= 0) return 0;
  const PoolPlan pl = pool_plan(B, T, D, dtype == DALM_F32 ? 4 : 8);
  if (!pl.fused || pl.tz <= 1) return 0;
  return static_cast<size_t>(B) * pl.tz * (static_cast<size_t>(D) + 1) * sizeof(float);
}

extern "C" int dalm_pool_l2norm_fwd_ws(const void* h, int dtype, const int64_t* mask, int64_t B, int64_t T,
                                       int64_t D, int normalize, float* emb, float* norm, float* inv_count,
                                       void* ws, size_t ws_bytes, dalm_stream_t stream) {
  DALM_REQUIRE(h && mask && emb && norm && inv_count, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dtype == DALM_F32 || dtype == DALM_BF16, DALM_E_DTYPE, "dtype must be DALM_F32 or DALM_BF16");
  DALM_REQUIRE(B > 0 && T > 0 && D > 0 && B <= 65535 && T <= 0x7fffffffll && D <= 0x3fffffffll, DALM_E_SHAPE,
               "need 0<B<=65535, T>0, D>0");
  hipStream_t s = as_stream(stream);
  const int vec = (dtype == DALM_F32) ? 4 : 8;
  const int vok = (reinterpret_cast<uintptr_t>(h) % 16 == 0) && (D % vec == 0);
  PoolPlan pl = pool_plan(B, T, D, vec);
  const size_t need = dalm_pool_l2norm_fwd_workspace_bytes(B, T, D, dtype);
  if (pl.fused && pl.tz > 1 && (ws == nullptr || ws_bytes < need)) pl.tz = 1;   // no scratch: one slice per sample
  if (pl.fused) {
    float* part = static_cast<float*>(ws);
    float* part_cnt = part ? part + static_cast<size_t>(B) * pl.tz * D : nullptr;
    if (dtype == DALM_F32)
      launch_pool_fused<float>(pl, static_cast<const float*>(h), mask, static_cast<int>(B), static_cast<int>(T),
                               static_cast<int>(D), vok, normalize, emb, norm, inv_count, part, part_cnt, s);
    else
      launch_pool_fused<bf16_t>(pl, static_cast<const bf16_t*>(h), mask, static_cast<int>(B), static_cast<int>(T),
                                static_cast<int>(D), vok, normalize, emb, norm, inv_count, part, part_cnt, s);
    return check_launch(__func__);
  }
  const dim3 grid(static_cast<unsigned>((D + 64 * vec - 1) / (64 * vec)), static_cast<unsigned>(B));
  if (dtype == DALM_F32)
    hipLaunchKernelGGL(pool_sum_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(h), mask,
                       static_cast<int>(T), static_cast<int>(D), vok, emb, inv_count);
  else
    hipLaunchKernelGGL(pool_sum_kernel<bf16_t>, grid, dim3(256), 0, s, static_cast<const bf16_t*>(h), mask,
                       static_cast<int>(T), static_cast<int>(D), vok, emb, inv_count);
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3(static_cast<unsigned>(B)), dim3(256), 0, s, emb,
                     static_cast<int>(D), normalize, norm);
  return check_launch(__func__);
}

extern "C" int dalm_pool_l2norm_fwd(const void* h, int dtype, const int64_t* mask, int64_t B, int64_t T,
                                    int64_t D, int normalize, float* emb, float* norm, float* inv_count,
                                    dalm_stream_t stream) {
  return dalm_pool_l2norm_fwd_ws(h, dtype, mask, B, T, D, normalize, emb, norm, inv_count, nullptr, 0, stream);
}

extern "C" int dalm_pool_l2norm_bwd(const float* d_emb, const float* emb, const float* norm,
                                    const float* inv_count, const int64_t* mask, int64_t B, int64_t T,
                                    int64_t D, int normalize, void* dh, int dtype, dalm_stream_t stream) {
  DALM_REQUIRE(d_emb && emb && norm && inv_count && mask && dh, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dtype == DALM_F32 || dtype == DALM_BF16, DALM_E_DTYPE, "dtype must be DALM_F32 or DALM_BF16");
  DALM_REQUIRE(B > 0 && T > 0 && D > 0 && B <= 65535 && T <= 0x7fffffffll && D <= 0x3fffffffll, DALM_E_SHAPE,
               "need 0<B<=65535, T>0, D>0");
  hipStream_t s = as_stream(stream);
  const int vec = (dtype == DALM_F32) ? 4 : 8;
  const int vok = (reinterpret_cast<uintptr_t>(dh) % 16 == 0) && (D % vec == 0);
  const int64_t dc = (D + 64 * vec - 1) / (64 * vec);
  // token slices: every workgroup first rebuilds du (dot product over D, ~16 scalar loads per lane), so many thin slices
  // cost more than they spread - measured (tools/kernel_bench.py, hipGraph timing): cfg3 passage bf16 [18,128,1024]
  // 1044 WGs 12.4 us, 540 WGs 9.2 us, 396 WGs 7.7 us; at B = 150 two slices beat one (17.3 vs 19.2 us bf16)
  int64_t tz = (384 + B * dc - 1) / (B * dc);
  if (B * dc < 1024 && T >= 64 && tz < 2) tz = 2;
  // (slicing large batches further does NOT help: every workgroup first rebuilds du, and at [1200,128,1024] bf16 2 / 4 / 8
  // slices measured 91 / 123 / 192 us against 81 us for one - profiles/history/r04_pool_probe_bwd_tz.txt; large batches take the
  // row-major kernel below instead)
  static const int tz_env = [] { const char* e = getenv("DALM_POOL_BWD_TZ"); return e ? atoi(e) : 0; }();
  if (tz_env > 0) tz = tz_env;
  const int64_t tz_max = (T + 3) / 4;
  if (tz > tz_max) tz = tz_max;
  if (tz < 1) tz = 1;
  if (tz > 64) tz = 64;
  // cached stores: the encoder's backward reads dh next.  DALM_POOL_BWD_NT=1 selects non-temporal stores (measured alone:
  // 17.5 -> 13.8 us at [150,128,1024] bf16, 79 -> 84 us at B = 1200; profiles/history/r04_pool_probe.txt)
  static const int nts_env = [] { const char* e = getenv("DALM_POOL_BWD_NT"); return e ? atoi(e) : -1; }();
  const bool nts = nts_env > 0;
  static const int rows_env = [] { const char* e = getenv("DALM_POOL_BWD_ROWS"); return e ? atoi(e) : -1; }();
  PoolPlan pl{6, 1, 1, 256, true};                 // rows kernel: TPR <= 256 threads per row, up to 4 d-chunks per thread
  const int64_t row_lanes = (D + vec - 1) / vec;
  while ((1ll << pl.tpr_log2) < row_lanes && pl.tpr_log2 < 8) ++pl.tpr_log2;
  pl.nch = static_cast<int>((row_lanes + (1ll << pl.tpr_log2) - 1) >> pl.tpr_log2);
  const bool rows_ok = vok && pl.nch <= 4;
  const bool rows = rows_ok && (rows_env >= 0 ? rows_env != 0 : true);
  if (rows) {
    // token slices: ~768 workgroups, at most 8 slices (measured, profiles/history/r04_pool_probe_bwd_rows.txt: [18,128,1024] bf16
    // 16.8 / 10.9 / 7.9 / 6.5 / 7.5 us for 1 / 2 / 4 / 8 / 16 slices; [1200,128,1024] 71.7 / 78.6 / 86.4 / 112 us for 1 / 2 / 4 / 8)
    int64_t rz = tz_env > 0 ? tz_env : (768 + B - 1) / B;
    if (tz_env <= 0 && rz > 8) rz = 8;
    const int64_t groups = 256 >> pl.tpr_log2;
    const int64_t rz_max = (T + 4 * groups - 1) / (4 * groups);
    if (rz > rz_max) rz = rz_max;
    if (rz < 1) rz = 1;
    const dim3 rgrid(static_cast<unsigned>(rz), static_cast<unsigned>(B));
#define DALM_POOL_BWD_ROWS(TT, N, S) \
    hipLaunchKernelGGL((pool_bwd_rows_kernel<TT, N, S>), rgrid, dim3(256), 0, s, d_emb, emb, norm, inv_count, mask, \
                       static_cast<int>(T), static_cast<int>(D), normalize, pl.tpr_log2, static_cast<TT*>(dh))
#define DALM_POOL_BWD_ROWS_N(TT, S) \
    switch (pl.nch) { case 1: DALM_POOL_BWD_ROWS(TT, 1, S); break; case 2: DALM_POOL_BWD_ROWS(TT, 2, S); break; \
                      case 3: DALM_POOL_BWD_ROWS(TT, 3, S); break; default: DALM_POOL_BWD_ROWS(TT, 4, S); break; }
    if (dtype == DALM_F32) { if (nts) { DALM_POOL_BWD_ROWS_N(float, true) } else { DALM_POOL_BWD_ROWS_N(float, false) } }
    else { if (nts) { DALM_POOL_BWD_ROWS_N(bf16_t, true) } else { DALM_POOL_BWD_ROWS_N(bf16_t, false) } }
#undef DALM_POOL_BWD_ROWS_N
#undef DALM_POOL_BWD_ROWS
    return check_launch(__func__);
  }
  const dim3 grid(static_cast<unsigned>(dc), static_cast<unsigned>(B), static_cast<unsigned>(tz));
#define DALM_POOL_BWD(TT, N) \
  hipLaunchKernelGGL((pool_bwd_kernel<TT, N>), grid, dim3(256), 0, s, d_emb, emb, norm, inv_count, mask, \
                     static_cast<int>(T), static_cast<int>(D), normalize, vok, static_cast<TT*>(dh))
  if (dtype == DALM_F32) { if (nts) DALM_POOL_BWD(float, true); else DALM_POOL_BWD(float, false); }
  else { if (nts) DALM_POOL_BWD(bf16_t, true); else DALM_POOL_BWD(bf16_t, false); }
#undef DALM_POOL_BWD
  return check_launch(__func__);
}
