// dalm_rope_qk / dalm_swiglu_*: the two elementwise chains of a Llama-family generator layer that the step spends the most
// launches on, each as ONE streaming kernel per direction (gfx950).
//
// The reference runs its generator through transformers (dalm/models/rag_e2e_base_model.py:104-106 ->
// transformers/models/llama/modeling_llama.py): apply_rotary_pos_emb is  q*cos + rotate_half(q)*sin  for q and k
// (8 eager launches forward, ~14 backward per layer) and LlamaMLP is  down(silu(gate(x)) * up(x))  (2 forward, 4 backward,
// plus a saved [tokens, intermediate] activation).  Both are pure HBM streams; measured in the cfg3 step they were ~12 ms +
// ~5 ms of 175 ms (profiles/history/r04_bench_step_kernel_stats.txt: roll / addcmul / MulFunctor / silu / silu_backward rows).
//
// Rounding contract: torch evaluates every eager elementwise op in f32 and rounds its result to the tensor dtype.  These
// kernels round at exactly the same points (rb() below: a bf16 round trip for bf16 tensors, the identity for f32; separate
// mul / add, never a fused multiply-add), so forward AND backward results are the values the eager chain - and autograd's
// backward of it - produce, not merely close to them.
//   rope forward :  o1 = rb(rb(x1 c1) - rb(x2 s1))          o2 = rb(rb(x2 c2) + rb(x1 s2))        (halves 1 | 2 of head_dim)
//   rope backward:  d1 = rb(rb(g1 c1) + rb(g2 s2))          d2 = rb(rb(g2 c2) - rb(g1 s1))
//   swiglu forward :  s = rb(g / (1 + exp(-g)));  a = rb(s u)
//   swiglu backward:  ds = rb(da u);  du = rb(da s);  sig = 1 / (1 + exp(-g));  dg = rb(ds sig (1 + g (1 - sig)))
// Algorithmic bytes: rope 2 * (|q| + |k|) * el (+ cos/sin, shared by all heads); swiglu forward 3 * n * el, backward 5 * n * el.
#include "common.hpp"

namespace dalm {
namespace {

struct bf16_t { unsigned short v; };

template <typename T> struct EV;
template <> struct EV<float> {
  static constexpr int VEC = 4;
  __device__ static __forceinline__ float rb(float x) { return x; }
  __device__ static __forceinline__ void load(const float* p, float (&x)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
  }
  __device__ static __forceinline__ void store(float* p, const float (&x)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
  }
  __device__ static __forceinline__ float load1(const float* p) { return *p; }
  __device__ static __forceinline__ void store1(float* p, float x) { *p = x; }
};
template <> struct EV<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float rb(float x) { return bf16_to_f32(f32_to_bf16(x)); }
  __device__ static __forceinline__ void load(const bf16_t* p, float (&x)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[2 * i] = __uint_as_float(w[i] << 16);
      x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float (&x)[8]) {   // x already on the bf16 grid
    uint4 o;
    o.x = pack_bf16x2(x[0], x[1]); o.y = pack_bf16x2(x[2], x[3]);
    o.z = pack_bf16x2(x[4], x[5]); o.w = pack_bf16x2(x[6], x[7]);
    *reinterpret_cast<uint4*>(p) = o;
  }
  __device__ static __forceinline__ float load1(const bf16_t* p) { return bf16_to_f32(p->v); }
  __device__ static __forceinline__ void store1(bf16_t* p, float x) { p->v = f32_to_bf16(x); }
};

// ---------------------------------------------------------------------------------------------------
// rotary embedding of q and k in one launch.  Tensors are [B, H, T, hd] VIEWS with arbitrary (b, h, t) strides and a
// contiguous last dimension (transformers hands over transposed views of the [B, T, H*hd] projections); cos / sin are
// [B, T, hd] with (b, t) strides.  A row = one (b, t, head); rows are walked head-fastest, the memory order of those views.
// ---------------------------------------------------------------------------------------------------
struct RopeTensor { const void* x; void* o; int64_t xs[3]; int64_t os[3]; };   // strides of (b, h, t) in elements
struct RopeParams {
  RopeTensor q, k;
  const void* cos; const void* sin; int64_t cs[2];   // strides of (b, t)
  int B, T, Hq, Hk, hd, backward;
};

template <typename T, bool VECTOR>
__global__ __launch_bounds__(256) void rope_qk_kernel(const RopeParams p) {
#pragma clang fp contract(off)   // the eager chain is separate mul and add kernels: a fused multiply-add would round once less (f32)
  constexpr int VEC = VECTOR ? EV<T>::VEC : 1;
  const int h = p.hd >> 1;
  const int tpr = h / VEC;                       // threads per row
  const int64_t gid = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t row = gid / tpr;
  const int c = static_cast<int>(gid - row * tpr) * VEC;
  const int H = p.Hq + p.Hk;
  const int64_t rows = static_cast<int64_t>(p.B) * p.T * H;
  if (row >= rows) return;
  const int hh = static_cast<int>(row % H);
  const int64_t bt = row / H;
  const int t = static_cast<int>(bt % p.T), b = static_cast<int>(bt / p.T);
  const bool is_k = hh >= p.Hq;
  const RopeTensor& R = is_k ? p.k : p.q;
  const int head = is_k ? hh - p.Hq : hh;
  const T* x = static_cast<const T*>(R.x) + b * R.xs[0] + head * R.xs[1] + t * R.xs[2] + c;
  T* o = static_cast<T*>(R.o) + b * R.os[0] + head * R.os[1] + t * R.os[2] + c;
  const T* cs = static_cast<const T*>(p.cos) + b * p.cs[0] + t * p.cs[1] + c;
  const T* sn = static_cast<const T*>(p.sin) + b * p.cs[0] + t * p.cs[1] + c;
  float x1[VEC], x2[VEC], c1[VEC], c2[VEC], s1[VEC], s2[VEC], o1[VEC], o2[VEC];
  if constexpr (VECTOR) {
    EV<T>::load(x, x1); EV<T>::load(x + h, x2);
    EV<T>::load(cs, c1); EV<T>::load(cs + h, c2);
    EV<T>::load(sn, s1); EV<T>::load(sn + h, s2);
  } else {
    x1[0] = EV<T>::load1(x); x2[0] = EV<T>::load1(x + h);
    c1[0] = EV<T>::load1(cs); c2[0] = EV<T>::load1(cs + h);
    s1[0] = EV<T>::load1(sn); s2[0] = EV<T>::load1(sn + h);
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    // plain operators on purpose: the contract(off) pragma above governs THIS function's operations (the __f*_rn helpers are
    // inlined from a header compiled with contraction allowed and would fuse)
    const float a1 = EV<T>::rb(x1[e] * c1[e]), a2 = EV<T>::rb(x2[e] * c2[e]);
    if (!p.backward) {
      const float b1 = EV<T>::rb(x2[e] * s1[e]), b2 = EV<T>::rb(x1[e] * s2[e]);
      o1[e] = EV<T>::rb(a1 - b1);
      o2[e] = EV<T>::rb(a2 + b2);
    } else {
      const float b1 = EV<T>::rb(x2[e] * s2[e]), b2 = EV<T>::rb(x1[e] * s1[e]);
      o1[e] = EV<T>::rb(a1 + b1);
      o2[e] = EV<T>::rb(a2 - b2);
    }
  }
  if constexpr (VECTOR) {
    EV<T>::store(o, o1); EV<T>::store(o + h, o2);
  } else {
    EV<T>::store1(o, o1[0]); EV<T>::store1(o + h, o2[0]);
  }
}

// ---------------------------------------------------------------------------------------------------
// SwiGLU over n contiguous elements.  A workgroup walks STEPS tiles of 256*VEC elements with every load issued before the
// first store (single-tile workgroups are bound by workgroup dispatch at this size: profiles/history/r04_nf4_steps.txt).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + expf(-x)); }

// [R, C] views with a row stride (gate and up as the two halves of ONE [R, 2C] GEMM output, models/frozen_linear.py): element
// e of the logical [R, C] matrix lives at (e / C) * ld + e % C; C is a multiple of VEC, so a vector never crosses a row
struct SwigluLd { int64_t C, g, u, a, dg, du; };
// element index -> (row, column): 32-bit arithmetic (the entry points require R C < 2^32; a 64-bit division per vector made the
// first form of these kernels run at 0.47 of HBM where the contiguous ones reach 0.66)
struct RowCol { unsigned int row, col; };
__device__ __forceinline__ RowCol row_col(int64_t e, int64_t C) {
  const unsigned int ee = static_cast<unsigned int>(e), cc = static_cast<unsigned int>(C), r = ee / cc;
  return {r, ee - r * cc};
}
__device__ __forceinline__ int64_t ld_at(const RowCol& rc, int64_t ld) { return static_cast<int64_t>(rc.row) * ld + rc.col; }
// the tile `step` elements further on: no division (ONE per thread for its first tile; the silu itself is a dozen f32 operations
// per element, eight divisions per thread cost the first form a third of its rate)
__device__ __forceinline__ RowCol advance(RowCol rc, unsigned int step, unsigned int C) {
  rc.col += step;
  while (rc.col >= C) { rc.col -= C; ++rc.row; }
  return rc;
}

template <typename T, int STEPS>
__global__ __launch_bounds__(256) void swiglu_fwd_2d_kernel(const T* __restrict__ g, const T* __restrict__ u, T* __restrict__ a,
                                                            int64_t n, SwigluLd ld) {
  constexpr int VEC = EV<T>::VEC;
  const int64_t base = (static_cast<int64_t>(blockIdx.x) * STEPS * 256 + threadIdx.x) * VEC;
  float gv[STEPS][VEC], uv[STEPS][VEC];
  RowCol rc[STEPS];
  rc[0] = row_col(base < n ? base : 0, ld.C);
#pragma unroll
  for (int k = 1; k < STEPS; ++k) rc[k] = advance(rc[k - 1], 256 * VEC, static_cast<unsigned int>(ld.C));
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    const int64_t e0 = base + static_cast<int64_t>(k) * 256 * VEC;
    if (e0 < n) {
      EV<T>::load(g + ld_at(rc[k], ld.g), gv[k]);
      EV<T>::load(u + ld_at(rc[k], ld.u), uv[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    const int64_t e0 = base + static_cast<int64_t>(k) * 256 * VEC;
    if (e0 >= n) continue;
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = EV<T>::rb(__fmul_rn(EV<T>::rb(silu_f32(gv[k][e])), uv[k][e]));
    EV<T>::store(a + ld_at(rc[k], ld.a), o);
  }
}

template <typename T, int STEPS>
__global__ __launch_bounds__(256) void swiglu_bwd_2d_kernel(const T* __restrict__ da, const T* __restrict__ g,
                                                            const T* __restrict__ u, T* __restrict__ dg, T* __restrict__ du,
                                                            int64_t n, SwigluLd ld) {
  constexpr int VEC = EV<T>::VEC;
  const int64_t base = (static_cast<int64_t>(blockIdx.x) * STEPS * 256 + threadIdx.x) * VEC;
  float av[STEPS][VEC], gv[STEPS][VEC], uv[STEPS][VEC];
  RowCol rc[STEPS];
  rc[0] = row_col(base < n ? base : 0, ld.C);
#pragma unroll
  for (int k = 1; k < STEPS; ++k) rc[k] = advance(rc[k - 1], 256 * VEC, static_cast<unsigned int>(ld.C));
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    const int64_t e0 = base + static_cast<int64_t>(k) * 256 * VEC;
    if (e0 < n) {
      EV<T>::load(da + ld_at(rc[k], ld.a), av[k]);
      EV<T>::load(g + ld_at(rc[k], ld.g), gv[k]);
      EV<T>::load(u + ld_at(rc[k], ld.u), uv[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    const int64_t e0 = base + static_cast<int64_t>(k) * 256 * VEC;
    if (e0 >= n) continue;
    float og[VEC], ou[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {                                  // the contiguous kernel's arithmetic, statement for statement
      const float x = gv[k][e];
      const float s = EV<T>::rb(silu_f32(x));
      const float ds = EV<T>::rb(__fmul_rn(av[k][e], uv[k][e]));
      ou[e] = EV<T>::rb(__fmul_rn(av[k][e], s));
      const float sig = 1.0f / (1.0f + expf(-x));
      og[e] = EV<T>::rb(ds * sig * (1.0f + x * (1.0f - sig)));
    }
    EV<T>::store(dg + ld_at(rc[k], ld.dg), og);
    EV<T>::store(du + ld_at(rc[k], ld.du), ou);
  }
}

template <typename T, int STEPS>
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const T* __restrict__ g, const T* __restrict__ u, T* __restrict__ a,
                                                         int64_t n) {
  constexpr int VEC = EV<T>::VEC;
  const int64_t base = (static_cast<int64_t>(blockIdx.x) * STEPS * 256 + threadIdx.x) * VEC;
  float gv[STEPS][VEC], uv[STEPS][VEC];
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    const int64_t e0 = base + static_cast<int64_t>(k) * 256 * VEC;
    if (e0 + VEC <= n) { EV<T>::load(g + e0, gv[k]); EV<T>::load(u + e0, uv[k]); }
    else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        gv[k][e] = e0 + e < n ? EV<T>::load1(g + e0 + e) : 0.f;
        uv[k][e] = e0 + e < n ? EV<T>::load1(u + e0 + e) : 0.f;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    const int64_t e0 = base + static_cast<int64_t>(k) * 256 * VEC;
    if (e0 >= n) continue;
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = EV<T>::rb(__fmul_rn(EV<T>::rb(silu_f32(gv[k][e])), uv[k][e]));
    if (e0 + VEC <= n) EV<T>::store(a + e0, o);
    else
      for (int e = 0; e < VEC && e0 + e < n; ++e) EV<T>::store1(a + e0 + e, o[e]);
  }
}

template <typename T, int STEPS>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const T* __restrict__ da, const T* __restrict__ g,
                                                         const T* __restrict__ u, T* __restrict__ dg, T* __restrict__ du,
                                                         int64_t n) {
  constexpr int VEC = EV<T>::VEC;
  const int64_t base = (static_cast<int64_t>(blockIdx.x) * STEPS * 256 + threadIdx.x) * VEC;
  float av[STEPS][VEC], gv[STEPS][VEC], uv[STEPS][VEC];
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    const int64_t e0 = base + static_cast<int64_t>(k) * 256 * VEC;
    if (e0 + VEC <= n) { EV<T>::load(da + e0, av[k]); EV<T>::load(g + e0, gv[k]); EV<T>::load(u + e0, uv[k]); }
    else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const bool ok = e0 + e < n;
        av[k][e] = ok ? EV<T>::load1(da + e0 + e) : 0.f;
        gv[k][e] = ok ? EV<T>::load1(g + e0 + e) : 0.f;
        uv[k][e] = ok ? EV<T>::load1(u + e0 + e) : 0.f;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    const int64_t e0 = base + static_cast<int64_t>(k) * 256 * VEC;
    if (e0 >= n) continue;
    float og[VEC], ou[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float x = gv[k][e];
      const float s = EV<T>::rb(silu_f32(x));                       // the activation the eager chain saved
      const float ds = EV<T>::rb(__fmul_rn(av[k][e], uv[k][e]));
      ou[e] = EV<T>::rb(__fmul_rn(av[k][e], s));
      const float sig = 1.0f / (1.0f + expf(-x));
      og[e] = EV<T>::rb(ds * sig * (1.0f + x * (1.0f - sig)));      // torch's silu_backward expression, contraction as compiled
    }
    if (e0 + VEC <= n) { EV<T>::store(dg + e0, og); EV<T>::store(du + e0, ou); }
    else
      for (int e = 0; e < VEC && e0 + e < n; ++e) { EV<T>::store1(dg + e0 + e, og[e]); EV<T>::store1(du + e0 + e, ou[e]); }
  }
}

// ---------------------------------------------------------------------------------------------------
// RMSNorm with the residual add in front of it, forward and backward, one wave per row (the row stays in registers between
// the reduction and the scaling: every byte moves once).
//   forward : h = rb(res + delta)   [ADD; h = x otherwise]     rstd = rsqrt(mean(h^2) + eps)     y = rb(w * rb(h * rstd))
//             - the two roundings of transformers' LlamaRMSNorm (hidden.to(input_dtype), then the product with the weight)
//   backward: g = dy * w;  xh = h * rstd;  dx = rstd * (g - xh * mean(g * xh))  [+ dres: the gradient that reaches h through
//             the residual path, ADD]  - f32 throughout, rounded once
// torch's own kernels for the same work in the cfg3 step: add 17.8 us + rms_norm 21 us forward, layer_norm_grad_input 60 us +
// add 17 us backward per norm ([4608, 4096] bf16).  Algorithmic bytes: forward (2 or 4) * R * D * el, backward (3 or 4) * R * D * el.
// ---------------------------------------------------------------------------------------------------
template <typename T> struct RowVec;            // 16 bytes per lane and chunk
template <> struct RowVec<float> { static constexpr int N = 4; };
template <> struct RowVec<bf16_t> { static constexpr int N = 8; };

template <typename T, int NCH, bool ADD>
__global__ __launch_bounds__(256) void rms_norm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ delta,
                                                           const T* __restrict__ w, T* __restrict__ h_out, T* __restrict__ y,
                                                           float* __restrict__ rstd_out, int R, int D, float eps) {
  constexpr int N = RowVec<T>::N;
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int64_t base = static_cast<int64_t>(row) * D;
  float v[NCH][N];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * N;
    if (d < D) {
      EV<T>::load(x + base + d, v[c]);
      if constexpr (ADD) {
        float dl[N];
        EV<T>::load(delta + base + d, dl);
#pragma unroll
        for (int e = 0; e < N; ++e) v[c][e] = EV<T>::rb(v[c][e] + dl[e]);
        EV<T>::store(h_out + base + d, v[c]);
      }
#pragma unroll
      for (int e = 0; e < N; ++e) ss = fmaf(v[c][e], v[c][e], ss);
    }
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / static_cast<float>(D) + eps);
  if (lane == 0) rstd_out[row] = rstd;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * N;
    if (d < D) {
      float wv[N], o[N];
      EV<T>::load(w + d, wv);
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] = EV<T>::rb(wv[e] * EV<T>::rb(v[c][e] * rstd));
      EV<T>::store(y + base + d, o);
    }
  }
}

template <typename T, int NCH, bool ADD>
__global__ __launch_bounds__(256) void rms_norm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ h,
                                                           const T* __restrict__ w, const float* __restrict__ rstd_in,
                                                           const T* __restrict__ dres, T* __restrict__ dx, int R, int D) {
  constexpr int N = RowVec<T>::N;
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int64_t base = static_cast<int64_t>(row) * D;
  const float rstd = rstd_in[row];
  float g[NCH][N], xh[NCH][N];
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * N;
    if (d < D) {
      float wv[N];
      EV<T>::load(dy + base + d, g[c]);
      EV<T>::load(h + base + d, xh[c]);
      EV<T>::load(w + d, wv);
#pragma unroll
      for (int e = 0; e < N; ++e) {
        g[c][e] *= wv[e];
        xh[c][e] *= rstd;
        dot = fmaf(g[c][e], xh[c][e], dot);
      }
    }
  }
  dot = wave_sum(dot) / static_cast<float>(D);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * N;
    if (d < D) {
      float o[N];
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] = rstd * (g[c][e] - xh[c][e] * dot);
      if constexpr (ADD) {
        float r[N];
        EV<T>::load(dres + base + d, r);
#pragma unroll
        for (int e = 0; e < N; ++e) o[e] += r[e];
      }
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] = EV<T>::rb(o[e]);
      EV<T>::store(dx + base + d, o);
    }
  }
}

// Second form (bf16; the default since round 5, DALM_RMS_BWD_V2=0 selects the first): the three streams of a row (dy, h, and
// the residual-path gradient) are requested as raw 16-byte chunks BEFORE the wave reduction - 3 NCH loads in flight per lane
// instead of 2 NCH, then NCH after the reduction - and decoded twice (raw data in registers instead of f32 copies).
// [4608, 4096] bf16, 151 MB: 26.2 -> 23.4 us (profiles/r05_tower_attempts.txt).
template <int NCH, bool ADD>
__global__ __launch_bounds__(256) void rms_norm_bwd_v2_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ h,
                                                              const bf16_t* __restrict__ w, const float* __restrict__ rstd_in,
                                                              const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, int R, int D) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int64_t base = static_cast<int64_t>(row) * D;
  uint4 rg[NCH], rh[NCH], rr[NCH], rw[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * 8;
    const bool ok = d < D;
    rg[c] = ok ? *reinterpret_cast<const uint4*>(dy + base + d) : make_uint4(0u, 0u, 0u, 0u);
    rh[c] = ok ? *reinterpret_cast<const uint4*>(h + base + d) : make_uint4(0u, 0u, 0u, 0u);
    if constexpr (ADD) rr[c] = ok ? *reinterpret_cast<const uint4*>(dres + base + d) : make_uint4(0u, 0u, 0u, 0u);
    rw[c] = ok ? *reinterpret_cast<const uint4*>(w + d) : make_uint4(0u, 0u, 0u, 0u);
  }
  const float rstd = rstd_in[row];
  auto dec = [](const uint4& v, float (&x)[8]) {
    const unsigned int q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[2 * i] = __uint_as_float(q[i] << 16); x[2 * i + 1] = __uint_as_float(q[i] & 0xffff0000u); }
  };
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    float g[8], xh[8], wv[8];
    dec(rg[c], g); dec(rh[c], xh); dec(rw[c], wv);
#pragma unroll
    for (int e = 0; e < 8; ++e) dot = fmaf(g[e] * wv[e], xh[e] * rstd, dot);
  }
  dot = wave_sum(dot) / static_cast<float>(D);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * 8;
    if (d >= D) continue;
    float g[8], xh[8], wv[8], o[8];
    dec(rg[c], g); dec(rh[c], xh); dec(rw[c], wv);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = rstd * (g[e] * wv[e] - (xh[e] * rstd) * dot);
    if constexpr (ADD) {
      float r[8];
      dec(rr[c], r);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += r[e];
    }
    *reinterpret_cast<uint4*>(dx + base + d) = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]),
                                                           pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace dalm

using namespace dalm;

extern "C" int dalm_rope_qk(const void* q, const void* k, void* q_out, void* k_out, const void* cos, const void* sin, int dtype,
                            int64_t B, int64_t T, int64_t Hq, int64_t Hk, int64_t hd, const int64_t* q_strides,
                            const int64_t* k_strides, const int64_t* qo_strides, const int64_t* ko_strides,
                            const int64_t* cs_strides, int backward, dalm_stream_t stream) {
  DALM_REQUIRE(q && k && q_out && k_out && cos && sin && q_strides && k_strides && qo_strides && ko_strides && cs_strides,
               DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dtype == DALM_F32 || dtype == DALM_BF16, DALM_E_DTYPE, "dtype must be DALM_F32 or DALM_BF16");
  DALM_REQUIRE(B > 0 && T > 0 && Hq > 0 && Hk >= 0 && hd >= 2 && hd % 2 == 0 && hd <= 4096 && Hq + Hk <= 65535,
               DALM_E_SHAPE, "need B, T, Hq > 0, Hk >= 0, even 2 <= head_dim <= 4096");
  const int64_t rows = B * T * (Hq + Hk);
  DALM_REQUIRE(rows * (hd / 2) < (1ll << 40), DALM_E_SHAPE, "tensor too large for one launch");
  RopeParams p;
  p.q.x = q; p.q.o = q_out; p.k.x = k; p.k.o = k_out;
  const int vec = dtype == DALM_F32 ? 4 : 8;
  bool vector = (hd / 2) % vec == 0 && al16(q) && al16(k) && al16(q_out) && al16(k_out) && al16(cos) && al16(sin);
  for (int i = 0; i < 3; ++i) {
    p.q.xs[i] = q_strides[i]; p.k.xs[i] = k_strides[i]; p.q.os[i] = qo_strides[i]; p.k.os[i] = ko_strides[i];
    vector = vector && q_strides[i] % vec == 0 && k_strides[i] % vec == 0 && qo_strides[i] % vec == 0 && ko_strides[i] % vec == 0;
  }
  p.cos = cos; p.sin = sin; p.cs[0] = cs_strides[0]; p.cs[1] = cs_strides[1];
  vector = vector && cs_strides[0] % vec == 0 && cs_strides[1] % vec == 0;
  p.B = static_cast<int>(B); p.T = static_cast<int>(T); p.Hq = static_cast<int>(Hq); p.Hk = static_cast<int>(Hk);
  p.hd = static_cast<int>(hd); p.backward = backward != 0;
  const int64_t threads = rows * ((hd / 2) / (vector ? vec : 1));
  const int64_t blocks = (threads + 255) / 256;
  DALM_REQUIRE(blocks <= 0x7fffffffLL, DALM_E_SHAPE, "tensor too large for one launch");
  const dim3 grid(static_cast<unsigned>(blocks));
  hipStream_t s = as_stream(stream);
  if (dtype == DALM_F32) {
    if (vector) hipLaunchKernelGGL((rope_qk_kernel<float, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((rope_qk_kernel<float, false>), grid, dim3(256), 0, s, p);
  } else {
    if (vector) hipLaunchKernelGGL((rope_qk_kernel<bf16_t, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((rope_qk_kernel<bf16_t, false>), grid, dim3(256), 0, s, p);
  }
  return check_launch(__func__);
}

namespace {
// tiles of 256 * VEC elements per workgroup: 8 / 12 sixteen-byte loads in flight per lane (forward / backward).  8 tiles
// measured SLOWER at [4608, 11008] bf16 (forward 70.3 -> 76.2 us, backward 110.7 -> 123.7 us: profiles/r05_tower_attempts.txt)
constexpr int kSwigluSteps = 4;
inline int64_t swiglu_blocks(int64_t n, int vec) {
  const int64_t per = static_cast<int64_t>(kSwigluSteps) * 256 * vec;
  return (n + per - 1) / per;
}
}  // namespace

extern "C" int dalm_swiglu_fwd(const void* gate, const void* up, void* act, int dtype, int64_t n, dalm_stream_t stream) {
  DALM_REQUIRE(n >= 0, DALM_E_SHAPE, "n must be >= 0");
  if (n == 0) return 0;
  DALM_REQUIRE(gate && up && act, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dtype == DALM_F32 || dtype == DALM_BF16, DALM_E_DTYPE, "dtype must be DALM_F32 or DALM_BF16");
  DALM_REQUIRE(al16(gate) && al16(up) && al16(act), DALM_E_ALIGN, "gate / up / act must be 16-byte aligned");
  const int64_t blocks = swiglu_blocks(n, dtype == DALM_F32 ? 4 : 8);
  DALM_REQUIRE(blocks <= 0x7fffffffLL, DALM_E_SHAPE, "n too large for one launch");
  const dim3 grid(static_cast<unsigned>(blocks));
  if (dtype == DALM_F32)
    hipLaunchKernelGGL((swiglu_fwd_kernel<float, kSwigluSteps>), grid, dim3(256), 0, as_stream(stream),
                       static_cast<const float*>(gate), static_cast<const float*>(up), static_cast<float*>(act), n);
  else
    hipLaunchKernelGGL((swiglu_fwd_kernel<bf16_t, kSwigluSteps>), grid, dim3(256), 0, as_stream(stream),
                       static_cast<const bf16_t*>(gate), static_cast<const bf16_t*>(up), static_cast<bf16_t*>(act), n);
  return check_launch(__func__);
}

extern "C" int dalm_swiglu_bwd(const void* d_act, const void* gate, const void* up, void* d_gate, void* d_up, int dtype,
                               int64_t n, dalm_stream_t stream) {
  DALM_REQUIRE(n >= 0, DALM_E_SHAPE, "n must be >= 0");
  if (n == 0) return 0;
  DALM_REQUIRE(d_act && gate && up && d_gate && d_up, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dtype == DALM_F32 || dtype == DALM_BF16, DALM_E_DTYPE, "dtype must be DALM_F32 or DALM_BF16");
  DALM_REQUIRE(al16(d_act) && al16(gate) && al16(up) && al16(d_gate) && al16(d_up), DALM_E_ALIGN,
               "all tensors must be 16-byte aligned");
  const int64_t blocks = swiglu_blocks(n, dtype == DALM_F32 ? 4 : 8);
  DALM_REQUIRE(blocks <= 0x7fffffffLL, DALM_E_SHAPE, "n too large for one launch");
  const dim3 grid(static_cast<unsigned>(blocks));
  if (dtype == DALM_F32)
    hipLaunchKernelGGL((swiglu_bwd_kernel<float, kSwigluSteps>), grid, dim3(256), 0, as_stream(stream),
                       static_cast<const float*>(d_act), static_cast<const float*>(gate), static_cast<const float*>(up),
                       static_cast<float*>(d_gate), static_cast<float*>(d_up), n);
  else
    hipLaunchKernelGGL((swiglu_bwd_kernel<bf16_t, kSwigluSteps>), grid, dim3(256), 0, as_stream(stream),
                       static_cast<const bf16_t*>(d_act), static_cast<const bf16_t*>(gate), static_cast<const bf16_t*>(up),
                       static_cast<bf16_t*>(d_gate), static_cast<bf16_t*>(d_up), n);
  return check_launch(__func__);
}

extern "C" int dalm_swiglu_fwd_2d(const void* gate, const void* up, void* act, int dtype, int64_t R, int64_t C, int64_t ld_gate,
                                  int64_t ld_up, int64_t ld_act, dalm_stream_t stream) {
  DALM_REQUIRE(R >= 0 && C >= 0, DALM_E_SHAPE, "R, C must be >= 0");
  if (R == 0 || C == 0) return 0;
  DALM_REQUIRE(gate && up && act, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dtype == DALM_F32 || dtype == DALM_BF16, DALM_E_DTYPE, "dtype must be DALM_F32 or DALM_BF16");
  const int vec = dtype == DALM_F32 ? 4 : 8;
  DALM_REQUIRE(al16(gate) && al16(up) && al16(act) && C % vec == 0 && ld_gate % vec == 0 && ld_up % vec == 0 && ld_act % vec == 0
                   && ld_gate >= C && ld_up >= C && ld_act >= C,
               DALM_E_ALIGN, "16-byte aligned pointers, C and the row strides multiples of 16 bytes, strides >= C");
  const int64_t n = R * C, blocks = swiglu_blocks(n, vec);
  DALM_REQUIRE(n < (1ll << 32) && blocks <= 0x7fffffffLL, DALM_E_SHAPE, "tensor too large for one launch (R C must stay below 2^32)");
  const dim3 grid(static_cast<unsigned>(blocks));
  const SwigluLd ld = {C, ld_gate, ld_up, ld_act, 0, 0};
  if (dtype == DALM_F32)
    hipLaunchKernelGGL((swiglu_fwd_2d_kernel<float, kSwigluSteps>), grid, dim3(256), 0, as_stream(stream),
                       static_cast<const float*>(gate), static_cast<const float*>(up), static_cast<float*>(act), n, ld);
  else
    hipLaunchKernelGGL((swiglu_fwd_2d_kernel<bf16_t, kSwigluSteps>), grid, dim3(256), 0, as_stream(stream),
                       static_cast<const bf16_t*>(gate), static_cast<const bf16_t*>(up), static_cast<bf16_t*>(act), n, ld);
  return check_launch(__func__);
}

extern "C" int dalm_swiglu_bwd_2d(const void* d_act, const void* gate, const void* up, void* d_gate, void* d_up, int dtype, int64_t R,
                                  int64_t C, int64_t ld_dact, int64_t ld_gate, int64_t ld_up, int64_t ld_dgate, int64_t ld_dup,
                                  dalm_stream_t stream) {
  DALM_REQUIRE(R >= 0 && C >= 0, DALM_E_SHAPE, "R, C must be >= 0");
  if (R == 0 || C == 0) return 0;
  DALM_REQUIRE(d_act && gate && up && d_gate && d_up, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dtype == DALM_F32 || dtype == DALM_BF16, DALM_E_DTYPE, "dtype must be DALM_F32 or DALM_BF16");
  const int vec = dtype == DALM_F32 ? 4 : 8;
  bool ok = al16(d_act) && al16(gate) && al16(up) && al16(d_gate) && al16(d_up) && C % vec == 0;
  for (int64_t l : {ld_dact, ld_gate, ld_up, ld_dgate, ld_dup}) ok = ok && l % vec == 0 && l >= C;
  DALM_REQUIRE(ok, DALM_E_ALIGN, "16-byte aligned pointers, C and the row strides multiples of 16 bytes, strides >= C");
  const int64_t n = R * C, blocks = swiglu_blocks(n, vec);
  DALM_REQUIRE(n < (1ll << 32) && blocks <= 0x7fffffffLL, DALM_E_SHAPE, "tensor too large for one launch (R C must stay below 2^32)");
  const dim3 grid(static_cast<unsigned>(blocks));
  const SwigluLd ld = {C, ld_gate, ld_up, ld_dact, ld_dgate, ld_dup};
  if (dtype == DALM_F32)
    hipLaunchKernelGGL((swiglu_bwd_2d_kernel<float, kSwigluSteps>), grid, dim3(256), 0, as_stream(stream),
                       static_cast<const float*>(d_act), static_cast<const float*>(gate), static_cast<const float*>(up),
                       static_cast<float*>(d_gate), static_cast<float*>(d_up), n, ld);
  else
    hipLaunchKernelGGL((swiglu_bwd_2d_kernel<bf16_t, kSwigluSteps>), grid, dim3(256), 0, as_stream(stream),
                       static_cast<const bf16_t*>(d_act), static_cast<const bf16_t*>(gate), static_cast<const bf16_t*>(up),
                       static_cast<bf16_t*>(d_gate), static_cast<bf16_t*>(d_up), n, ld);
  return check_launch(__func__);
}

#define DALM_RMS_CHECKS                                                                                                  \
  DALM_REQUIRE(dtype == DALM_F32 || dtype == DALM_BF16, DALM_E_DTYPE, "dtype must be DALM_F32 or DALM_BF16");             \
  const int vecn = dtype == DALM_F32 ? 4 : 8;                                                                            \
  DALM_REQUIRE(R > 0 && D > 0 && D % vecn == 0 && D <= 64ll * vecn * 16 && R <= 0x7ffffff0ll, DALM_E_SHAPE,               \
               "need rows > 0 and a width that is a multiple of 16 bytes, at most 8192 (bf16) / 4096 (f32) elements");    \
  const int nch = static_cast<int>((D + 64 * vecn - 1) / (64 * vecn));                                                   \
  const dim3 grid(static_cast<unsigned>((R + 3) / 4))

#define DALM_RMS_DISPATCH(KERNEL, TT, ADDV, ...)                                                                         \
  do {                                                                                                                   \
    if (nch <= 1) hipLaunchKernelGGL((KERNEL<TT, 1, ADDV>), grid, dim3(256), 0, as_stream(stream), __VA_ARGS__);         \
    else if (nch <= 2) hipLaunchKernelGGL((KERNEL<TT, 2, ADDV>), grid, dim3(256), 0, as_stream(stream), __VA_ARGS__);    \
    else if (nch <= 4) hipLaunchKernelGGL((KERNEL<TT, 4, ADDV>), grid, dim3(256), 0, as_stream(stream), __VA_ARGS__);    \
    else if (nch <= 8) hipLaunchKernelGGL((KERNEL<TT, 8, ADDV>), grid, dim3(256), 0, as_stream(stream), __VA_ARGS__);    \
    else hipLaunchKernelGGL((KERNEL<TT, 16, ADDV>), grid, dim3(256), 0, as_stream(stream), __VA_ARGS__);                 \
  } while (0)

extern "C" int dalm_rms_norm_fwd(const void* x, const void* delta, const void* w, int dtype, int64_t R, int64_t D, float eps,
                                 void* h_out, void* y, float* rstd, dalm_stream_t stream) {
  DALM_REQUIRE(x && w && y && rstd, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE((delta == nullptr) == (h_out == nullptr), DALM_E_NULL, "delta and h_out go together");
  DALM_RMS_CHECKS;
  DALM_REQUIRE(al16(x) && al16(w) && al16(y) && al16(delta) && al16(h_out), DALM_E_ALIGN, "tensors must be 16-byte aligned");
  const int Ri = static_cast<int>(R), Di = static_cast<int>(D);
  if (dtype == DALM_F32) {
    if (delta) DALM_RMS_DISPATCH(rms_norm_fwd_kernel, float, true, static_cast<const float*>(x), static_cast<const float*>(delta),
                                 static_cast<const float*>(w), static_cast<float*>(h_out), static_cast<float*>(y), rstd, Ri, Di, eps);
    else DALM_RMS_DISPATCH(rms_norm_fwd_kernel, float, false, static_cast<const float*>(x), static_cast<const float*>(nullptr),
                           static_cast<const float*>(w), static_cast<float*>(nullptr), static_cast<float*>(y), rstd, Ri, Di, eps);
  } else {
    if (delta) DALM_RMS_DISPATCH(rms_norm_fwd_kernel, bf16_t, true, static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(delta),
                                 static_cast<const bf16_t*>(w), static_cast<bf16_t*>(h_out), static_cast<bf16_t*>(y), rstd, Ri, Di, eps);
    else DALM_RMS_DISPATCH(rms_norm_fwd_kernel, bf16_t, false, static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(nullptr),
                           static_cast<const bf16_t*>(w), static_cast<bf16_t*>(nullptr), static_cast<bf16_t*>(y), rstd, Ri, Di, eps);
  }
  return check_launch(__func__);
}

extern "C" int dalm_rms_norm_bwd(const void* dy, const void* h, const void* w, const float* rstd, const void* dres, int dtype,
                                 int64_t R, int64_t D, void* dx, dalm_stream_t stream) {
  DALM_REQUIRE(dy && h && w && rstd && dx, DALM_E_NULL, "null pointer argument");
  DALM_RMS_CHECKS;
  DALM_REQUIRE(al16(dy) && al16(h) && al16(w) && al16(dx) && al16(dres), DALM_E_ALIGN, "tensors must be 16-byte aligned");
  const int Ri = static_cast<int>(R), Di = static_cast<int>(D);
  if (dtype == DALM_F32) {
    if (dres) DALM_RMS_DISPATCH(rms_norm_bwd_kernel, float, true, static_cast<const float*>(dy), static_cast<const float*>(h),
                                static_cast<const float*>(w), rstd, static_cast<const float*>(dres), static_cast<float*>(dx), Ri, Di);
    else DALM_RMS_DISPATCH(rms_norm_bwd_kernel, float, false, static_cast<const float*>(dy), static_cast<const float*>(h),
                           static_cast<const float*>(w), rstd, static_cast<const float*>(nullptr), static_cast<float*>(dx), Ri, Di);
  } else if (static const bool v2 = [] { const char* e = getenv("DALM_RMS_BWD_V2"); return !e || atoi(e) != 0; }(); v2 && nch <= 8) {
#define DALM_RMS_V2(ADDV, DRES)                                                                                           \
    do {                                                                                                                  \
      if (nch <= 1) hipLaunchKernelGGL((rms_norm_bwd_v2_kernel<1, ADDV>), grid, dim3(256), 0, as_stream(stream), static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(h), static_cast<const bf16_t*>(w), rstd, DRES, static_cast<bf16_t*>(dx), Ri, Di); \
      else if (nch <= 2) hipLaunchKernelGGL((rms_norm_bwd_v2_kernel<2, ADDV>), grid, dim3(256), 0, as_stream(stream), static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(h), static_cast<const bf16_t*>(w), rstd, DRES, static_cast<bf16_t*>(dx), Ri, Di); \
      else if (nch <= 4) hipLaunchKernelGGL((rms_norm_bwd_v2_kernel<4, ADDV>), grid, dim3(256), 0, as_stream(stream), static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(h), static_cast<const bf16_t*>(w), rstd, DRES, static_cast<bf16_t*>(dx), Ri, Di); \
      else hipLaunchKernelGGL((rms_norm_bwd_v2_kernel<8, ADDV>), grid, dim3(256), 0, as_stream(stream), static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(h), static_cast<const bf16_t*>(w), rstd, DRES, static_cast<bf16_t*>(dx), Ri, Di); \
    } while (0)
    if (dres) DALM_RMS_V2(true, static_cast<const bf16_t*>(dres)); else DALM_RMS_V2(false, static_cast<const bf16_t*>(nullptr));
#undef DALM_RMS_V2
  } else {
    if (dres) DALM_RMS_DISPATCH(rms_norm_bwd_kernel, bf16_t, true, static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(h),
                                static_cast<const bf16_t*>(w), rstd, static_cast<const bf16_t*>(dres), static_cast<bf16_t*>(dx), Ri, Di);
    else DALM_RMS_DISPATCH(rms_norm_bwd_kernel, bf16_t, false, static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(h),
                           static_cast<const bf16_t*>(w), rstd, static_cast<const bf16_t*>(nullptr), static_cast<bf16_t*>(dx), Ri, Di);
  }
  return check_launch(__func__);
}
