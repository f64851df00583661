// Elementwise chains of a Falcon-7B decoder layer (BASELINE.json config 5) as one HIP launch each (gfx950):
//   LayerNorm (bias, f32 statistics, output in the activation dtype), exact-erf GELU, and the three-way residual add.
// transformers evaluates them as eager ops (modeling_falcon.py FalconDecoderLayer.forward / FalconMLP.forward / dropout_add; the
// reference reaches them through self.generator_model(...), dalm/models/rag_e2e_base_model.py:104-106).  Under bf16 autocast the
// LayerNorm alone is 2 up-casts, an f32 kernel and 2 down-casts per layer; per cfg5 step the layer's elementwise chains were
// ~27 ms of 182 (profiles/r05cfg5_step_by_stream.txt).  The kernels round where those chains round:
//   layer_norm : y = bf16( (x - mean) * rstd * w + b ) computed in f32 from the up-cast input - what torch's autocast LayerNorm
//                (f32) followed by the consumers' casts to bf16 produces, up to the summation order of the two means
//   gelu       : y = bf16( 0.5 x (1 + erf(x / sqrt 2)) ) in f32 (torch's bf16 GELU kernel computes in f32 and rounds once)
//   add3       : out = bf16( c + bf16(a + b) )   (mlp_output += attention_output; residual + out)
// One wave per row for the norm (the row stays in registers between the reductions and the scaling), flat 16-byte streams for
// the other two.  Algorithmic bytes: norm fwd 2 R D el, bwd 3-4 R D el; gelu fwd 2 n el, bwd 3 n el; add3 4 n el.
#include "common.hpp"

namespace dalm {
namespace {

struct bf16_t { unsigned short v; };

__device__ __forceinline__ void dec8(const uint4& v, float (&x)[8]) {
  const unsigned int q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { x[2 * i] = __uint_as_float(q[i] << 16); x[2 * i + 1] = __uint_as_float(q[i] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 enc8(const float (&o)[8]) {
  return make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
}
__device__ __forceinline__ float rb(float x) { return bf16_to_f32(f32_to_bf16(x)); }

// ---- LayerNorm, bf16 rows of D elements (D % 8 == 0, D <= 8192), one wave per row, 4 rows per workgroup ----
template <int NCH>
__global__ __launch_bounds__(256) void layer_norm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                             const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out, int R,
                                                             int D, float eps) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int64_t base = static_cast<int64_t>(row) * D;
  uint4 rx[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * 8;
    rx[c] = d < D ? *reinterpret_cast<const uint4*>(x + base + d) : make_uint4(0u, 0u, 0u, 0u);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    float v[8];
    dec8(rx[c], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
  }
  const float mean = wave_sum(s) / static_cast<float>(D);
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * 8;
    if (d >= D) continue;
    float v[8];
    dec8(rx[c], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf(v[e] - mean, v[e] - mean, ss);
  }
  const float rstd = rsqrtf(wave_sum(ss) / static_cast<float>(D) + eps);
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
  for (int c = 0; c < NCH; ++c) asm volatile("" : "+v"(rx[c].x), "+v"(rx[c].y), "+v"(rx[c].z), "+v"(rx[c].w));   // decode again
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * 8;
    if (d >= D) continue;
    float v[8], wv[8], bv[8], o[8];
    dec8(rx[c], v);
    dec8(*reinterpret_cast<const uint4*>(w + d), wv);
    if (b) dec8(*reinterpret_cast<const uint4*>(b + d), bv);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaf((v[e] - mean) * rstd, wv[e], b ? bv[e] : 0.f);
    *reinterpret_cast<uint4*>(y + base + d) = enc8(o);
  }
}

// dx = rstd * (g - mean(g) - xh * mean(g * xh)) [+ dres],  g = dy * w,  xh = (x - mean) * rstd
template <int NCH, bool ADD>
__global__ __launch_bounds__(256) void layer_norm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                             const bf16_t* __restrict__ w, const float* __restrict__ mean_in,
                                                             const float* __restrict__ rstd_in, const bf16_t* __restrict__ dres,
                                                             bf16_t* __restrict__ dx, int R, int D) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int64_t base = static_cast<int64_t>(row) * D;
  uint4 rg[NCH], rxx[NCH], rr[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * 8;
    const bool ok = d < D;
    rg[c] = ok ? *reinterpret_cast<const uint4*>(dy + base + d) : make_uint4(0u, 0u, 0u, 0u);
    rxx[c] = ok ? *reinterpret_cast<const uint4*>(x + base + d) : make_uint4(0u, 0u, 0u, 0u);
    if constexpr (ADD) rr[c] = ok ? *reinterpret_cast<const uint4*>(dres + base + d) : make_uint4(0u, 0u, 0u, 0u);
  }
  const float mean = mean_in[row], rstd = rstd_in[row];
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * 8;
    if (d >= D) continue;
    float g[8], xv[8], wv[8];
    dec8(rg[c], g); dec8(rxx[c], xv); dec8(*reinterpret_cast<const uint4*>(w + d), wv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float ge = g[e] * wv[e];
      sg += ge;
      sgx = fmaf(ge, (xv[e] - mean) * rstd, sgx);
    }
  }
  sg = wave_sum(sg) / static_cast<float>(D);
  sgx = wave_sum(sgx) / static_cast<float>(D);
  // the second pass decodes the packed registers again (shifts) and reads w again (L1): carrying the decoded f32 values of the
  // first pass across the reductions is 3 x 8 x NCH registers - 256 VGPRs + AGPR spills at NCH = 9 (Falcon-7B), one wave per SIMD
  const bf16_t* w2 = w;
  asm volatile("" : "+s"(w2));
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    asm volatile("" : "+v"(rg[c].x), "+v"(rg[c].y), "+v"(rg[c].z), "+v"(rg[c].w));
    asm volatile("" : "+v"(rxx[c].x), "+v"(rxx[c].y), "+v"(rxx[c].z), "+v"(rxx[c].w));
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * 8;
    if (d >= D) continue;
    float g[8], xv[8], wv[8], o[8];
    dec8(rg[c], g); dec8(rxx[c], xv); dec8(*reinterpret_cast<const uint4*>(w2 + d), wv);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = rstd * (g[e] * wv[e] - sg - (xv[e] - mean) * rstd * sgx);
    if constexpr (ADD) {
      float r[8];
      dec8(rr[c], r);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += r[e];
    }
    *reinterpret_cast<uint4*>(dx + base + d) = enc8(o);
  }
}

// ---- flat streams: 4 tiles of 256 x 8 elements per workgroup, every load issued before the first store ----
constexpr int kSteps = 4;
constexpr float kInvSqrt2 = 0.70710678118654752440f, kInvSqrt2Pi = 0.39894228040143267794f;

__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t n8) {
  const int64_t c0 = static_cast<int64_t>(blockIdx.x) * kSteps * 256 + threadIdx.x;
  uint4 rx[kSteps];
#pragma unroll
  for (int k = 0; k < kSteps; ++k) rx[k] = c0 + k * 256 < n8 ? *reinterpret_cast<const uint4*>(x + (c0 + k * 256) * 8) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int k = 0; k < kSteps; ++k) {
    if (c0 + k * 256 >= n8) continue;
    float v[8], o[8];
    dec8(rx[k], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.5f * v[e] * (1.0f + erff(v[e] * kInvSqrt2));
    *reinterpret_cast<uint4*>(y + (c0 + k * 256) * 8) = enc8(o);
  }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                       bf16_t* __restrict__ dx, int64_t n8) {
  const int64_t c0 = static_cast<int64_t>(blockIdx.x) * kSteps * 256 + threadIdx.x;
  uint4 rg[kSteps], rx[kSteps];
#pragma unroll
  for (int k = 0; k < kSteps; ++k) {
    const bool ok = c0 + k * 256 < n8;
    rg[k] = ok ? *reinterpret_cast<const uint4*>(dy + (c0 + k * 256) * 8) : make_uint4(0u, 0u, 0u, 0u);
    rx[k] = ok ? *reinterpret_cast<const uint4*>(x + (c0 + k * 256) * 8) : make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int k = 0; k < kSteps; ++k) {
    if (c0 + k * 256 >= n8) continue;
    float g[8], v[8], o[8];
    dec8(rg[k], g); dec8(rx[k], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float cdf = 0.5f * (1.0f + erff(v[e] * kInvSqrt2));
      const float pdf = __expf(-0.5f * v[e] * v[e]) * kInvSqrt2Pi;
      o[e] = g[e] * (cdf + v[e] * pdf);
    }
    *reinterpret_cast<uint4*>(dx + (c0 + k * 256) * 8) = enc8(o);
  }
}

__global__ __launch_bounds__(256) void add3_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                   const bf16_t* __restrict__ c, bf16_t* __restrict__ out, int64_t n8) {
  const int64_t c0 = static_cast<int64_t>(blockIdx.x) * kSteps * 256 + threadIdx.x;
  uint4 ra[kSteps], rbb[kSteps], rc[kSteps];
#pragma unroll
  for (int k = 0; k < kSteps; ++k) {
    const bool ok = c0 + k * 256 < n8;
    const int64_t off = (c0 + k * 256) * 8;
    ra[k] = ok ? *reinterpret_cast<const uint4*>(a + off) : make_uint4(0u, 0u, 0u, 0u);
    rbb[k] = ok ? *reinterpret_cast<const uint4*>(b + off) : make_uint4(0u, 0u, 0u, 0u);
    rc[k] = ok ? *reinterpret_cast<const uint4*>(c + off) : make_uint4(0u, 0u, 0u, 0u);
  }
#pragma unroll
  for (int k = 0; k < kSteps; ++k) {
    if (c0 + k * 256 >= n8) continue;
    float x[8], y[8], z[8], o[8];
    dec8(ra[k], x); dec8(rbb[k], y); dec8(rc[k], z);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = z[e] + rb(x[e] + y[e]);
    *reinterpret_cast<uint4*>(out + (c0 + k * 256) * 8) = enc8(o);
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace dalm

using namespace dalm;

// chunks of 64 lanes x 8 elements a wave holds per row: 9 is Falcon-7B's 4544 (8.875 chunks), 16 the 8192 limit
#define DALM_LN_DISPATCH(M)                                                                                                       \
  do {                                                                                                                            \
    if (nch <= 1) { M(1); } else if (nch <= 2) { M(2); } else if (nch <= 4) { M(4); } else if (nch <= 6) { M(6); }                 \
    else if (nch <= 8) { M(8); } else if (nch <= 9) { M(9); } else if (nch <= 12) { M(12); } else { M(16); }                       \
  } while (0)

extern "C" int dalm_layer_norm_fwd(const void* x, const void* w, const void* b, int64_t R, int64_t D, float eps, void* y,
                                   float* mean, float* rstd, dalm_stream_t stream) {
  DALM_REQUIRE(x && w && y && mean && rstd, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(R > 0 && D > 0 && D % 8 == 0 && D <= 8192 && R <= 0x7ffffff0ll, DALM_E_SHAPE, "need R > 0 and D a multiple of 8, at most 8192");
  DALM_REQUIRE(al16(x) && al16(w) && al16(y) && (!b || al16(b)), DALM_E_ALIGN, "tensors must be 16-byte aligned");
  const int nch = static_cast<int>((D + 511) / 512);
  const dim3 grid(static_cast<unsigned>((R + 3) / 4));
  hipStream_t s = as_stream(stream);
  const bf16_t* xp = static_cast<const bf16_t*>(x);
  const bf16_t* wp = static_cast<const bf16_t*>(w);
  const bf16_t* bp = static_cast<const bf16_t*>(b);
  bf16_t* yp = static_cast<bf16_t*>(y);
  const int Ri = static_cast<int>(R), Di = static_cast<int>(D);
#define DALM_LNF(N) hipLaunchKernelGGL((layer_norm_fwd_kernel<N>), grid, dim3(256), 0, s, xp, wp, bp, yp, mean, rstd, Ri, Di, eps)
  DALM_LN_DISPATCH(DALM_LNF);
#undef DALM_LNF
  return check_launch(__func__);
}

extern "C" int dalm_layer_norm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd,
                                   const void* dres, int64_t R, int64_t D, void* dx, dalm_stream_t stream) {
  DALM_REQUIRE(dy && x && w && mean && rstd && dx, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(R > 0 && D > 0 && D % 8 == 0 && D <= 8192 && R <= 0x7ffffff0ll, DALM_E_SHAPE, "need R > 0 and D a multiple of 8, at most 8192");
  DALM_REQUIRE(al16(dy) && al16(x) && al16(w) && al16(dx) && (!dres || al16(dres)), DALM_E_ALIGN, "tensors must be 16-byte aligned");
  const int nch = static_cast<int>((D + 511) / 512);
  const dim3 grid(static_cast<unsigned>((R + 3) / 4));
  hipStream_t s = as_stream(stream);
  const bf16_t* gp = static_cast<const bf16_t*>(dy);
  const bf16_t* xp = static_cast<const bf16_t*>(x);
  const bf16_t* wp = static_cast<const bf16_t*>(w);
  const bf16_t* rp = static_cast<const bf16_t*>(dres);
  bf16_t* op = static_cast<bf16_t*>(dx);
  const int Ri = static_cast<int>(R), Di = static_cast<int>(D);
#define DALM_LNB_T(N) hipLaunchKernelGGL((layer_norm_bwd_kernel<N, true>), grid, dim3(256), 0, s, gp, xp, wp, mean, rstd, rp, op, Ri, Di)
#define DALM_LNB_F(N) hipLaunchKernelGGL((layer_norm_bwd_kernel<N, false>), grid, dim3(256), 0, s, gp, xp, wp, mean, rstd, rp, op, Ri, Di)
  if (dres) DALM_LN_DISPATCH(DALM_LNB_T);
  else DALM_LN_DISPATCH(DALM_LNB_F);
#undef DALM_LNB_T
#undef DALM_LNB_F
  return check_launch(__func__);
}

#define DALM_FLAT_CHECKS(n)                                                                              \
  DALM_REQUIRE((n) >= 0 && (n) % 8 == 0, DALM_E_SHAPE, "element count must be a non-negative multiple of 8"); \
  if ((n) == 0) return 0;                                                                                \
  const int64_t n8 = (n) / 8;                                                                            \
  const int64_t blocks = (n8 + kSteps * 256 - 1) / (kSteps * 256);                                        \
  DALM_REQUIRE(blocks <= 0x7fffffffll, DALM_E_SHAPE, "too many elements for one launch")

extern "C" int dalm_gelu_fwd(const void* x, void* y, int64_t n, dalm_stream_t stream) {
  DALM_FLAT_CHECKS(n);
  DALM_REQUIRE(x && y, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(al16(x) && al16(y), DALM_E_ALIGN, "tensors must be 16-byte aligned");
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream),
                     static_cast<const bf16_t*>(x), static_cast<bf16_t*>(y), n8);
  return check_launch(__func__);
}

extern "C" int dalm_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, dalm_stream_t stream) {
  DALM_FLAT_CHECKS(n);
  DALM_REQUIRE(dy && x && dx, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(al16(dy) && al16(x) && al16(dx), DALM_E_ALIGN, "tensors must be 16-byte aligned");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream),
                     static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(x), static_cast<bf16_t*>(dx), n8);
  return check_launch(__func__);
}

extern "C" int dalm_add3(const void* a, const void* b, const void* c, void* out, int64_t n, dalm_stream_t stream) {
  DALM_FLAT_CHECKS(n);
  DALM_REQUIRE(a && b && c && out, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(al16(a) && al16(b) && al16(c) && al16(out), DALM_E_ALIGN, "tensors must be 16-byte aligned");
  hipLaunchKernelGGL(add3_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream),
                     static_cast<const bf16_t*>(a), static_cast<const bf16_t*>(b), static_cast<const bf16_t*>(c),
                     static_cast<bf16_t*>(out), n8);
  return check_launch(__func__);
}
