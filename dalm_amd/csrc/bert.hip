// BERT encoder layer: `dropout + residual add + LayerNorm` as ONE HIP launch per direction (gfx950).
// transformers' BertSelfOutput / BertOutput (modeling_bert.py) evaluate, after their dense projection,
//     hidden = dropout(hidden);  hidden = LayerNorm(hidden + input_tensor)
// as eager ops; the reference reaches them through self.retriever_model(...) (dalm/models/rag_e2e_base_model.py:84-93,
// retriever_only_base_model.py:43-64).  Under bf16 autocast - the configuration bench.py measures - that chain is: a bf16 dropout
// (+ its mask), a bf16 + f32 -> f32 add, an f32 LayerNorm (autocast runs layer_norm in f32) and, in front of every consumer GEMM,
// a cast of the f32 result back to bf16: 4-5 launches forward, 6-7 backward, ~25 bytes per element each way; 33 % of the kernel time
// of the retriever-only step (profiles/r05cfg2_step_by_stream.txt).  Here, rounding where that chain rounds:
//   forward   d = bf16(a keep / (1 - p));  s = f32(d) + res;  y32 = (s - mean) rstd w + b  (f32);  y16 = bf16(y32)
//             a [R, D] bf16 (the dense output), res [R, D] f32 (the previous LayerNorm's f32 output), w / b f32 or bf16;
//             writes y32 (the module's output, and the next residual), y16 (what the consumers' casts would produce),
//             mean / rstd [R], and the keep mask as BITS [R][D / 8] (bit e of byte c = element 8 c + e survives)
//   backward  dy = g32 + f32(g16);  ds = LayerNorm backward of dy (f32);  d_res = ds;  d_a = bf16(bf16(ds) keep / (1 - p))
//             s is recomputed from a, res and the bits (nothing [R, D]-sized is saved beyond the inputs)
// The keep mask is this library's generator (lora_common.hpp: one two-multiply hash per chunk of 8 elements, four words chained by
// xorshift32, 16-bit fields against round(p 65536); oracle/lora_mask.py::keep_mask_v2 restates it) - torch's philox stream
// differs between devices and kernels, there is no reference stream to match.  One wave per row, the row stays in registers.
// Algorithmic bytes per element: forward 2 + 4 read, 4 + 2 + 1/8 written; backward 4 + 2 + 2 + 4 + 1/8 read, 4 + 2 written.
#include "lora_common.hpp"

namespace dalm {
namespace {

using lora::DropArgs;
using lora::DropKey;

__device__ __forceinline__ void dec8(const uint4& v, float (&x)[8]) {
  const unsigned int q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { x[2 * i] = __uint_as_float(q[i] << 16); x[2 * i + 1] = __uint_as_float(q[i] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 enc8(const float (&o)[8]) {
  return make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
}
__device__ __forceinline__ float rb(float x) { return bf16_to_f32(f32_to_bf16(x)); }
__device__ __forceinline__ void ld8f(const float* p, float (&x)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}
__device__ __forceinline__ void st8f(float* p, const float (&x)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(x[4], x[5], x[6], x[7]);
}
// weight / bias vector element chunk, f32 or bf16 storage
template <bool WBF16>
__device__ __forceinline__ void ldw(const void* w, int d, float (&x)[8]) {
  if constexpr (WBF16) dec8(*reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(w) + d), x);
  else ld8f(static_cast<const float*>(w) + d, x);
}

// keep bits of the chunk of 8 elements with flat chunk index c (mask v2 of lora2.hip): bit e = element 8 c + e survives
__device__ __forceinline__ unsigned int keep_bits8(unsigned int c, const DropKey& k, unsigned int thr) {
  unsigned int x = c ^ k.a;
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x += k.b; x *= 0x846ca68bU; x ^= x >> 16;
  unsigned int bits = 0u;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; }
    bits |= ((x & 0xffffu) >= thr ? 1u : 0u) << (2 * q);
    bits |= ((x >> 16) >= thr ? 1u : 0u) << (2 * q + 1);
  }
  return bits;
}

struct AddNormArgs {
  const unsigned short* a;      // [R, D] bf16
  const float* res;             // [R, D] f32
  const void *w, *b;            // [D]
  float* y32;
  unsigned short* y16;
  unsigned char* bits;          // [R][D / 8] or NULL (no dropout)
  float *mean, *rstd;
  const float* g32;             // backward: gradient of y32 (may be NULL)
  const unsigned short* g16;    // backward: gradient of y16 (may be NULL)
  float* d_res;
  unsigned short* d_a;
  int R, D;
  float eps, keep_scale;        // 1 / (1 - p)
  DropArgs drop;                // forward only (thr16 == 0: no dropout)
};

// s = f32(bf16(a keep / (1 - p))) + res for one chunk
__device__ __forceinline__ void sum_chunk(const uint4& ra, const float (&r)[8], unsigned int bits, float ks, bool drop, float (&s)[8]) {
  float av[8];
  dec8(ra, av);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float d = drop ? (((bits >> e) & 1u) ? rb(av[e] * ks) : 0.f) : av[e];
    s[e] = d + r[e];
  }
}

template <int NCH, bool WBF16>
__global__ __launch_bounds__(256) void bert_add_norm_fwd_kernel(const AddNormArgs p) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.R) return;
  const int64_t base = static_cast<int64_t>(row) * p.D;
  const bool drop = p.drop.thr16 != 0u;
  DropKey key = {0u, 1u};
  if (drop) key = lora::drop_key(p.drop);
  float s[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * 8;
    if (d >= p.D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) s[c][e] = 0.f;
      continue;
    }
    const uint4 ra = *reinterpret_cast<const uint4*>(p.a + base + d);
    float r[8];
    ld8f(p.res + base + d, r);
    unsigned int bits = 0xffu;
    if (drop) {
      bits = keep_bits8(static_cast<unsigned int>((base + d) >> 3), key, p.drop.thr16);
      p.bits[(base + d) >> 3] = static_cast<unsigned char>(bits);
    }
    sum_chunk(ra, r, bits, p.keep_scale, drop, s[c]);
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += s[c][e];
  }
  const float mean = wave_sum(sum) / static_cast<float>(p.D);
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * 8;
    if (d >= p.D) continue;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf(s[c][e] - mean, s[c][e] - mean, ss);
  }
  const float rstd = rsqrtf(wave_sum(ss) / static_cast<float>(p.D) + p.eps);
  if (lane == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * 8;
    if (d >= p.D) continue;
    float wv[8], bv[8], o[8];
    ldw<WBF16>(p.w, d, wv);
    ldw<WBF16>(p.b, d, bv);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaf((s[c][e] - mean) * rstd, wv[e], bv[e]);
    st8f(p.y32 + base + d, o);
    *reinterpret_cast<uint4*>(p.y16 + base + d) = enc8(o);
  }
}

template <int NCH, bool WBF16>
__global__ __launch_bounds__(256) void bert_add_norm_bwd_kernel(const AddNormArgs p) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.R) return;
  const int64_t base = static_cast<int64_t>(row) * p.D;
  const bool drop = p.bits != nullptr;
  const float mean = p.mean[row], rstd = p.rstd[row];
  float xh[NCH][8], g[NCH][8];
  unsigned int kb[NCH];
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * 8;
    kb[c] = 0xffu;
    if (d >= p.D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { xh[c][e] = 0.f; g[c][e] = 0.f; }
      continue;
    }
    const uint4 ra = *reinterpret_cast<const uint4*>(p.a + base + d);
    float r[8], s[8], wv[8], dy[8];
    ld8f(p.res + base + d, r);
    if (drop) kb[c] = p.bits[(base + d) >> 3];
    sum_chunk(ra, r, kb[c], p.keep_scale, drop, s);
    ldw<WBF16>(p.w, d, wv);
    if (p.g32) ld8f(p.g32 + base + d, dy);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) dy[e] = 0.f;
    }
    if (p.g16) {
      float h[8];
      dec8(*reinterpret_cast<const uint4*>(p.g16 + base + d), h);
#pragma unroll
      for (int e = 0; e < 8; ++e) dy[e] += h[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xh[c][e] = (s[e] - mean) * rstd;
      g[c][e] = dy[e] * wv[e];
      sg += g[c][e];
      sgx = fmaf(g[c][e], xh[c][e], sgx);
    }
  }
  sg = wave_sum(sg) / static_cast<float>(p.D);
  sgx = wave_sum(sgx) / static_cast<float>(p.D);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int d = (c * 64 + lane) * 8;
    if (d >= p.D) continue;
    float ds[8], da[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ds[e] = rstd * (g[c][e] - sg - xh[c][e] * sgx);
      const float h = rb(ds[e]);                                  // the gradient of the bf16 operand of the add, cast to bf16
      da[e] = drop ? (((kb[c] >> e) & 1u) ? h * p.keep_scale : 0.f) : h;
    }
    st8f(p.d_res + base + d, ds);
    *reinterpret_cast<uint4*>(p.d_a + base + d) = enc8(da);
  }
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace
}  // namespace dalm

using namespace dalm;

#define DALM_BERT_DISPATCH(KERNEL)                                                                          \
  do {                                                                                                      \
    const dim3 grid(static_cast<unsigned>((R + 3) / 4));                                                    \
    hipStream_t s = as_stream(stream);                                                                      \
    if (D <= 512) {                                                                                         \
      if (w_bf16) hipLaunchKernelGGL((KERNEL<1, true>), grid, dim3(256), 0, s, p);                          \
      else hipLaunchKernelGGL((KERNEL<1, false>), grid, dim3(256), 0, s, p);                                \
    } else if (D <= 1024) {                                                                                 \
      if (w_bf16) hipLaunchKernelGGL((KERNEL<2, true>), grid, dim3(256), 0, s, p);                          \
      else hipLaunchKernelGGL((KERNEL<2, false>), grid, dim3(256), 0, s, p);                                \
    } else {                                                                                                \
      if (w_bf16) hipLaunchKernelGGL((KERNEL<4, true>), grid, dim3(256), 0, s, p);                          \
      else hipLaunchKernelGGL((KERNEL<4, false>), grid, dim3(256), 0, s, p);                                \
    }                                                                                                       \
  } while (0)

extern "C" int dalm_bert_add_norm_fwd(const void* a, const float* res, const void* w, const void* b, int w_bf16, int64_t R, int64_t D,
                                      float eps, float dropout_p, const void* seed, uint32_t salt, float* y32, void* y16,
                                      uint8_t* keep_bits, float* mean, float* rstd, dalm_stream_t stream) {
  DALM_REQUIRE(R >= 0, DALM_E_SHAPE, "R must be >= 0");
  if (R == 0) return 0;
  DALM_REQUIRE(a && res && w && b && y32 && y16 && mean && rstd, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(D > 0 && D % 8 == 0 && D <= 2048 && R <= (1ll << 30), DALM_E_SHAPE, "D must be a multiple of 8, at most 2048");
  DALM_REQUIRE(al16(a) && al16(res) && al16(w) && al16(b) && al16(y32) && al16(y16), DALM_E_ALIGN, "tensors must be 16-byte aligned");
  DALM_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || (seed && keep_bits)), DALM_E_SHAPE,
               "dropout needs 0 <= p < 1, a device seed word and the keep-bit buffer");
  DALM_REQUIRE(R * D < (1ll << 35), DALM_E_SHAPE, "tensor too large (the mask's chunk index is 32 bits)");
  AddNormArgs p = {};
  p.a = static_cast<const unsigned short*>(a); p.res = res; p.w = w; p.b = b;
  p.y32 = y32; p.y16 = static_cast<unsigned short*>(y16); p.bits = dropout_p > 0.f ? keep_bits : nullptr;
  p.mean = mean; p.rstd = rstd;
  p.R = static_cast<int>(R); p.D = static_cast<int>(D); p.eps = eps;
  p.keep_scale = dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f;
  p.drop = lora::drop_args(dropout_p, seed, salt);
  if (dropout_p == 0.f) p.drop.thr16 = 0u;
  DALM_BERT_DISPATCH(bert_add_norm_fwd_kernel);
  return check_launch(__func__);
}

extern "C" int dalm_bert_add_norm_bwd(const float* g32, const void* g16, const void* a, const float* res, const void* w, int w_bf16,
                                      const uint8_t* keep_bits, const float* mean, const float* rstd, int64_t R, int64_t D,
                                      float dropout_p, float* d_res, void* d_a, dalm_stream_t stream) {
  DALM_REQUIRE(R >= 0, DALM_E_SHAPE, "R must be >= 0");
  if (R == 0) return 0;
  DALM_REQUIRE((g32 || g16) && a && res && w && mean && rstd && d_res && d_a, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(D > 0 && D % 8 == 0 && D <= 2048 && R <= (1ll << 30), DALM_E_SHAPE, "D must be a multiple of 8, at most 2048");
  DALM_REQUIRE(al16(g32) && al16(g16) && al16(a) && al16(res) && al16(w) && al16(d_res) && al16(d_a), DALM_E_ALIGN,
               "tensors must be 16-byte aligned");
  DALM_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f && (dropout_p == 0.f || keep_bits), DALM_E_SHAPE,
               "dropout needs 0 <= p < 1 and the forward's keep bits");
  AddNormArgs p = {};
  p.a = static_cast<const unsigned short*>(a); p.res = res; p.w = w;
  p.bits = dropout_p > 0.f ? const_cast<unsigned char*>(keep_bits) : nullptr;
  p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd);
  p.g32 = g32; p.g16 = static_cast<const unsigned short*>(g16);
  p.d_res = d_res; p.d_a = static_cast<unsigned short*>(d_a);
  p.R = static_cast<int>(R); p.D = static_cast<int>(D);
  p.keep_scale = dropout_p > 0.f ? 1.0f / (1.0f - dropout_p) : 1.0f;
  DALM_BERT_DISPATCH(bert_add_norm_bwd_kernel);
  return check_launch(__func__);
}
#undef DALM_BERT_DISPATCH
