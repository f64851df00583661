// dalm_nf4_*: blockwise 4-bit NormalFloat storage of the frozen base weights (`use_bnb`).
//
// What the reference asks bitsandbytes for (dalm/models/rag_e2e_base_model.py:137-142,
// retriever_only_base_model.py:26,86): BitsAndBytesConfig(load_in_4bit, bnb_4bit_quant_type="nf4",
// bnb_4bit_compute_dtype=bfloat16) - blocksize 64, f32 absmax per block (no double quantisation), uint8 storage.
// bitsandbytes is not vendored in the reference and absent from this image; the algorithm restated here is the
// published one (QLoRA, Dettmers et al. 2023, appendix E: the 16 NF4 levels; bitsandbytes csrc/kernels.cu
// kQuantizeBlockwise / kDequantizeBlockwise with DATA_TYPE = NF4):
//   per block of 64 consecutive elements of the flattened weight: a = max|w|;  q_i = nearest level to w_i * (1/a)
//   (ties to the lower level - the decision tree compares with `>` against the midpoints);
//   two 4-bit indices per byte, element 2j in the HIGH nibble;  w'_i = level[q_i] * a.
// Both kernels are pure streaming work (HBM-bound): dequantise moves 0.5 B + 1/16 B in and 2 B (bf16) out per weight.
#include "common.hpp"

namespace {
using namespace dalm;

__device__ __constant__ float kNf4Level[16] = {
    -1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
    -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
    0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
    0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};
// decision thresholds between neighbouring levels (the midpoints, as bitsandbytes' dQuantizeNF4 spells them)
__device__ __constant__ float kNf4Mid[15] = {
    -0.8480964004993439f, -0.6106329262256622f, -0.4599952697753906f, -0.33967943489551544f,
    -0.23460740596055984f, -0.13791173323988914f, -0.045525018125772476f, 0.03979014977812767f,
    0.1202552504837513f, 0.2035212516784668f, 0.2920137718319893f, 0.3893125355243683f,
    0.5016634166240692f, 0.6427869200706482f, 0.8614784181118011f};

__device__ __forceinline__ unsigned int nf4_index(float x) {
  // binary search over the 15 thresholds (4 compares), `>` so that a tie goes to the lower level
  unsigned int i = x > kNf4Mid[7] ? 8u : 0u;
  i += x > kNf4Mid[i + 3] ? 4u : 0u;
  i += x > kNf4Mid[i + 1] ? 2u : 0u;
  i += x > kNf4Mid[i] ? 1u : 0u;
  return i;
}

template <typename T> __device__ __forceinline__ float load_as_f32(const T* p, int64_t i);
template <> __device__ __forceinline__ float load_as_f32<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float load_as_f32<unsigned short>(const unsigned short* p, int64_t i) {
  return bf16_to_f32(p[i]);
}

// One thread per 8 consecutive weights (4 packed bytes out), 8 neighbouring lanes per 64-weight block.
template <typename T>
__global__ __launch_bounds__(256) void nf4_quantize_kernel(const T* __restrict__ w, int64_t n,
                                                           unsigned char* __restrict__ packed,
                                                           float* __restrict__ absmax) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int64_t e0 = t * 8;
  float v[8];
  if (e0 + 8 <= n) {
    if constexpr (sizeof(T) == 4) {
      const float4 a = *reinterpret_cast<const float4*>(w + e0), b = *reinterpret_cast<const float4*>(w + e0 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
      const uint4 a = *reinterpret_cast<const uint4*>(w + e0);
      const unsigned int u[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[2 * j] = __uint_as_float(u[j] << 16);
        v[2 * j + 1] = __uint_as_float(u[j] & 0xffff0000u);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = e0 + j < n ? load_as_f32<T>(w, e0 + j) : 0.f;
  }
  float a = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) a = fmaxf(a, fabsf(v[j]));
  a = fmaxf(a, __shfl_xor(a, 1, 64));
  a = fmaxf(a, __shfl_xor(a, 2, 64));
  a = fmaxf(a, __shfl_xor(a, 4, 64));
  if ((threadIdx.x & 7) == 0 && e0 < n) absmax[e0 >> 6] = a;
  const float inv = a > 0.f ? 1.0f / a : 0.f;   // an all-zero block stores level 7 (0.0) everywhere
  unsigned int word = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned int byte = (nf4_index(v[2 * j] * inv) << 4) | nf4_index(v[2 * j + 1] * inv);
    word |= byte << (8 * j);
  }
  const int64_t b0 = t * 4, nbytes = (n + 1) >> 1;
  if (b0 + 4 <= nbytes) {
    *reinterpret_cast<unsigned int*>(packed + b0) = word;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (b0 + j < nbytes) packed[b0 + j] = static_cast<unsigned char>(word >> (8 * j));
  }
}

// One thread per 8 consecutive weights and step; a workgroup walks STEPS consecutive tiles of 2048 weights with all its loads
// issued before the first store (22 000 single-step workgroups of one 4-byte load + one 16-byte store each were bound by
// workgroup dispatch, not by HBM).
template <typename T, int STEPS, bool NTL>
__global__ __launch_bounds__(256) void nf4_dequantize_kernel(const unsigned char* __restrict__ packed,
                                                             const float* __restrict__ absmax, int64_t n,
                                                             T* __restrict__ out) {
  __shared__ float level[16];
  if (threadIdx.x < 16) level[threadIdx.x] = kNf4Level[threadIdx.x];
  __syncthreads();
  const int64_t nbytes = (n + 1) >> 1;
  const int64_t t0 = static_cast<int64_t>(blockIdx.x) * (256 * STEPS) + threadIdx.x;
  unsigned int word[STEPS];
  float a[STEPS];
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    const int64_t t = t0 + k * 256, e0 = t * 8, b0 = t * 4;
    word[k] = 0; a[k] = 0.f;
    if (e0 < n) {
      a[k] = absmax[e0 >> 6];
      if (b0 + 4 <= nbytes) {
        const unsigned int* src = reinterpret_cast<const unsigned int*>(packed + b0);
        word[k] = NTL ? __builtin_nontemporal_load(src) : *src;
      } else {
        for (int j = 0; j < 4; ++j)
          if (b0 + j < nbytes) word[k] |= static_cast<unsigned int>(packed[b0 + j]) << (8 * j);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    const int64_t e0 = (t0 + k * 256) * 8;
    if (e0 >= n) continue;
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned int byte = (word[k] >> (8 * j)) & 0xffu;
      v[2 * j] = level[byte >> 4] * a[k];
      v[2 * j + 1] = level[byte & 15u] * a[k];
    }
    if (e0 + 8 <= n) {
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(out + e0) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(out + e0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4*>(out + e0) = o;
      }
    } else {
      for (int j = 0; j < 8 && e0 + j < n; ++j) {
        if constexpr (sizeof(T) == 4) out[e0 + j] = v[j];
        else out[e0 + j] = f32_to_bf16(v[j]);
      }
    }
  }
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

}  // namespace

extern "C" size_t dalm_nf4_packed_bytes(int64_t n) { return n > 0 ? static_cast<size_t>((n + 1) / 2) : 0; }
extern "C" size_t dalm_nf4_absmax_count(int64_t n) { return n > 0 ? static_cast<size_t>((n + 63) / 64) : 0; }

extern "C" int dalm_nf4_quantize(const void* w, int dtype, int64_t n, uint8_t* packed, float* absmax,
                                 dalm_stream_t stream) {
  DALM_REQUIRE(n >= 0, DALM_E_SHAPE, "n must be >= 0");
  if (n == 0) return 0;
  DALM_REQUIRE(w && packed && absmax, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dtype == DALM_F32 || dtype == DALM_BF16, DALM_E_DTYPE, "dtype must be DALM_F32 or DALM_BF16");
  DALM_REQUIRE(aligned(w, 16) && aligned(packed, 4) && aligned(absmax, 4), DALM_E_ALIGN,
               "w must be 16-byte aligned, packed / absmax 4-byte aligned");
  const int64_t threads = (n + 7) / 8, blocks = (threads + 255) / 256;
  DALM_REQUIRE(blocks <= 0x7fffffffLL, DALM_E_SHAPE, "n too large for one launch");
  if (dtype == DALM_F32)
    hipLaunchKernelGGL(nf4_quantize_kernel<float>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, as_stream(stream),
                       static_cast<const float*>(w), n, packed, absmax);
  else
    hipLaunchKernelGGL(nf4_quantize_kernel<unsigned short>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0,
                       as_stream(stream), static_cast<const unsigned short*>(w), n, packed, absmax);
  return check_launch(__func__);
}

extern "C" int dalm_nf4_dequantize(const uint8_t* packed, const float* absmax, int64_t n, int dtype, void* out,
                                   dalm_stream_t stream) {
  DALM_REQUIRE(n >= 0, DALM_E_SHAPE, "n must be >= 0");
  if (n == 0) return 0;
  DALM_REQUIRE(packed && absmax && out, DALM_E_NULL, "null pointer argument");
  DALM_REQUIRE(dtype == DALM_F32 || dtype == DALM_BF16, DALM_E_DTYPE, "dtype must be DALM_F32 or DALM_BF16");
  DALM_REQUIRE(aligned(out, 16) && aligned(packed, 4) && aligned(absmax, 4), DALM_E_ALIGN,
               "out must be 16-byte aligned, packed / absmax 4-byte aligned");
  const int64_t threads = (n + 7) / 8;
  static const int steps_env = [] { const char* e = getenv("DALM_NF4_STEPS"); return e ? atoi(e) : 0; }();
  // measured (profiles/history/r04_nf4_steps.txt, 8 different weights in turn inside a hipGraph): bf16 out 11008x4096 28.0 / 23.0 /
  // 21.3 / 21.7 us and 4096^2 9.3 / 9.2 / 8.6 / 8.6 us for 1 / 2 / 4 / 8 tiles per workgroup; f32 out is best at 1 (46.7 us
  // vs 48.9 at 4: twice the store bytes per tile already); a non-temporal load of the packed words is no gain (DALM_NF4_NT=1)
  int steps = steps_env > 0 ? steps_env : ((dtype == DALM_BF16 && threads >= 256 * 4 * 2048) ? 4 : 1);
  if (steps != 1 && steps != 2 && steps != 8) steps = 4;
  const int64_t blocks = (threads + 256 * steps - 1) / (256 * steps);
  DALM_REQUIRE(blocks <= 0x7fffffffLL, DALM_E_SHAPE, "n too large for one launch");
  static const bool ntl = [] { const char* e = getenv("DALM_NF4_NT"); return e && atoi(e) != 0; }();
#define DALM_NF4_DEQ(TT, S) \
  if (ntl) hipLaunchKernelGGL((nf4_dequantize_kernel<TT, S, true>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, \
                              as_stream(stream), packed, absmax, n, static_cast<TT*>(out)); \
  else hipLaunchKernelGGL((nf4_dequantize_kernel<TT, S, false>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, \
                          as_stream(stream), packed, absmax, n, static_cast<TT*>(out))
#define DALM_NF4_DEQ_S(TT) \
  switch (steps) { case 1: DALM_NF4_DEQ(TT, 1); break; case 2: DALM_NF4_DEQ(TT, 2); break; case 8: DALM_NF4_DEQ(TT, 8); break; \
                   default: DALM_NF4_DEQ(TT, 4); break; }
  if (dtype == DALM_F32) { DALM_NF4_DEQ_S(float) } else { DALM_NF4_DEQ_S(unsigned short) }
#undef DALM_NF4_DEQ_S
#undef DALM_NF4_DEQ
  return check_launch(__func__);
}
