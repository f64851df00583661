// Shared pieces of the LoRA kernels (lora.hip: f32 / odd shapes, the mask recomputed; lora2.hip: bf16, stacked projections,
// the mask stored as bits): 8-element row chunks, the dropout key.
#pragma once
#include "common.hpp"

namespace dalm {
namespace lora {

struct bf16_t { unsigned short v; };

// ---- 8 consecutive elements of a row as f32 ----
template <typename T> struct Chunk8;
template <> struct Chunk8<float> {
  struct Raw { float4 a, b; };
  __device__ static __forceinline__ Raw load_raw(const float* p) {
    Raw r; r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 4); return r;
  }
  __device__ static __forceinline__ Raw zero_raw() { Raw r; r.a = make_float4(0.f, 0.f, 0.f, 0.f); r.b = r.a; return r; }
  __device__ static __forceinline__ void decode(const Raw& r, float (&x)[8]) {
    x[0] = r.a.x; x[1] = r.a.y; x[2] = r.a.z; x[3] = r.a.w; x[4] = r.b.x; x[5] = r.b.y; x[6] = r.b.z; x[7] = r.b.w;
  }
  __device__ static __forceinline__ void load(const float* p, float (&x)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
  }
  __device__ static __forceinline__ void store(float* p, const float (&x)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(x[4], x[5], x[6], x[7]);
  }
};
template <> struct Chunk8<bf16_t> {
  typedef uint4 Raw;
  __device__ static __forceinline__ Raw load_raw(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  __device__ static __forceinline__ Raw zero_raw() { return make_uint4(0u, 0u, 0u, 0u); }
  __device__ static __forceinline__ void decode(const Raw& v, float (&x)[8]) {
    const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[2 * i] = __uint_as_float(w[i] << 16);
      x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
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
    uint4 o;
    o.x = pack_bf16x2(x[0], x[1]); o.y = pack_bf16x2(x[2], x[3]);
    o.z = pack_bf16x2(x[4], x[5]); o.w = pack_bf16x2(x[6], x[7]);
    *reinterpret_cast<uint4*>(p) = o;
  }
};

// ---- the dropout mask ----
struct DropArgs { const unsigned long long* seed; unsigned int salt; unsigned int thr16; };   // keep iff 16-bit field >= thr16

__device__ __forceinline__ unsigned int lowbias32(unsigned int x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
struct DropKey { unsigned int a, b; };
__device__ __forceinline__ DropKey drop_key(const DropArgs& d) {
  const unsigned long long s = d.seed ? *d.seed : 0ull;
  DropKey k;
  k.a = lowbias32(static_cast<unsigned int>(s) ^ (d.salt * 0x9E3779B9u));
  k.b = lowbias32(static_cast<unsigned int>(s >> 32) + d.salt + 0x85ebca6bu) | 1u;
  return k;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline DropArgs drop_args(float p, const void* seed, unsigned int salt) {
  DropArgs d;
  d.seed = static_cast<const unsigned long long*>(seed);
  d.salt = salt;
  d.thr16 = static_cast<unsigned int>(p * 65536.0f + 0.5f);
  return d;
}

}  // namespace lora
}  // namespace dalm
