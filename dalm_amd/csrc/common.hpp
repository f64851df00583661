// Shared device/host helpers for libdalm_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/dalm_hip.h"

// internal cross-file entry points (csrc/lmhead.hip -> csrc/sim.hip): bf16x3 group maxima for the exact top-k's first pass
extern "C" size_t dalm_x3_group_max_workspace_bytes(int64_t m, int64_t n, int64_t D);
extern "C" int dalm_x3_group_max(const float* A, const float* Bm, int64_t m, int64_t n, int64_t D, float scale, float* gmax,
                                 int64_t ng, void* ws, size_t ws_bytes, dalm_stream_t stream);

namespace dalm {

// ---- error plumbing (host) -------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const char* fn, const char* what);
int check_launch(const char* fn);

#define DALM_REQUIRE(cond, code, what) \
  do { if (!(cond)) return ::dalm::fail((code), __func__, (what)); } while (0)

static inline hipStream_t as_stream(dalm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- device helpers ----------------------------------------------------------
constexpr int kWave = 64;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Block-wide reductions through a small LDS scratch (one float per wave).
// Every thread gets the result.  `red` must hold >= BS/64 floats; two barriers
// so that the scratch may be reused right after return.
template <int BS>
__device__ __forceinline__ float block_sum(float v, float* red) {
  constexpr int NW = BS / kWave;
  v = wave_sum(v);
  if constexpr (NW == 1) return v;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) r += red[i];  // fixed order: deterministic
  __syncthreads();
  return r;
}
template <int BS>
__device__ __forceinline__ float block_max(float v, float* red) {
  constexpr int NW = BS / kWave;
  v = wave_max(v);
  if constexpr (NW == 1) return v;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) r = fmaxf(r, red[i]);
  __syncthreads();
  return r;
}

// exp(x) for x <= 0 via v_exp_f32 (2^x); results below the normal range flush to 0.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }

// bf16 <-> f32 (round-to-nearest-even on the way down, as torch does)
__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
  return __uint_as_float(static_cast<unsigned int>(h) << 16);
}
// f32 -> bf16 through the gfx950 hardware converter (v_cvt_pk_bf16_f32: RNE, NaN-preserving)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  f32x2_t v;
  v.x = lo; v.y = hi;
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  return __builtin_bit_cast(unsigned short, static_cast<__bf16>(f));
}

// ---- in-launch hand-off between workgroups (guide section 6, Guideline 16; MI355X_MICROARCH "inter-workgroup visibility"):
// payloads leave through WRITE-THROUGH agent-scope stores (global_store ... sc1) and are re-read with agent-scope,
// L1-bypassing loads (global_load ... sc1); every storing wave drains (s_waitcnt vmcnt(0)), __syncthreads(), ONE lane draws
// an agent-scope ticket.  No fences, no polling.  Used by the one-launch small forward / backward (sim_small.hip) and the
// streaming row statistics (sim.hip).
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_pair(unsigned long long* p, float mx, float l) {
  const unsigned long long g = (static_cast<unsigned long long>(__float_as_uint(l)) << 32) | __float_as_uint(mx);
  __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ld_pair(const unsigned long long* p, float& mx, float& l) {
  const unsigned long long g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  mx = __uint_as_float(static_cast<unsigned>(g));
  l = __uint_as_float(static_cast<unsigned>(g >> 32));
}
// publish: every wave has drained its stores, then ONE lane draws the ticket; returns it to every thread through `slot`
__device__ __forceinline__ unsigned draw_ticket(unsigned* counter, unsigned* slot) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) *slot = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  return *slot;
}

}  // namespace dalm
