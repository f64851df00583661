// What a streaming kernel shaped like the fused CE (one 64 000-byte row per workgroup, read once, written once in
// place) can reach on this box, without any of the CE arithmetic: the ceiling the CE kernel's roofline fraction is
// judged against.  Stand-alone:  hipcc --offload-arch=gfx950 -O3 tools/hbm_stream_probe.hip -o
// tools/_bin/hbm_stream_probe && tools/_bin/hbm_stream_probe [rows] [row_bytes]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// A: flat grid-stride 16-byte copy (the guide's "float4 copy")
template <bool NT>
__global__ __launch_bounds__(256) void flat_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = static_cast<size_t>(gridDim.x) * 256;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n16; i += stride) st<NT>(dst + i, ld<NT>(src + i));
}
// A4: the same, 4 loads in flight per thread
template <bool NT>
__global__ __launch_bounds__(256) void flat_copy4(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = static_cast<size_t>(gridDim.x) * 256;
  size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    u32x4 v0 = ld<NT>(src + i), v1 = ld<NT>(src + i + stride), v2 = ld<NT>(src + i + 2 * stride), v3 = ld<NT>(src + i + 3 * stride);
    st<NT>(dst + i, v0); st<NT>(dst + i + stride, v1); st<NT>(dst + i + 2 * stride, v2); st<NT>(dst + i + 3 * stride, v3);
  }
  for (; i < n16; i += stride) st<NT>(dst + i, ld<NT>(src + i));
}

// B: one row per workgroup, row held in registers between the read and the write; BARRIERS block-wide
// synchronisations in between (the CE has two reductions there); dst may alias src (in place)
template <int BS, int SLOTS, bool NT, int BARRIERS>
__global__ __launch_bounds__(BS) void row_copy(const char* __restrict__ src, char* dst, int row_bytes, int live_mod) {
  __shared__ unsigned red[16];
  const size_t off = static_cast<size_t>(blockIdx.x) * row_bytes;
  const int nslots = row_bytes / 16;
  if (live_mod && (blockIdx.x % live_mod) == live_mod - 1) {          // a padded row: zeros, nothing read
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int s = k * BS + threadIdx.x;
      if (s < nslots) st<NT>(reinterpret_cast<u32x4*>(dst + off) + s, u32x4{0, 0, 0, 0});
    }
    return;
  }
  u32x4 raw[SLOTS];
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const int s = k * BS + threadIdx.x;
    raw[k] = s < nslots ? ld<NT>(reinterpret_cast<const u32x4*>(src + off) + s) : u32x4{0, 0, 0, 0};
  }
  unsigned acc = 0;
#pragma unroll
  for (int b = 0; b < BARRIERS; ++b) {
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) acc ^= raw[k].x + b;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    acc += red[(threadIdx.x >> 6) ^ 1];
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const int s = k * BS + threadIdx.x;
    if (BARRIERS) raw[k].y ^= (acc & 1u);      // keep the dependency on the "reduction"
    if (s < nslots) st<NT>(reinterpret_cast<u32x4*>(dst + off) + s, raw[k]);
  }
}

// C: persistent workgroups: row r+G's loads are issued before row r's stores (software pipeline across rows)
template <int BS, int SLOTS, bool NT>
__global__ __launch_bounds__(BS) void row_copy_pipelined(const char* __restrict__ src, char* dst, int row_bytes, int rows) {
  const int nslots = row_bytes / 16;
  int r = blockIdx.x;
  if (r >= rows) return;
  u32x4 cur[SLOTS], nxt[SLOTS];
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const int s = k * BS + threadIdx.x;
    cur[k] = s < nslots ? ld<NT>(reinterpret_cast<const u32x4*>(src + static_cast<size_t>(r) * row_bytes) + s) : u32x4{0, 0, 0, 0};
  }
  for (; r < rows; r += gridDim.x) {
    const int rn = r + gridDim.x;
    if (rn < rows) {
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        const int s = k * BS + threadIdx.x;
        nxt[k] = s < nslots ? ld<NT>(reinterpret_cast<const u32x4*>(src + static_cast<size_t>(rn) * row_bytes) + s) : u32x4{0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int s = k * BS + threadIdx.x;
      if (s < nslots) st<NT>(reinterpret_cast<u32x4*>(dst + static_cast<size_t>(r) * row_bytes) + s, cur[k]);
    }
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) cur[k] = nxt[k];
  }
}

// D / E: read-only and write-only
template <bool NT>
__global__ __launch_bounds__(256) void read_only(const u32x4* __restrict__ src, size_t n16, unsigned* sink) {
  const size_t stride = static_cast<size_t>(gridDim.x) * 256;
  unsigned a = 0;
  size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    u32x4 v0 = ld<NT>(src + i), v1 = ld<NT>(src + i + stride), v2 = ld<NT>(src + i + 2 * stride), v3 = ld<NT>(src + i + 3 * stride);
    a ^= v0.x ^ v1.y ^ v2.z ^ v3.w;
  }
  for (; i < n16; i += stride) a ^= ld<NT>(src + i).x;
  if (a == 0x12345678u) *sink = a;
}
template <bool NT>
__global__ __launch_bounds__(256) void write_only(u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = static_cast<size_t>(gridDim.x) * 256;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n16; i += stride) st<NT>(dst + i, u32x4{1, 2, 3, 4});
}

template <typename F>
static double time_us(F&& launch, int iters = 30) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int i = 0; i < iters; ++i) {
    CK(hipEventRecord(a, 0));
    launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

static void report(const char* name, double us, double bytes) {
  printf("%-58s %9.2f us  %7.0f GB/s  %.3f of 8 TB/s\n", name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 4608;
  const int row_bytes = argc > 2 ? atoi(argv[2]) : 64000;
  const size_t bytes = static_cast<size_t>(rows) * row_bytes, n16 = bytes / 16;
  char *a, *b; unsigned* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  printf("# %d rows x %d bytes = %.1f MB per buffer; median of 30 launches, HIP events on the null stream\n", rows, row_bytes, bytes / 1e6);
  const double rw = 2.0 * bytes;
  auto A = reinterpret_cast<const u32x4*>(a); auto B = reinterpret_cast<u32x4*>(b);
  report("hipMemcpyDtoD", time_us([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }), rw);
  for (int g : {1024, 2048, 4096, 8192}) {
    char nm[96];
    snprintf(nm, sizeof nm, "flat copy, plain, grid %d", g);
    report(nm, time_us([&] { hipLaunchKernelGGL(flat_copy<false>, dim3(g), dim3(256), 0, 0, A, B, n16); }), rw);
    snprintf(nm, sizeof nm, "flat copy, nt, grid %d", g);
    report(nm, time_us([&] { hipLaunchKernelGGL(flat_copy<true>, dim3(g), dim3(256), 0, 0, A, B, n16); }), rw);
    snprintf(nm, sizeof nm, "flat copy x4 in flight, nt, grid %d", g);
    report(nm, time_us([&] { hipLaunchKernelGGL(flat_copy4<true>, dim3(g), dim3(256), 0, 0, A, B, n16); }), rw);
  }
  report("read only, nt, grid 2048", time_us([&] { hipLaunchKernelGGL(read_only<true>, dim3(2048), dim3(256), 0, 0, A, n16, sink); }), bytes);
  report("read only, plain, grid 2048", time_us([&] { hipLaunchKernelGGL(read_only<false>, dim3(2048), dim3(256), 0, 0, A, n16, sink); }), bytes);
  report("write only, nt, grid 2048", time_us([&] { hipLaunchKernelGGL(write_only<true>, dim3(2048), dim3(256), 0, 0, B, n16); }), bytes);
  report("write only, plain, grid 2048", time_us([&] { hipLaunchKernelGGL(write_only<false>, dim3(2048), dim3(256), 0, 0, B, n16); }), bytes);
  if (row_bytes <= 512 * 8 * 16 && row_bytes % 16 == 0) {
#define ROW(BS, SL, NT, BAR, dstp, lm, label, nbytes) \
    report(label, time_us([&] { hipLaunchKernelGGL((row_copy<BS, SL, NT, BAR>), dim3(rows), dim3(BS), 0, 0, a, dstp, row_bytes, lm); }), nbytes)
    ROW(512, 8, true, 0, b, 0, "row/WG 512thr, nt, no barrier, out of place", rw);
    ROW(512, 8, true, 0, a, 0, "row/WG 512thr, nt, no barrier, IN PLACE", rw);
    ROW(512, 8, true, 2, a, 0, "row/WG 512thr, nt, 2 reductions, IN PLACE  (CE shape)", rw);
    ROW(512, 8, false, 2, a, 0, "row/WG 512thr, plain, 2 reductions, IN PLACE", rw);
    ROW(1024, 4, true, 2, a, 0, "row/WG 1024thr, nt, 2 reductions, IN PLACE", rw);
    ROW(256, 16, true, 2, a, 0, "row/WG 256thr x16 slots, nt, 2 reductions, IN PLACE", rw);
    // the bench's mask: 2873 of 4608 rows live (read + write), the rest written as zeros: every 8th..  approx 3 of 8 dead
    const double live_frac = 2.0 / 3.0;   // live_mod = 3 -> 1 of 3 rows is padding
    ROW(512, 8, true, 2, a, 3, "row/WG 512thr, nt, 2 reductions, IN PLACE, 1/3 rows zero-filled", bytes * (live_frac * 2 + (1 - live_frac)));
    for (int g : {512, 768, 1024, 2048}) {
      char nm[96];
      snprintf(nm, sizeof nm, "persistent row pipeline 512thr nt, grid %d, IN PLACE", g);
      report(nm, time_us([&] { hipLaunchKernelGGL((row_copy_pipelined<512, 8, true>), dim3(g), dim3(512), 0, 0, a, a, row_bytes, rows); }), rw);
    }
    report("persistent row pipeline 256thr x16, nt, grid 1024, IN PLACE",
           time_us([&] { hipLaunchKernelGGL((row_copy_pipelined<256, 16, true>), dim3(1024), dim3(256), 0, 0, a, a, row_bytes, rows); }), rw);
  }
  return 0;
}
