// What ds_read_b64_tr_b16 delivers (gfx950): LDS holds element index e at byte 2 e; lane l reads 8 bytes at `addr(l)` and prints the
// four 16-bit values it receives.  Run on the GPU box: hipcc --offload-arch=gfx950 tools/tr_read_probe.hip -o /tmp/trp && /tmp/trp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void probe(int mode, int stride_b, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = static_cast<uint16_t>(i);
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = 8u * l;                                               // lane-linear
  else addr = static_cast<unsigned>((l & 15) / 4 * stride_b + 8 * ((l & 15) % 4) + 32 * (l >> 4));   // 16-lane group: 4 rows x 16 cols, row stride stride_b
  const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds));   // LDS address = low 32 bits of the generic pointer
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + addr) : "memory");
  for (int e = 0; e < 4; ++e) out[4 * l + e] = static_cast<uint16_t>(v >> (16 * e));
}

int main() {
  uint16_t* d;
  hipMalloc(&d, 256 * 2);
  uint16_t h[256];
  const int modes[3][2] = {{0, 0}, {1, 64}, {1, 272}};
  for (auto& m : modes) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, m[0], m[1], d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("mode %d stride %d B (values are element indices = byte offset / 2)\n", m[0], m[1]);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  }
  return 0;
}
