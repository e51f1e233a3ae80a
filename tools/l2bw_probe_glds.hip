// Micro-probe 2: L2 -> LDS bandwidth per CU with global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void probe_glds(const u32x4* __restrict__ buf, size_t region_vec, int iters, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) u32x4 lds[UNROLL * 256];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  size_t base = ((size_t)blockIdx.x * 977) % region_vec;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t idx = (base + (size_t)u * 256 + tid) % region_vec;
      // LDS destination: wave-uniform base + lane*16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(buf + idx),
                                       (__attribute__((address_space(3))) void*)(&lds[u * 256 + wave * 64]), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    base = (base + 4099 * 16) % region_vec;
  }
  __syncthreads();
  if (lds[tid][0] == 0x12345678u) sink[0] = 1;
}

int main() {
  const size_t region_bytes = 2u << 20;
  u32x4* buf; unsigned* sink;
  hipMalloc(&buf, region_bytes + (1 << 20)); hipMalloc(&sink, 64);
  hipMemset(buf, 1, region_bytes + (1 << 20));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int bpc = 1; bpc <= 4; bpc *= 2) {
    const int grid = 256 * bpc;
    auto run = [&](auto kern, int unroll) {
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf, region_bytes / 16, 10, sink);
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf, region_bytes / 16, iters, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)grid * 256 * 16 * unroll * iters;
      printf("glds  blocks/CU %d  in flight/thread %2d : %7.2f TB/s = %5.1f B/clk/CU @2.1GHz\n", bpc, unroll, bytes / ms / 1e9,
             bytes / (ms * 1e-3) / 256 / 2.1e9);
    };
    run(probe_glds<4>, 4);
    run(probe_glds<8>, 8);
  }
  return 0;
}
