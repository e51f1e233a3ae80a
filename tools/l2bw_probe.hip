// Micro-probe: how many bytes per clock can one CU pull from L2 with global_load_dwordx4?
// (sizing input for the trunk kernel: its 128x128 hl16 tile needs 64 KB per 1536 MFMA cycles)
//   hipcc --offload-arch=gfx950 -O3 tools/l2bw_probe.hip -o /tmp/l2bw && /tmp/l2bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// pattern 0: fully contiguous 1 KB per wave instruction; pattern 1: 8 lanes x 16 B = 128 B per row,
// rows `row_stride` bytes apart (the trunk kernel's activation pattern); UNROLL independent loads in flight.
template <int UNROLL>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ buf, size_t region_vec, int iters,
                                             int pattern, int row_stride_vec, unsigned* sink) {
  const int tid = threadIdx.x;
  const int lane_row = tid >> 3, piece = tid & 7;
  size_t base = ((size_t)blockIdx.x * 977) % region_vec;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      size_t idx;
      if (pattern == 0) idx = base + (size_t)u * 256 + tid;
      else if (pattern == 1) idx = base + ((size_t)(u * 32 + lane_row)) * row_stride_vec + piece;
      // pattern 2: like 1 but every access of every wave falls into the first 128 bytes of a 1-KiB-aligned
      // row (the trunk kernel at Cin = 256: all workgroups read the same 32-channel slab offset at once)
      else idx = (base / row_stride_vec) * row_stride_vec + ((size_t)(u * 32 + lane_row)) * row_stride_vec + piece;
      v[u] = buf[idx % region_vec];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    base = (base + 4099 * 16) % region_vec;
  }
  if (acc[0] == 0x12345678u) sink[0] = acc[1] + acc[2] + acc[3];
}

int main() {
  const size_t region_bytes = 3u << 20;  // 3 MiB: L2 resident in every XCD
  u32x4* buf;
  unsigned* sink;
  hipMalloc(&buf, region_bytes + (1 << 20));
  hipMalloc(&sink, 64);
  hipMemset(buf, 1, region_bytes + (1 << 20));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2000;
  for (int pattern = 0; pattern < 3; ++pattern)
    for (int blocks_per_cu = 1; blocks_per_cu <= 4; blocks_per_cu *= 2) {
      const int grid = 256 * blocks_per_cu;
      auto run = [&](auto kern, int unroll) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf, region_bytes / 16, 10, pattern, 64, sink);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf, region_bytes / 16, iters, pattern, 64, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)grid * 256 * 16 * unroll * iters;
        printf("pattern %d  blocks/CU %d  loads in flight/thread %2d : %7.2f TB/s  = %5.1f B/clk/CU @2.1GHz\n", pattern,
               blocks_per_cu, unroll, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.1e9);
      };
      run(probe<4>, 4);
      run(probe<16>, 16);
    }
  return 0;
}
