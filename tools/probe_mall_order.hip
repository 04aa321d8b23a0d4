// Does the direction a streaming pass sweeps its array in matter when the PREVIOUS launch just wrote (or read) the same array in
// ascending order?  (256 MiB memory-side cache: the tail of the previous sweep may still be resident.)  Per size: a write sweep followed
// by a read sweep ascending / descending, a read sweep followed by a read sweep ascending / descending; HIP events around the SECOND
// launch only, median of 15.   hipcc -O3 --offload-arch=gfx950 tools/probe_mall_order.hip -o scratch/probe_mall_order
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
constexpr int U = 4;

// workgroup b takes the contiguous pieces b, b + grid, ... of 256 * U 16-byte units (ascending) or the mirrored ones (descending)
__global__ __launch_bounds__(256) void read_kernel(const u4* __restrict__ src, u4* __restrict__ sink, size_t n, int descending) {
  u4 acc = {0, 0, 0, 0};
  const size_t piece = 256 * U, pieces = n / piece;
  for (size_t p = blockIdx.x; p < pieces; p += gridDim.x) {
    const size_t q = descending ? pieces - 1 - p : p;
    u4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = src[q * piece + k * 256 + threadIdx.x];
#pragma unroll
    for (int k = 0; k < U; ++k) acc ^= v[k];
  }
  if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void write_kernel(u4* __restrict__ dst, size_t n) {
  const size_t piece = 256 * U, pieces = n / piece;
  const u4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
  for (size_t p = blockIdx.x; p < pieces; p += gridDim.x)
#pragma unroll
    for (int k = 0; k < U; ++k) dst[p * piece + k * 256 + threadIdx.x] = v;
}

int main() {
  const size_t sizes_mb[] = {128, 256, 458, 916, 1832};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  u4 *buf, *sink;
  hipMalloc(&buf, (size_t)2048 << 20);
  hipMalloc(&sink, 1 << 24);
  const int grid = 256 * 8;
  for (size_t mb : sizes_mb) {
    const size_t n = (mb << 20) / 16;
    for (int first = 0; first < 2; ++first)          // 0: a write sweep first, 1: a read sweep first
      for (int desc = 0; desc < 2; ++desc) {
        std::vector<float> t;
        for (int rep = 0; rep < 15; ++rep) {
          if (first == 0) hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, buf, n);
          else hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, buf, sink, n, 0);
          hipEventRecord(e0, 0);
          hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, buf, sink, n, desc);
          hipEventRecord(e1, 0);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        const float ms = t[t.size() / 2];
        printf("%5zu MB  after a %s sweep, read %s: %.4f ms = %.2f TB/s\n", mb, first ? "read " : "write", desc ? "descending" : "ascending ", ms,
               (double)(mb << 20) / ms / 1e9);
      }
  }
  return 0;
}
