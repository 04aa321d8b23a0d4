// Read-only, write-only and copy streaming rates of the box (HIP events, median of 20), to put the segment kernels' read-dominated
// traffic against the ceiling its direction mix has:  hipcc -O3 --offload-arch=gfx950 tools/probe_read_bw.hip -o /tmp/probe_read_bw
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void read_kernel(const u4* __restrict__ src, u4* __restrict__ sink, size_t n) {
  u4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
    u4 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = i + k * 256 < n ? __builtin_nontemporal_load(src + i + k * 256) : u4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < U; ++k) acc ^= v[k];
  }
  if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void write_kernel(u4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  const u4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
}

__global__ __launch_bounds__(256) void write_nt_kernel(u4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  const u4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) __builtin_nontemporal_store(v, dst + i);
}

// every workgroup writes one contiguous share (a "persistent" output stream) instead of grid-strided 4-KB pieces
__global__ __launch_bounds__(256) void write_chunk_kernel(u4* __restrict__ dst, size_t n) {
  const size_t per = (n + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
  const u4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) dst[i] = v;
}

// grid-strided pieces of `piece` 16-byte units per workgroup (4 KB = 256 units is write_kernel)
__global__ __launch_bounds__(256) void write_piece_kernel(u4* __restrict__ dst, size_t n, int piece) {
  const u4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
  for (size_t base = (size_t)blockIdx.x * piece; base < n; base += (size_t)gridDim.x * piece)
    for (int i = threadIdx.x; i < piece && base + i < n; i += 256) dst[base + i] = v;
}

// the segment kernels' store shape: 16 lanes write one 256-byte row; a wavefront's 4 rows are `gap` rows apart
__global__ __launch_bounds__(256) void write_rows_kernel(u4* __restrict__ dst, size_t rows, int rows_per_pass) {
  const u4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
  const int g = threadIdx.x >> 4, sub = threadIdx.x & 15;         // 16 lane groups per workgroup
  for (size_t base = (size_t)blockIdx.x * rows_per_pass; base < rows; base += (size_t)gridDim.x * rows_per_pass)
    for (int r = g; r < rows_per_pass && base + r < rows; r += 16) dst[(base + r) * 16 + sub] = v;
}

__global__ __launch_bounds__(256) void copy_kernel(const u4* __restrict__ src, u4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// rows of 256 B read through an index list (a random permutation of blocks of `run` consecutive rows), 16 lanes per row
__global__ __launch_bounds__(256) void gather_kernel(const u4* __restrict__ src, const int* __restrict__ idx, u4* __restrict__ sink, size_t rows) {
  u4 acc = {0, 0, 0, 0};
  const int sub = threadIdx.x & 15;
  const size_t stride = (size_t)gridDim.x * 16;
  for (size_t r = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < rows; r += stride) acc ^= src[(size_t)idx[r] * 16 + sub];
  if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename F> static float median_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  std::vector<float> t;
  for (int i = 0; i < 20; ++i) {
    hipEventRecord(e0, 0);
    launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main() {
  const size_t bytes = (size_t)1 << 30;                  // 1 GiB per buffer: far beyond L2 + MALL
  const size_t n = bytes / 16;
  u4 *a, *b, *sink;
  hipMalloc(&a, bytes);
  hipMalloc(&b, bytes);
  hipMalloc(&sink, 4096 * 256 * 16);
  hipMemset(a, 1, bytes);
  hipMemset(b, 2, bytes);
  for (int grid : {1024, 2048, 4096}) {
    float ms = median_ms([&] { hipLaunchKernelGGL(read_kernel<1>, dim3(grid), dim3(256), 0, 0, a, sink, n); });
    printf("{\"kernel\": \"read_u1\", \"grid\": %d, \"ms\": %.4f, \"TBps\": %.3f}\n", grid, ms, bytes / ms / 1e9);
    ms = median_ms([&] { hipLaunchKernelGGL(read_kernel<4>, dim3(grid), dim3(256), 0, 0, a, sink, n); });
    printf("{\"kernel\": \"read_u4\", \"grid\": %d, \"ms\": %.4f, \"TBps\": %.3f}\n", grid, ms, bytes / ms / 1e9);
    ms = median_ms([&] { hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, 0, a, sink, n); });
    printf("{\"kernel\": \"read_u8\", \"grid\": %d, \"ms\": %.4f, \"TBps\": %.3f}\n", grid, ms, bytes / ms / 1e9);
    ms = median_ms([&] { hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, b, n); });
    printf("{\"kernel\": \"write\", \"grid\": %d, \"ms\": %.4f, \"TBps\": %.3f}\n", grid, ms, bytes / ms / 1e9);
    ms = median_ms([&] { hipLaunchKernelGGL(write_nt_kernel, dim3(grid), dim3(256), 0, 0, b, n); });
    printf("{\"kernel\": \"write_nontemporal\", \"grid\": %d, \"ms\": %.4f, \"TBps\": %.3f}\n", grid, ms, bytes / ms / 1e9);
    ms = median_ms([&] { hipLaunchKernelGGL(write_chunk_kernel, dim3(grid), dim3(256), 0, 0, b, n); });
    printf("{\"kernel\": \"write_contiguous_share\", \"grid\": %d, \"ms\": %.4f, \"TBps\": %.3f}\n", grid, ms, bytes / ms / 1e9);
    ms = median_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n); });
    printf("{\"kernel\": \"copy\", \"grid\": %d, \"ms\": %.4f, \"TBps_read_plus_write\": %.3f}\n", grid, ms, 2.0 * bytes / ms / 1e9);
  }
  for (int grid : {1024, 2048})
    for (int piece : {256, 512, 1024, 2048, 4096, 16384}) {
      float ms = median_ms([&] { hipLaunchKernelGGL(write_piece_kernel, dim3(grid), dim3(256), 0, 0, b, n, piece); });
      printf("{\"kernel\": \"write_piece\", \"grid\": %d, \"piece_KB\": %d, \"ms\": %.4f, \"TBps\": %.3f}\n", grid, piece / 64, ms, bytes / ms / 1e9);
    }
  for (int grid : {2048})
    for (int rpp : {16, 32, 64, 128, 512}) {
      float ms = median_ms([&] { hipLaunchKernelGGL(write_rows_kernel, dim3(grid), dim3(256), 0, 0, b, bytes / 256, rpp); });
      printf("{\"kernel\": \"write_rows256\", \"grid\": %d, \"rows_per_pass\": %d, \"ms\": %.4f, \"TBps\": %.3f}\n", grid, rpp, ms, bytes / ms / 1e9);
    }
  {
    float ms = median_ms([&] { hipMemsetAsync(b, 3, bytes, 0); });
    printf("{\"kernel\": \"hipMemsetAsync\", \"ms\": %.4f, \"TBps\": %.3f}\n", ms, bytes / ms / 1e9);
    ms = median_ms([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    printf("{\"kernel\": \"hipMemcpyAsync_d2d\", \"ms\": %.4f, \"TBps_read_plus_write\": %.3f}\n", ms, 2.0 * bytes / ms / 1e9);
  }
  // gathered rows: permutation of runs of consecutive 256-B rows
  const size_t rows = bytes / 256;
  std::vector<int> h(rows);
  int* idx;
  hipMalloc(&idx, rows * 4);
  for (int run : {1, 8, 64}) {
    const size_t nb = rows / run;
    std::vector<size_t> perm(nb);
    for (size_t i = 0; i < nb; ++i) perm[i] = i;
    uint64_t s = 88172645463325252ull;
    for (size_t i = nb - 1; i > 0; --i) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      std::swap(perm[i], perm[s % (i + 1)]);
    }
    for (size_t i = 0; i < nb; ++i)
      for (int k = 0; k < run; ++k) h[i * run + k] = (int)(perm[i] * run + k);
    hipMemcpy(idx, h.data(), rows * 4, hipMemcpyHostToDevice);
    float ms = median_ms([&] { hipLaunchKernelGGL(gather_kernel, dim3(4096), dim3(256), 0, 0, a, idx, sink, rows); });
    printf("{\"kernel\": \"gather_rows256\", \"run\": %d, \"ms\": %.4f, \"TBps\": %.3f}\n", run, ms, (bytes + rows * 4) / ms / 1e9);
  }
  return 0;
}
