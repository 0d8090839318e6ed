// dev tool: pure write and copy bandwidth with the 8 KiB-per-wave tile pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void write_kernel(u32x4* out, int64_t tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < tiles; t += stride) {
    u32x4 q = {(uint32_t)t, (uint32_t)lane, 3u, 4u};
#pragma unroll
    for (int u = 0; u < 8; ++u) out[t * 512 + u * 64 + lane] = q;
  }
}
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* in, u32x4* out, int64_t tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < tiles; t += stride) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = in[t * 512 + u * 64 + lane];
#pragma unroll
    for (int u = 0; u < 8; ++u) out[t * 512 + u * 64 + lane] = v[u];
  }
}
int main() {
  const int64_t bytes = 1ll << 30;
  void *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, double moved, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    float best = 1e9;
    for (int r = 0; r < 8; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    printf("%-28s %8.1f us  %8.1f GB/s\n", name, best * 1e3, moved / best / 1e6);
  };
  for (int grid : {1024, 4096}) {
    char nm[64];
    snprintf(nm, 64, "write 1 GiB grid=%d", grid);
    run(nm, (double)bytes, [&] { hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, (u32x4*)b, bytes / 8192); });
    snprintf(nm, 64, "copy 1 GiB grid=%d", grid);
    run(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, bytes / 8192); });
  }
  run("hipMemsetAsync 1 GiB", (double)bytes, [&] { hipMemsetAsync(b, 0, bytes, 0); });
  run("hipMemcpyDtoD 1 GiB", 2.0 * bytes, [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
  return 0;
}
