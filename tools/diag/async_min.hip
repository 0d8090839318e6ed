// Diagnostic only: the allocation pattern of the ips_dict_encode variant that crashed facade_test
// (commit d0b066e), reduced to HIP runtime calls, one mode per process.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "segv_trace.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
__global__ void touch(unsigned long long* t, unsigned* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { t[i] = i; if (i == 0) c[0] = 7; }
}
int main(int argc, char** argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 0;
  hipStream_t s = nullptr;
  if (mode == 2) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  void* plain = nullptr;
  CK(hipMalloc(&plain, 1 << 20));
  for (int rep = 0; rep < 3; ++rep) {
    unsigned long long* t = nullptr; unsigned* c = nullptr; unsigned* a = nullptr; unsigned* b = nullptr;
    CK(hipMallocAsync((void**)&t, 131072 * 8, s));
    CK(hipMallocAsync((void**)&c, 8, s));
    CK(hipMallocAsync((void**)&a, 131072 * 4, s));
    CK(hipMallocAsync((void**)&b, 70001 * 4 + 16, s));
    CK(hipMemsetAsync(t, 0xFF, 131072 * 8, s));
    CK(hipMemsetAsync(c, 0, 8, s));
    hipLaunchKernelGGL(touch, dim3(512), dim3(256), 0, s, t, c, 131072);
    unsigned counters[2] = {0, 0};
    CK(hipMemcpyAsync(counters, c, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> host(131072);
    CK(hipMemcpy(host.data(), t, 131072 * 8, hipMemcpyDeviceToHost));
    CK(hipFreeAsync(t, s)); CK(hipFreeAsync(c, s)); CK(hipFreeAsync(a, s)); CK(hipFreeAsync(b, s));
    printf("mode %d rep %d ok: counter %u, table[5] %llu\n", mode, rep, counters[0], host[5]);
    if (mode == 3) { void* q = nullptr; CK(hipMalloc(&q, 4096)); CK(hipFree(q)); }
  }
  if (mode == 1) CK(hipDeviceSynchronize());
  CK(hipFree(plain));
  printf("mode %d: returning from main\n", mode);
  return 0;
}
