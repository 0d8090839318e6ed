// dev tool: which ingredient of the scan kernels costs what on top of the 8 KiB-per-wave streaming
// read: VALU work per tile, the LDS round trip, the bitmap store, the scattered value stores.
//   occupancy is limited with dynamic LDS (like the kernels' 9 KiB per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Opt { int work; int lds; int bitmap; int scatter; int chunked; };

template <int DEPTH>
__global__ __launch_bounds__(256) void tiles_kernel(const u32x4* __restrict__ in, int64_t tiles, Opt o,
                                                    uint32_t* out, uint32_t* bitmap, uint32_t* vals,
                                                    unsigned long long* clocks) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t stride = (int64_t)gridDim.x * 4;
  uint32_t acc = 0, bmacc[4] = {0, 0, 0, 0}, nacc = 0, nacc2 = 0, bmacc8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  u32x4 v[DEPTH][8];
  int64_t t = (int64_t)blockIdx.x * 4 + wave;
  if (o.chunked) {   // every wave streams its own contiguous run of tiles
    const int64_t per = (tiles + stride - 1) / stride;
    t = t * per; tiles = t + per < tiles ? t + per : tiles; stride = 1;
  }
  unsigned long long c0 = 0, w0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = clock64(); w0 = wall_clock64(); }
  uint32_t* slot = lds + wave * 2304;   // 9216 B per wave
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) {
    int64_t tt = t + d * stride;
    if (tt < tiles) {
#pragma unroll
      for (int u = 0; u < 8; ++u) v[d][u] = in[tt * 512 + u * 64 + lane];
    }
  }
  for (; t < tiles; t += stride * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int slot_in = (d + DEPTH - 1) % DEPTH;
      const int64_t tn = t + (int64_t)(d + DEPTH - 1) * stride;
      const int64_t tc = t + (int64_t)d * stride;
      if (tc >= tiles) break;
      uint32_t x = 0;
      if (o.lds) {
        // plane tile with odd word stride 33 (words of 8 B): lane u*64+lane holds words 2j,2j+1
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int wi = (u * 64 + lane) * 2;             // word index in tile
          const int blk = wi >> 5, pl = wi & 31;
          uint32_t* p = slot + (blk * 33 + pl) * 2;
          p[0] = v[d][u].x; p[1] = v[d][u].y; p[2] = v[d][u].z; p[3] = v[d][u].w;
        }
      }
      if (tn < tiles) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[slot_in][u] = in[tn * 512 + u * 64 + lane];
      }
      if (o.lds) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t* p = slot + ((lane >> 1) * 33) * 2 + (lane & 1);
#pragma unroll
        for (int i = 0; i < 32; ++i) x ^= p[i * 2] + i;
        __builtin_amdgcn_wave_barrier();
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) x ^= v[d][u].x ^ v[d][u].y ^ v[d][u].z ^ v[d][u].w;
      }
      for (int i = 0; i < o.work; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) x = __builtin_amdgcn_alignbit(x, x, 7) ^ (uint32_t)(i + k);   // 32 VALU per i
      }
      // bitmap variants: 1 = 256 B per tile; 2 = 1 KiB every 4th tile; 3 = 256 B per tile into an
      // L2-resident 1 MiB window
      if (o.bitmap == 1) bitmap[tc * 64 + lane] = x;
      else if (o.bitmap == 2) { bmacc[nacc++ & 3] = x; if ((nacc & 3) == 0) { u32x4 q = {bmacc[0], bmacc[1], bmacc[2], bmacc[3]}; *reinterpret_cast<u32x4*>(bitmap + (tc >> 2) * 256 + lane * 4) = q; } }
      else if (o.bitmap == 3) bitmap[(tc * 64 + lane) & 0x3FFFF] = x;
      else if (o.bitmap == 7) {   // unique 1 KiB per wave every 4th tile
        bmacc[nacc2++ & 3] = x;
        if ((nacc2 & 3) == 0) {
          const int64_t g = ((int64_t)blockIdx.x * 4 + wave), kk = (nacc2 - 1) / 4;
          u32x4 q = {bmacc[0], bmacc[1], bmacc[2], bmacc[3]};
          *reinterpret_cast<u32x4*>(bitmap + (kk * (int64_t)gridDim.x * 4 + g) * 256 + lane * 4) = q;
        }
      }
      else if (o.bitmap == 5 || o.bitmap == 6) {   // deferred: K dword stores back to back every K-th tile, each to its own tile's place
        const uint32_t K = o.bitmap == 5 ? 4 : 8;
        bmacc8[nacc2 & 7] = x; ++nacc2;
        if ((nacc2 & (K - 1)) == 0) {
#pragma unroll
          for (uint32_t k = 0; k < 8; ++k) if (k < K) bitmap[(tc - (int64_t)(K - 1 - k) * stride) * 64 + lane] = bmacc8[k];
        }
      }
      // value-store variants (all ~1 KiB per tile into the tile's own 8 KiB region):
      // 1 = 4 dword stores, lanes 12 B apart; 2 = 4 dword stores, each 256 B contiguous;
      // 3 = one dwordx4 store (1 KiB contiguous); 4 = like 1 but into an L2-resident window
      if (o.scatter == 1 || o.scatter == 4) {
        uint32_t* dst = (o.scatter == 1 ? vals + tc * 2048 : vals + (tc & 127) * 2048) + lane * 3;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) if (k < 3 || (x & 1)) dst[k] = x + k;
      } else if (o.scatter == 2) {
        uint32_t* dst = vals + tc * 2048 + lane;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) dst[k * 64] = x + k;
      } else if (o.scatter == 3) {
        u32x4 q = {x, x + 1, x + 2, x + 3};
        *reinterpret_cast<u32x4*>(vals + tc * 2048 + lane * 4) = q;
      } else if (o.scatter == 5) {   // dense: tile regions adjacent (1 KiB each)
        u32x4 q = {x, x + 1, x + 2, x + 3};
        *reinterpret_cast<u32x4*>(vals + tc * 256 + lane * 4) = q;
      } else if (o.scatter == 6) {   // dense, 4 KiB burst every 4th tile
        if ((++nacc & 3) == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { u32x4 q = {x, x + 1, x + 2, x + k}; *reinterpret_cast<u32x4*>(vals + (tc >> 2) * 1024 + k * 256 + lane * 4) = q; }
        }
      } else if (o.scatter == 8 || o.scatter == 9) {   // deferred: K x 1 KiB back to back every K-th tile, 8 KiB-stride regions
        const uint32_t K = o.scatter == 8 ? 4 : 8;
        if ((++nacc & (K - 1)) == 0) {
#pragma unroll
          for (uint32_t k = 0; k < 8; ++k) if (k < K) { u32x4 q = {x, x + 1, x + 2, x + k}; *reinterpret_cast<u32x4*>(vals + (tc - (int64_t)(K - 1 - k) * stride) * 2048 + lane * 4) = q; }
        }
      } else if (o.scatter == 10) {   // dense, 8 KiB burst every 8th tile
        if ((++nacc & 7) == 0) {
#pragma unroll
          for (int k = 0; k < 8; ++k) { u32x4 q = {x, x + 1, x + 2, x + k}; *reinterpret_cast<u32x4*>(vals + (tc >> 3) * 2048 + k * 256 + lane * 4) = q; }
        }
      } else if (o.scatter >= 11 && o.scatter <= 14) {   // unique, contiguous K KiB per wave every K-th tile (K = 2,4,8,16)
        const uint32_t K = 2u << (o.scatter - 11);
        ++nacc;
        if ((nacc & (K - 1)) == 0) {
          const int64_t g = ((int64_t)blockIdx.x * 4 + wave), kk = (nacc - 1) / K;
          uint32_t* dst = vals + ((kk * (int64_t)gridDim.x * 4 + g) * K) * 256;
          for (uint32_t k = 0; k < K; ++k) { u32x4 q = {x, x + 1, x + 2, x + k}; *reinterpret_cast<u32x4*>(dst + k * 256 + lane * 4) = q; }
        }
      } else if (o.scatter == 7) {   // 1 KiB per tile, regions 2 KiB apart
        u32x4 q = {x, x + 1, x + 2, x + 3};
        *reinterpret_cast<u32x4*>(vals + tc * 512 + lane * 4) = q;
      }
      acc ^= x;
    }
  }
  if (acc == 0x12345678u) { out[0] = acc; lds[threadIdx.x] = acc; }
  if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = clock64() - c0; clocks[1] = wall_clock64() - w0; }
}

__global__ void fill_random(uint64_t* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    p[i] = z;
  }
}

int main(int argc, char** argv) {
  const int64_t bytes = 1ll << 30;
  void* d; uint32_t *o, *bm, *vals; unsigned long long* clk;
  hipMalloc(&d, bytes); hipMalloc(&o, 64); hipMalloc(&bm, bytes / 32 + 4096); hipMalloc(&vals, bytes + 4096);
  hipMalloc(&clk, 16);
  hipMemset(d, 1, bytes);
  const bool random_data = argc > 1;
  if (random_data) { hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (uint64_t*)d, bytes / 8); hipDeviceSynchronize(); }
  printf("data: %s\n", random_data ? "random" : "constant 0x01");
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    float best = 1e9;
    for (int r = 0; r < 6; ++r) {
      hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-58s %8.1f us  %8.1f GB/s   clk %.0f MHz\n", name, best * 1e3, bytes / best / 1e6,
           h[1] ? (double)h[0] / ((double)h[1] / 100.0) : 0.0);
    fflush(stdout);
  };
  hipFuncSetAttribute((const void*)tiles_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)tiles_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const Opt opts[] = {
      {0, 0, 0, 0, 0},
      {4, 1, 1, 0, 0}, {4, 1, 7, 0, 0},
      {4, 1, 0, 3, 0}, {4, 1, 0, 11, 0}, {4, 1, 0, 12, 0}, {4, 1, 0, 13, 0}, {4, 1, 0, 14, 0},
      {4, 1, 7, 12, 0}, {4, 1, 7, 13, 0}, {12, 1, 7, 13, 0},
  };
  for (int wps : {4}) {
    const int lds = (160 * 1024 / wps) & ~1023;
    for (const Opt& op : opts) {
      const int grid = 256 * wps * 4;
      char nm[128];
      snprintf(nm, 128, "chunked=%d valu=%3d lds=%d bitmap=%d scatter=%d depth=1", op.chunked, op.work * 32, op.lds, op.bitmap, op.scatter);
      run(nm, [&] { hipLaunchKernelGGL(tiles_kernel<1>, dim3(grid), dim3(256), lds, 0, (const u32x4*)d, bytes / 8192, op, o, bm, vals, clk); });
      snprintf(nm, 128, "chunked=%d valu=%3d lds=%d bitmap=%d scatter=%d depth=2", op.chunked, op.work * 32, op.lds, op.bitmap, op.scatter);
      run(nm, [&] { hipLaunchKernelGGL(tiles_kernel<2>, dim3(grid), dim3(256), lds, 0, (const u32x4*)d, bytes / 8192, op, o, bm, vals, clk); });
    }
  }
  return 0;
}
