// Per-CU intake probe (measurement only): how many bytes per clock a CU can pull from L2-resident memory
//   mode 0: global_load_dwordx4 -> VGPR      mode 1: global_load_lds 16 B/lane -> LDS      mode 2: global_load_lds 4 B/lane
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/intake_probe tools/probes/intake_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int U>
__global__ void probe(const char* __restrict__ src, size_t region, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const char* base = src + (size_t)blockIdx.x * region;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x4 acc = {0, 0, 0, 0};
  const size_t span = (size_t)nt * 16 * U;       // bytes one sweep of the block covers
  for (int it = 0; it < iters; ++it) {
    for (size_t off = 0; off + span <= region; off += span) {
      if (MODE == 0) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + off + ((size_t)u * nt + tid) * 16);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
      } else if (MODE == 1) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off + ((size_t)u * nt + tid) * 16),
                                           (__attribute__((address_space(3))) void*)(smem + (u & 3) * nt * 16 + wave * 1024), 16, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off + ((size_t)u * nt + tid) * 4),
                                           (__attribute__((address_space(3))) void*)(smem + (u & 3) * nt * 4 + wave * 256), 4, 0, 0);
      }
    }
  }
  if (MODE != 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc[0] = reinterpret_cast<float*>(smem)[tid];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

// GEMM-like pattern: one DMA piece = 8 rows x 128 B (row pitch `pitch` bytes), a workgroup of nt threads covers nt/8 rows,
// successive pieces walk along the rows (the K direction); SWZ permutes the 16-byte chunks inside a row like the GEMM does
template <int U, bool SWZ>
__global__ void probe_rows(const char* __restrict__ src, size_t pitch, int ksteps, int iters, float* sink, int nslots) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chunk = tid & 7, lrow = tid >> 3;
  const int cl = SWZ ? (chunk ^ ((lrow >> 1) & 7)) : chunk;
  const size_t rows_per_wg = (size_t)(nt / 8) * U;
  const char* base = src + (size_t)(blockIdx.x % nslots) * rows_per_wg * pitch + (size_t)lrow * pitch + cl * 16;
  for (int it = 0; it < iters; ++it)
    for (int k = 0; k < ksteps; ++k) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)u * (nt / 8) * pitch + (size_t)k * 128),
                                         (__attribute__((address_space(3))) void*)(smem + (u & 3) * nt * 16 + wave * 1024), 16, 0, 0);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (reinterpret_cast<float*>(smem)[tid] == 12345.678f) sink[0] = 1.f;
}

template <int U, bool SWZ>
static void run_rows(const char* d, float* sink, int wgs, int nt, size_t pitch, int iters, size_t total) {
  const size_t per_wg = (size_t)(nt / 8) * U * pitch;      // bytes one workgroup's rows span; every access stays inside slot*per_wg .. +per_wg
  const int nslots = (int)(total / per_wg);
  if (nslots < 1) { printf("skip pitch %zu\n", pitch); return; }
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int smem = 4 * nt * 16;
  const int ksteps = (int)(pitch / 128);
  hipLaunchKernelGGL((probe_rows<U, SWZ>), dim3(wgs), dim3(nt), smem, 0, d, pitch, ksteps, 2, sink, nslots);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe_rows<U, SWZ>), dim3(wgs), dim3(nt), smem, 0, d, pitch, ksteps, iters, sink, nslots);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)wgs * iters * ksteps * (double)nt * 16 * U;
  const double gbs = bytes / (ms * 1e-3) / 1e9;
  printf("rows8x128  U=%d swz=%d wgs=%4d threads=%4d pitch=%6zu (%zu KB/WG): %8.1f us  %8.1f GB/s total  (%5.1f B/clk/CU, %d WG/CU)\n", U, (int)SWZ, wgs, nt,
         pitch, (size_t)(nt / 8) * U * pitch >> 10, ms * 1e3, gbs, gbs / 256.0 / 2.4, (wgs + 255) / 256);
  hipEventDestroy(a); hipEventDestroy(b);
}

template <int MODE, int U>
static void run(const char* name, const char* d, float* sink, int wgs, int nt, size_t region, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int smem = 4 * nt * 16;
  hipLaunchKernelGGL((probe<MODE, U>), dim3(wgs), dim3(nt), smem, 0, d, region, 2, sink);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<MODE, U>), dim3(wgs), dim3(nt), smem, 0, d, region, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)wgs * iters * (double)(region / ((size_t)nt * 16 * U) * ((size_t)nt * 16 * U)) * (MODE == 2 ? 0.25 : 1.0);
  const double gbs = bytes / (ms * 1e-3) / 1e9;
  printf("%-10s U=%d wgs=%4d threads=%4d region=%6zuKB: %8.1f us  %8.1f GB/s total  %6.1f GB/s per WG  (%5.1f B/clk/CU at 2.4 GHz, %d WG/CU)\n", name, U, wgs,
         nt, region >> 10, ms * 1e3, gbs, gbs / wgs, gbs / 256.0 / 2.4, (wgs + 255) / 256);
  hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
  const size_t total = (size_t)1024 << 20;
  char* d; float* sink;
  hipMalloc(&d, total); hipMemset(d, 0, total); hipMalloc(&sink, 64);
  for (int wgs : {256, 512})
    for (size_t pitch : {(size_t)640, (size_t)2560, (size_t)5760, (size_t)23040}) {
      const int iters = pitch < 4096 ? 256 : 32;
      run_rows<4, false>(d, sink, wgs, 256, pitch, iters, total);
      run_rows<4, true>(d, sink, wgs, 256, pitch, iters, total);
      run_rows<7, true>(d, sink, wgs, 256, pitch, iters, total);
    }
  if (getenv("ROWS_ONLY")) return 0;
  for (int wgs : {256, 512, 1024}) {
    for (int nt : {256, 512, 1024}) {
      if ((size_t)wgs * nt > 1024 * 512) continue;
      for (size_t region : {(size_t)64 << 10, (size_t)1 << 20}) {
        const int iters = region > (256 << 10) ? 8 : 128;
        run<0, 4>("vgpr16", d, sink, wgs, nt, region, iters);
        run<0, 8>("vgpr16", d, sink, wgs, nt, region, iters);
        run<1, 4>("lds16", d, sink, wgs, nt, region, iters);
        run<1, 8>("lds16", d, sink, wgs, nt, region, iters);
        run<2, 8>("lds4", d, sink, wgs, nt, region, iters);
      }
    }
  }
  return 0;
}
