// L2 -> CU ingest microbenchmark for gfx950: how many bytes per second can a CU pull from an L2-resident buffer
//   mode 0: buffer_load_dwordx4 ... lds   (LDS-DMA, 1 KiB per wave instruction, what the GEMM loaders use)
//   mode 1: global_load_dwordx4 into VGPRs (register staging)
// Each workgroup loops over a private 64 KiB window of a small (L2-resident) buffer, UNROLL loads in flight per wave.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/fill_bench.hip -o /tmp/fill_bench ; run on the MI355X box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE, int UNROLL>
__global__ __launch_bounds__(512) void fill_kernel(const uint4* __restrict__ src, float* __restrict__ sink, int iters, int window_vecs) {
  __shared__ __attribute__((aligned(16))) uint4 lds[4096];   // 64 KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const uint4* base = src + (size_t)(blockIdx.x % 32) * window_vecs;     // 32 windows: 2 MiB total, L2-resident
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(base), 0, window_vecs * 16, 0x00020000);
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int vec = ((it * UNROLL + u) * nw + wave) * 64 % window_vecs;       // wave-uniform start, 1 KiB per instruction
      if (MODE == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (wave * UNROLL + u) % 64 * 64),
                                                 16, (vec + lane) * 16, 0, 0, 0);
      } else {
        const uint4 v = base[vec + lane];
        acc += __uint_as_float(v.x ^ v.y ^ v.z ^ v.w);
      }
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (MODE == 0) acc = __uint_as_float(lds[threadIdx.x].x);
  if (acc == 123.456f) sink[0] = acc;
}

template <int MODE, int UNROLL>
static void run(const uint4* src, float* sink, int blocks, int threads, int iters) {
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  const int window_vecs = 4096;
  hipLaunchKernelGGL((fill_kernel<MODE, UNROLL>), dim3(blocks), dim3(threads), 0, 0, src, sink, 10, window_vecs);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL((fill_kernel<MODE, UNROLL>), dim3(blocks), dim3(threads), 0, 0, src, sink, iters, window_vecs);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms = 0;
  hipEventElapsedTime(&ms, s, e);
  const double bytes = (double)blocks * (threads / 64) * iters * UNROLL * 1024.0;
  printf("mode %d (%s) unroll %2d blocks %4d x %3d threads: %.2f TB/s chip, %.1f GB/s per CU (256 CUs)\n", MODE,
         MODE == 0 ? "lds-dma" : "vgpr   ", UNROLL, blocks, threads, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
  uint4* src; float* sink;
  hipMalloc(&src, 32 * 4096 * 16 + 4096);
  hipMalloc(&sink, 4);
  hipMemset(src, 1, 32 * 4096 * 16);
  for (int blocks : {256, 512, 1024}) {
    run<0, 4>(src, sink, blocks, 256, 2000);
    run<0, 8>(src, sink, blocks, 256, 1000);
    run<0, 8>(src, sink, blocks, 512, 1000);
    run<1, 4>(src, sink, blocks, 256, 2000);
    run<1, 8>(src, sink, blocks, 256, 1000);
    run<1, 8>(src, sink, blocks, 512, 1000);
  }
  return 0;
}
