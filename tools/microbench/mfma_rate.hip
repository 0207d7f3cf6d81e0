// MFMA issue-rate microbenchmark: v_mfma_f32_32x32x16_bf16 from registers only (no LDS, no memory), NACC independent
// accumulators in rotation, operands random bf16, 1 or 2 waves per SIMD.  Prints achieved TFLOP/s and cycles per MFMA per SIMD
// at the clock the chip sustains under this load.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC, int NB>
__global__ __launch_bounds__(256) void k(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
  bf16x8 a[NACC], b[NB];
  for (int i = 0; i < NACC; ++i) a[i] = in[threadIdx.x + 256 * i];
  for (int i = 0; i < NB; ++i) b[i] = in[threadIdx.x + 256 * (NACC + i)];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < NB; ++kk)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[kk], acc[i], 0, 0, 0);
  }
  float s = 0.0f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int NB>
static void run(const char* name, int blocks_per_cu, const bf16x8* in, float* out) {
  const int iters = 2000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, NB>), dim3(grid), dim3(256), 0, 0, in, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, NB>), dim3(grid), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)grid * 4 * iters * NACC * NB;      // wave-level MFMAs
  const double tf = mfma * 32768.0 / (ms * 1e-3) / 1e12;
  printf("%-34s %d wave(s)/SIMD: %.3f ms, %.0f TFLOP/s, %.1f ns per MFMA per SIMD\n", name, blocks_per_cu, ms, tf,
         ms * 1e6 / (mfma / 1024.0));
}

int main() {
  const size_t n = 256 * 64;
  bf16x8* in; float* out;
  hipMalloc(&in, n * sizeof(bf16x8)); hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  unsigned short* h = (unsigned short*)malloc(n * 16);
  for (size_t i = 0; i < n * 8; ++i) { float f = (float)rand() / RAND_MAX * 2.0f - 1.0f; unsigned u; memcpy(&u, &f, 4); h[i] = u >> 16; }
  hipMemcpy(in, h, n * 16, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<4, 4>("4 accumulators, random operands", 1, in, out);
    run<4, 4>("4 accumulators, random operands", 2, in, out);
    run<8, 2>("8 accumulators, random operands", 1, in, out);
    run<8, 2>("8 accumulators, random operands", 2, in, out);
  }
  hipMemset(in, 0, n * 16);
  run<4, 4>("4 accumulators, zero operands", 2, in, out);
  return 0;
}
