// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction of v_exp_f32 / v_rcp_f32 / v_fma_f32 / v_cvt_pk_bf16_f32 /
// v_max3_f32, alone and next to MFMAs of another accumulator, with 1, 2 and 3 waves per SIMD.  Backs the roofline argument
// for the head-dim-40 attention (32 exponentials per lane per 64-key tile against 14 MFMAs) in DESIGN.md.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int OP, int MF>
__global__ void k(float* out, long long* cyc, int iters) {
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = 0.001f * (threadIdx.x + i + 1);
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * i); b[i] = (__bf16)(0.02f * threadIdx.x); }
  f32x16 acc0 = {0}, acc1 = {0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (MF) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0); }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        else if (OP == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
        else if (OP == 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
        else if (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(x[i]));
        else if (OP == 4) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x[i]));
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  s += acc0[0] + acc1[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP, int MF>
static void run(const char* name, int waves_per_simd) {
  float* out; long long* cyc; hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000, threads = 256 * waves_per_simd;       // one workgroup per CU-sized slot: waves_per_simd waves on each SIMD
  hipLaunchKernelGGL((k<OP, MF>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((k<OP, MF>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per = (double)c / (iters * 4.0);
  printf("%-22s mfma=%d waves/SIMD=%d: %7.1f cycles per group of 8 VALU%s -> %5.1f per VALU instruction per wave%s\n", name, MF, waves_per_simd, per,
         MF ? " + 2 MFMA" : "", (per - (MF ? 0 : 0)) / 8.0, MF ? " (MFMA pair alone = 64)" : "");
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 3; ++w) {
    run<0, 0>("v_exp_f32", w); run<1, 0>("v_rcp_f32", w); run<2, 0>("v_fma_f32", w); run<3, 0>("v_cvt_pk_bf16_f32", w); run<4, 0>("v_max3_f32", w);
    run<0, 1>("v_exp_f32", w); run<2, 1>("v_fma_f32", w); run<5, 1>("(no VALU)", w);
  }
  return 0;
}
