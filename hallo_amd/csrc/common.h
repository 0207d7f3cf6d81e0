// Shared device helpers for the gfx950 (CDNA4) kernels of the Hallo denoising path.
// Wave = 64 lanes; MFMA fragments follow the 32x32x16 layout:
//   A: lane l holds A[i = l&31][k = (l>>5)*8 .. +7]
//   B: lane l holds B[k = (l>>5)*8 .. +7][j = l&31]
//   D: lane l holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31], r = 0..15
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hallo {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

enum { DT_F16 = 0, DT_BF16 = 1 };
// ACT_GELU: erf GELU after the residual add (wav2vec2 feed-forward / conv feature layers); ACT_GELU_PRE: GELU of
// alpha * (acc + bias) BEFORE the residual add (wav2vec2 positional conv: hidden + gelu(conv(hidden))).
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_GELU = 3, ACT_GELU_PRE = 4 };

template <typename T> struct Vec;
template <> struct Vec<_Float16> {
  using v8 = f16x8;
  using v4 = f16x4;
  static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  // 16x16x32: A lane l holds A[i = l&15][k = (l>>4)*8 .. +7], B lane l holds B[k = (l>>4)*8 .. +7][j = l&15], D lane l holds D[i = (l>>4)*4 + r][j = l&15]
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Vec<__bf16> {
  using v8 = bf16x8;
  using v4 = bf16x4;
  static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

// packed 2-element dot product with fp32 accumulate (v_dot2c_f32_f16 / v_dot2c_f32_bf16): no conversions needed
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
__device__ __forceinline__ float dot2(f16x2 a, f16x2 b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }
__device__ __forceinline__ float dot2(bf16x2 a, bf16x2 b, float c) { return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false); }
// 8-element dot product of two 16-byte vectors
template <typename V8> __device__ __forceinline__ float dot8(V8 a, V8 b, float c) {
  c = dot2(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), c);
  c = dot2(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), c);
  c = dot2(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), c);
  c = dot2(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), c);
  return c;
}

template <typename T> __device__ __forceinline__ float to_f32(T x) { return static_cast<float>(x); }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return static_cast<T>(x); }

// 16-byte (8 x T) global/LDS access helpers
template <typename T> __device__ __forceinline__ typename Vec<T>::v8 ld8(const T* p) {
  return *reinterpret_cast<const typename Vec<T>::v8*>(p);
}
template <typename T> __device__ __forceinline__ void st8(T* p, typename Vec<T>::v8 v) {
  *reinterpret_cast<typename Vec<T>::v8*>(p) = v;
}
template <typename T> __device__ __forceinline__ typename Vec<T>::v8 zero8() {
  typename Vec<T>::v8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = static_cast<T>(0.0f);
  return z;
}

__device__ __forceinline__ float silu_f(float x) {   // x * sigmoid(x): v_exp + v_rcp (1 ulp), no IEEE division sequence
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// GELU (erf form, diffusers GEGLU: hallo/models/attention.py:601,905).  erf by Abramowitz-Stegun 7.1.26
// (|error| < 1.5e-7, far below the fp16 / bf16 output rounding): branch-free, 14 VALU ops against ~35 and a divergent
// branch for ocml's erff -- the GEGLU epilogue of the K = 320 feed-forward GEMMs is VALU-bound on this function.
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  const float erf_abs = __builtin_fmaf(-p * t, e, 1.0f);
  return 0.5f * x * (1.0f + __builtin_copysignf(erf_abs, x));
}

// GELU for the VALU-bound GEGLU epilogues (round 6 form).  With u = x * sqrt(log2(e) / 2):
//   gelu(x) = relu(x) - |x| * Phi(-|x|),   Phi(-|x|) = 2^P(|u|)
// P = degree-5 minimax fit of log2 Phi(-a) weighted by a * Phi(-a) (the sensitivity of the result), a in [0, 6.5 sigma]; the
// leading coefficient is negative and P decreases monotonically, so beyond the fit range 2^P underflows to 0 without a clamp.
// |gelu_u(u) - u * Phi(x)| <= 6.1e-7 over |u| <= 10 evaluated in fp32 (the A&S 7.1.26 form it replaces: 4.8e-7) -- three orders
// below the fp16 / bf16 rounding of the product it feeds.  9 VALU operations with ONE transcendental per element, against 13
// with two (rcp + exp2: quarter-rate instructions) -- the epilogues that call it are bound by VALU issue.
// gelu_u(u) returns gelu(x) / GELU_U_INV with GELU_U_INV = sqrt(2 / log2 e): the caller scales the argument by GELU_U_SCALE and
// the product by GELU_U_INV inside multiplications it performs anyway (LayerNorm affine of gate and value).
constexpr float GELU_U_SCALE = 0.84932180028801904272f;    // sqrt(log2(e) / 2)
constexpr float GELU_U_INV = 1.17741002251547469101f;      // 1 / GELU_U_SCALE
constexpr float GELU_P0 = -1.000037670135498f, GELU_P1 = -1.3549491167068481f, GELU_P2 = -0.6376851797103882f,
                GELU_P3 = -0.0845942497253418f, GELU_P4 = 0.0136150186881423f, GELU_P5 = -0.0010709537891671062f;
__device__ __forceinline__ float gelu_u(float u) {
  const float au = fabsf(u);
  float q = __builtin_fmaf(GELU_P5, au, GELU_P4);
  q = __builtin_fmaf(q, au, GELU_P3);
  q = __builtin_fmaf(q, au, GELU_P2);
  q = __builtin_fmaf(q, au, GELU_P1);
  q = __builtin_fmaf(q, au, GELU_P0);
  return __builtin_fmaf(-au, __builtin_amdgcn_exp2f(q), fmaxf(u, 0.0f));
}

// Two gelu_u at once on the packed fp32 pipe (v_pk_fma_f32: two IEEE fmas per issue slot): the same operations in the same order
// as gelu_u, element by element -- bit-identical results; |u|, max and the transcendental stay scalar.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 gelu_u2(f32x2 u) {
  const f32x2 au = {fabsf(u[0]), fabsf(u[1])};
  f32x2 q = pk_fma(f32x2{GELU_P5, GELU_P5}, au, f32x2{GELU_P4, GELU_P4});
  q = pk_fma(q, au, f32x2{GELU_P3, GELU_P3});
  q = pk_fma(q, au, f32x2{GELU_P2, GELU_P2});
  q = pk_fma(q, au, f32x2{GELU_P1, GELU_P1});
  q = pk_fma(q, au, f32x2{GELU_P0, GELU_P0});
  const f32x2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
  const f32x2 r = {fmaxf(u[0], 0.0f), fmaxf(u[1], 0.0f)};
  return pk_fma(-au, e, r);
}

// Sum over each aligned group of 8 lanes, result in all 8 (a fixed tree: neighbours, pairs, the two quads): three DPP moves, no
// LDS permute and no address registers -- the row-statistics epilogues run at the 128-VGPR boundary of the 128 x 128 GEMM kernel.
__device__ __forceinline__ float sum8_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror: lane i <-> 7 - i
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware remap of a linear workgroup id: the dispatcher places block b on XCD b%8;
// give each XCD a contiguous chunk of the tile space so neighbouring tiles share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace hallo

#define HALLO_CHECK_LAUNCH()                                    \
  do {                                                          \
    hipError_t e__ = hipGetLastError();                         \
    if (e__ != hipSuccess) return -(int)e__ - 1000;             \
  } while (0)
