// wav2vec2 audio front-end: the pieces of Wav2VecModel.forward
// (hallo/models/wav2vec.py:42-109,196-209) that are not plain GEMMs.
//
//   * hallo_w2v_conv0_gn_gelu: first feature-encoder layer of transformers' Wav2Vec2FeatureEncoder
//     (Wav2Vec2GroupNormConvLayer: Conv1d(1 -> C, k, stride, bias=False) -> GroupNorm(C groups over time) -> GELU).
//     One input channel makes this a VALU/HBM problem, not a GEMM: 2 k flops per output against 2 bytes written.
//     The convolution is therefore RECOMPUTED instead of stored: pass 1 reduces per-channel sum / sum of squares of
//     the fp32 conv output (deterministic: per-block partials, combined in block order in fp64), pass 2 recomputes
//     the conv, normalises, applies GELU and writes the token-major [L0, C] activation once, rounded once.
//     HBM traffic: waveform read twice (4 bytes per sample) + output written once.
//   * hallo_lerp_rows: linear_interpolation (wav2vec.py:196-209) = F.interpolate(mode="linear", align_corners=True)
//     along the time axis of a token-major [L, C] activation, with PyTorch's fp32 index arithmetic.
//
// The remaining layers (Conv1d k = 3 / 2, stride 2 as GEMMs over overlapping row windows, the grouped positional
// convolution, the 12 post-LN transformer layers) run on hallo_gemm / hallo_layernorm / hallo_softmax_rows.
#include "common.h"

namespace hallo {

constexpr int W2V_TL = 64;        // output positions per workgroup
constexpr int W2V_MAXK = 16;      // kernel taps
constexpr int W2V_MAXS = 8;       // stride

// Thread t owns the channel pairs {2 (t + 256 i), 2 (t + 256 i) + 1}; the block's input samples sit in LDS and every
// lane reads the same address (broadcast, conflict-free).  PASS 1: statistics, PASS 2: normalise + GELU + store.
template <int PASS, typename T>
__global__ __launch_bounds__(256) void w2v_conv0_kernel(const float* __restrict__ wave, long n_samples,
                                                        const float* __restrict__ w, const float* __restrict__ stats,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ partial, T* __restrict__ y, int L0, int C,
                                                        int k, int stride) {
  __shared__ float s_x[(W2V_TL - 1) * W2V_MAXS + W2V_MAXK];
  const int l0 = blockIdx.x * W2V_TL;
  const int nl = min(W2V_TL, L0 - l0);
  const int nx = (nl - 1) * stride + k;
  const long x0 = (long)l0 * stride;
  for (int i = threadIdx.x; i < nx; i += 256) s_x[i] = (x0 + i < n_samples) ? wave[x0 + i] : 0.0f;
  __syncthreads();
  typedef __attribute__((ext_vector_type(2))) T V2;
  for (int cp = threadIdx.x; 2 * cp < C; cp += 256) {
    const int c = 2 * cp;
    float w0[W2V_MAXK], w1[W2V_MAXK];
#pragma unroll
    for (int j = 0; j < W2V_MAXK; ++j) {
      w0[j] = (j < k) ? w[(long)c * k + j] : 0.0f;
      w1[j] = (j < k) ? w[(long)(c + 1) * k + j] : 0.0f;
    }
    float a0 = 0.0f, b0 = 0.0f, a1 = 0.0f, b1 = 0.0f;     // PASS 1: sum, sum of squares;  PASS 2: scale, shift
    if (PASS == 2) {
      const float m0 = stats[2 * c], r0 = stats[2 * c + 1], m1 = stats[2 * c + 2], r1 = stats[2 * c + 3];
      a0 = r0 * gamma[c]; b0 = beta[c] - m0 * a0;
      a1 = r1 * gamma[c + 1]; b1 = beta[c + 1] - m1 * a1;
    }
    for (int l = 0; l < nl; ++l) {
      const float* xs = s_x + l * stride;
      float v0 = 0.0f, v1 = 0.0f;
#pragma unroll
      for (int j = 0; j < W2V_MAXK; ++j) {
        if (j < k) {
          const float xv = xs[j];
          v0 = __builtin_fmaf(w0[j], xv, v0);
          v1 = __builtin_fmaf(w1[j], xv, v1);
        }
      }
      if (PASS == 1) {
        a0 += v0; b0 = __builtin_fmaf(v0, v0, b0);
        a1 += v1; b1 = __builtin_fmaf(v1, v1, b1);
      } else {
        const V2 o = {from_f32<T>(gelu_erf_f(__builtin_fmaf(v0, a0, b0))), from_f32<T>(gelu_erf_f(__builtin_fmaf(v1, a1, b1)))};
        *reinterpret_cast<V2*>(y + (long)(l0 + l) * C + c) = o;
      }
    }
    if (PASS == 1) {
      float* pp = partial + ((long)blockIdx.x * C + c) * 2;
      *reinterpret_cast<f32x4*>(pp) = f32x4{a0, b0, a1, b1};
    }
  }
}

// One thread per channel: combine the per-block partials in block order (fp64), emit (mean, rstd).
__global__ __launch_bounds__(256) void w2v_stats_finalize_kernel(const float* __restrict__ partial, int nblk, int C, int L0,
                                                                 float eps, float* __restrict__ stats) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int b = 0; b < nblk; ++b) {
    s += (double)partial[((long)b * C + c) * 2];
    q += (double)partial[((long)b * C + c) * 2 + 1];
  }
  const double mean = s / L0;
  double var = q / L0 - mean * mean;       // biased variance, as torch.nn.GroupNorm
  if (var < 0.0) var = 0.0;
  stats[2 * c] = (float)mean;
  stats[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// y[t, :] = (1 - f) x[i0, :] + f x[i1, :], src = t (in - 1) / (out - 1) evaluated in fp32 exactly as ATen's
// upsample_linear1d does (area_pixel_compute_scale / _source_index with align_corners = true).
template <typename T>
__global__ __launch_bounds__(256) void lerp_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int in_rows,
                                                        int out_rows, int C) {
  using V8 = typename Vec<T>::v8;
  const int vpr = C >> 3;
  const long v = (long)blockIdx.x * 256 + threadIdx.x;
  if (v >= (long)out_rows * vpr) return;
  const int t = (int)(v / vpr), cv = (int)(v - (long)t * vpr);
  const float scale = out_rows > 1 ? (float)(in_rows - 1) / (float)(out_rows - 1) : 0.0f;
  const float src = scale * (float)t;
  int i0 = (int)src;
  if (i0 > in_rows - 1) i0 = in_rows - 1;
  const int i1 = i0 + ((i0 < in_rows - 1) ? 1 : 0);
  const float f1 = src - (float)i0, f0 = 1.0f - f1;
  const V8 a = ld8<T>(x + (long)i0 * C + cv * 8), b = ld8<T>(x + (long)i1 * C + cv * 8);
  V8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = from_f32<T>(f0 * to_f32(a[j]) + f1 * to_f32(b[j]));
  st8<T>(y + (long)t * C + cv * 8, o);
}

static inline int w2v_len(long n_samples, int k, int stride) { return (int)((n_samples - k) / stride + 1); }

template <typename T>
static int launch_w2v_conv0(const float* wave, long n_samples, const float* w, const float* gamma, const float* beta,
                            void* y, float* ws, int C, int k, int stride, float eps, hipStream_t st) {
  const int L0 = w2v_len(n_samples, k, stride);
  const int nblk = (L0 + W2V_TL - 1) / W2V_TL;
  float* partial = ws;
  float* stats = ws + (long)nblk * C * 2;
  T* yy = reinterpret_cast<T*>(y);
  hipLaunchKernelGGL((w2v_conv0_kernel<1, T>), dim3(nblk), dim3(256), 0, st, wave, n_samples, w, nullptr, gamma, beta,
                     partial, yy, L0, C, k, stride);
  HALLO_CHECK_LAUNCH();
  hipLaunchKernelGGL(w2v_stats_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, partial, nblk, C, L0, eps, stats);
  HALLO_CHECK_LAUNCH();
  hipLaunchKernelGGL((w2v_conv0_kernel<2, T>), dim3(nblk), dim3(256), 0, st, wave, n_samples, w, stats, gamma, beta,
                     partial, yy, L0, C, k, stride);
  HALLO_CHECK_LAUNCH();
  return 0;
}

}  // namespace hallo

using namespace hallo;

extern "C" int64_t hallo_w2v_conv0_workspace(int64_t n_samples, int C, int k, int stride) {
  if (n_samples < k || C <= 0 || k <= 0 || stride <= 0) return -22;
  const int64_t L0 = (n_samples - k) / stride + 1;
  const int64_t nblk = (L0 + W2V_TL - 1) / W2V_TL;
  return (nblk * C * 2 + (int64_t)C * 2) * 4;
}

extern "C" int hallo_w2v_conv0_gn_gelu(const float* wave, int64_t n_samples, const float* w, const float* gamma,
                                       const float* beta, void* y, float* workspace, int C, int k, int stride, float eps,
                                       int dtype, void* stream) {
  if (!wave || !w || !gamma || !beta || !y || !workspace) return -22;
  if (C <= 0 || (C & 7) || k <= 0 || k > W2V_MAXK || stride <= 0 || stride > W2V_MAXS || n_samples < k) return -22;
  if ((n_samples - k) / stride + 1 > 0x7FFFFFFFL) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DT_F16) return launch_w2v_conv0<_Float16>(wave, n_samples, w, gamma, beta, y, workspace, C, k, stride, eps, st);
  if (dtype == DT_BF16) return launch_w2v_conv0<__bf16>(wave, n_samples, w, gamma, beta, y, workspace, C, k, stride, eps, st);
  return -22;
}

extern "C" int hallo_lerp_rows(const void* x, void* y, int in_rows, int out_rows, int C, int dtype, void* stream) {
  if (!x || !y || in_rows <= 0 || out_rows <= 0 || C <= 0 || (C & 7)) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long n = (long)out_rows * (C >> 3);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (dtype == DT_F16)
    hipLaunchKernelGGL((lerp_rows_kernel<_Float16>), grid, block, 0, st, reinterpret_cast<const _Float16*>(x),
                       reinterpret_cast<_Float16*>(y), in_rows, out_rows, C);
  else if (dtype == DT_BF16)
    hipLaunchKernelGGL((lerp_rows_kernel<__bf16>), grid, block, 0, st, reinterpret_cast<const __bf16*>(x),
                       reinterpret_cast<__bf16*>(y), in_rows, out_rows, C);
  else
    return -22;
  HALLO_CHECK_LAUNCH();
  return 0;
}
