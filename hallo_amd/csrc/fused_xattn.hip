// Fused LayerNorm + few-token cross-attention + output projection + residual for gfx950.
//
// Replaces, per spatial transformer block and DDIM step (hallo/models/mutual_self_attention.py:286-303 on
// TemporalBasicTransformerBlock, class at hallo/models/attention.py:410-540):
//     norm_hidden_states = self.norm2(hidden_states)
//     hidden_states = self.attn2(norm_hidden_states, encoder_hidden_states=face tokens) + hidden_states
// i.e. nn.LayerNorm, Attention.to_q, SDPA over T = 4 face tokens x 8 heads, Attention.to_out[0] and the residual add.
//
// With only H*T = 32 (head, token) pairs the two projections collapse into per-clip constants:
//     score[row, (h,t)] = LN(x)[row] . Sw[(h,t)]         Sw[(h,t)][c] = c0 * sum_d Wq[h*hd+d, c] K[t, h*hd+d]
//     y[row]            = sum_(h,t) p[row,(h,t)] Ow[(h,t)] + bo + x[row]     Ow[(h,t)][c] = sum_d Wo[c, h*hd+d] V[t, h*hd+d]
// and LayerNorm folds into the first contraction (per-row mean / rstd applied to the 32 scores, not to C channels):
//     score = rstd * (x . (gamma * Sw) - mean * G) + B,   G = sum_c gamma_c Sw_c,  B = sum_c beta_c Sw_c.
// One pass over x: read a row once (+ an L2-resident re-read for the residual), write it once -- instead of
// LayerNorm (read + write), to_q GEMM (read + write), attention (read + write) and to_out GEMM (read + read + write).
//
// Work decomposition: a workgroup owns 32 rows and its 4 waves split the channel axis.  Both contractions run on the matrix pipe in the swapped form so that a
// lane owns a row:  S^T[32 (h,t)][32 rows] = Sg . X^T  (C/16 MFMAs, X fragments straight from global memory),
// softmax over the 4 tokens of a head is lane-local (the 4 values sit in one lane's registers), P stays in
// registers, O^T[32 channels][32 rows] = OwP . P^T (2 MFMAs per 32 output channels, OwP stored in the k-slot order
// that matches the accumulator layout of the first contraction).
#include "common.h"
#include "../../include/hallo_amd.h"

namespace hallo {

struct XattnArgs {
  const void* x; void* y;
  const void* sg;        // [nb][32][C]   T   gamma * Sw (scale and log2(e) folded in)
  const float* g;        // [nb][32]      fp32
  const float* b;        // [nb][32]      fp32
  const void* owp;       // [nb][C][32]   T   Ow permuted: [c][ks2][hi][e] <-> (h,t) = 8*(2*ks2 + (e>>2)) + 4*hi + (e&3)
  const void* bo;        // [C] T
  long rows;
  int C;
  long rows_per_batch;
  float eps;
};

template <typename T>
__global__ __launch_bounds__(256) void face_xattn_kernel(const XattnArgs p) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  // One workgroup = 32 rows; its 4 waves split the channel axis of BOTH contractions (the first over k slices, the
  // second over output-channel blocks), so a row block exposes 4x the memory parallelism of one wave walking C/16
  // dependent MFMAs -- at C = 1280 (256 px feature maps) there are only 128 row blocks for 256 CUs.
  __shared__ float s_part[4][16][64];      // partial scores^T of each wave, [wave][register][lane]
  __shared__ float s_stat[4][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const long row0 = (long)blockIdx.x * 32;
  const long row = min(row0 + l31, p.rows - 1);
  const bool live = row0 + l31 < p.rows;
  const int C = p.C;
  const long bidx = row0 / p.rows_per_batch;                 // block-uniform: rows_per_batch % 32 == 0
  const T* __restrict__ xr = reinterpret_cast<const T*>(p.x) + row * C;
  const T* __restrict__ sg = reinterpret_cast<const T*>(p.sg) + (bidx * 32 + l31) * C;
  const T* __restrict__ owp = reinterpret_cast<const T*>(p.owp) + bidx * C * 32;
  const T* __restrict__ bo = reinterpret_cast<const T*>(p.bo);
  T* __restrict__ yr = reinterpret_cast<T*>(p.y) + row * C;

  // ---- partial scores^T = Sg[:, slice] . X[:, slice]^T and partial row statistics over this wave's k slices ----
  f32x16 s;
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = 0.0f;
  float sum = 0.0f, sq = 0.0f;
  const int nks = C / 16;
#pragma unroll 5
  for (int ks = wave; ks < nks; ks += 4) {
    const int c0 = ks * 16 + hi * 8;
    const V8 xf = ld8<T>(xr + c0);
    const V8 sf = ld8<T>(sg + c0);
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float f = to_f32(xf[e]); sum += f; sq = __builtin_fmaf(f, f, sq); }
    s = Vec<T>::mfma32(sf, xf, s);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) s_part[wave][r][lane] = s[r];
  s_stat[wave][0][lane] = sum;
  s_stat[wave][1][lane] = sq;
  __syncthreads();
  // every wave combines the four partials in the same order -> identical P in all waves, no second barrier
  sum = (s_stat[0][0][lane] + s_stat[1][0][lane]) + (s_stat[2][0][lane] + s_stat[3][0][lane]);
  sq = (s_stat[0][1][lane] + s_stat[1][1][lane]) + (s_stat[2][1][lane] + s_stat[3][1][lane]);
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = (s_part[0][r][lane] + s_part[1][r][lane]) + (s_part[2][r][lane] + s_part[3][r][lane]);
  sum += __shfl_xor(sum, 32, 64);
  sq += __shfl_xor(sq, 32, 64);
  const float mean = sum / (float)C;
  const float var = fmaxf(sq / (float)C - mean * mean, 0.0f);
  const float rstd = rsqrtf(var + p.eps);

  // ---- softmax over the 4 tokens of each head: register r = 4g + jj holds (h,t) = 8g + 4hi + jj -> head 2g + hi ----
  const float* gv = p.g + bidx * 32;
  const float* bv = p.b + bidx * 32;
  V8 pf[2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float sc[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = 8 * g + 4 * hi + jj;
      sc[jj] = __builtin_fmaf(rstd, s[4 * g + jj] - mean * gv[j], bv[j]);
    }
    const float m = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
    float e4[4], l = 0.0f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) { e4[jj] = __builtin_amdgcn_exp2f(sc[jj] - m); l += e4[jj]; }
    const float inv = __builtin_amdgcn_rcpf(l);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) pf[g >> 1][(g & 1) * 4 + jj] = from_f32<T>(e4[jj] * inv);
  }

  // ---- O^T = OwP . P^T for this wave's 32-channel blocks, + bias + residual, 8-byte stores ----
  const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  const int ncb = C / 32;
#pragma unroll 2
  for (int cb = wave; cb < ncb; cb += 4) {
    const T* wrow = owp + (long)(cb * 32 + l31) * 32 + hi * 8;       // [c][ks2][hi][e]
    const V8 w0 = ld8<T>(wrow), w1 = ld8<T>(wrow + 16);
    f32x16 o = Vec<T>::mfma32(w0, pf[0], zero16);
    o = Vec<T>::mfma32(w1, pf[1], o);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = cb * 32 + 8 * g + 4 * hi;
      const V4 res = *reinterpret_cast<const V4*>(xr + c);
      const V4 bb = *reinterpret_cast<const V4*>(bo + c);
      V4 out;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) out[jj] = from_f32<T>(o[4 * g + jj] + to_f32(bb[jj]) + to_f32(res[jj]));
      if (live) *reinterpret_cast<V4*>(yr + c) = out;
    }
  }
}

}  // namespace hallo

using namespace hallo;

extern "C" int hallo_face_xattn(const void* x, void* y, const void* sg, const float* g, const float* b, const void* owp,
                                const void* bo, int64_t rows, int C, int64_t rows_per_batch, float eps, int dtype,
                                void* stream) {
  if (!x || !y || !sg || !g || !b || !owp || !bo || rows <= 0 || C <= 0 || (C & 31)) return -22;
  if (rows_per_batch <= 0 || (rows_per_batch & 31)) return -22;
  XattnArgs a;
  a.x = x; a.y = y; a.sg = sg; a.g = g; a.b = b; a.owp = owp; a.bo = bo;
  a.rows = rows; a.C = C; a.rows_per_batch = rows_per_batch; a.eps = eps;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((rows + 31) / 32)), block(256);
  if (dtype == DT_F16) hipLaunchKernelGGL((face_xattn_kernel<_Float16>), grid, block, 0, st, a);
  else if (dtype == DT_BF16) hipLaunchKernelGGL((face_xattn_kernel<__bf16>), grid, block, 0, st, a);
  else return -22;
  HALLO_CHECK_LAUNCH();
  return 0;
}
