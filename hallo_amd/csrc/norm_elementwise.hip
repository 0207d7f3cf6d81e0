// HBM-bound operators of the Hallo denoising path on token-major [frames, H*W, C] activations:
// GroupNorm(+SiLU), LayerNorm(+positional encoding), row softmax, strided copies, layout
// conversion at the pipeline boundary, timestep embedding and the fused CFG + DDIM update.
// All loads/stores are 16-byte vectors (8 x fp16/bf16); statistics are fp32 (combined in fp64).
#include "common.h"
#include "../../include/hallo_amd.h"
#include <string.h>

namespace hallo {

// ------------------------------------------------------------------------------------------
// GroupNorm, pass 1: per-(frame, chunk) partial sum / sum of squares for each group.
// grid (chunks, n_img); every thread owns one fixed 8-channel vector column and strides rows.
// ------------------------------------------------------------------------------------------
constexpr int GN_MAX_GROUPS = 32;

// Two-source input (round 6): the C channels of a row are the C1 channels of x followed by the C - C1 channels of x2 -- the skip
// concatenation `torch.cat([hidden_states, res_hidden_states], dim=1)` in front of an up-block resnet (hallo/models/unet_3d_blocks.py:
// 1131,1373) read in place instead of materialised.  A thread owns whole 8-channel vector columns and C1 % 8 == 0, so a column lies in one
// source: its base pointer and row pitch are picked once.  x2 == nullptr / C1 == C: one source.
template <typename T>
struct GnSrc { const T* base; long pitch; };
template <typename T>
__device__ __forceinline__ GnSrc<T> gn_src(const T* x, const T* x2, int C, int C1, long img_rows, int c0) {
  if (c0 < C1) return {x + img_rows * C1 + c0, (long)C1};
  return {x2 + img_rows * (C - C1) + (c0 - C1), (long)(C - C1)};
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, const T* __restrict__ x2, int C1, float* __restrict__ ws,
                                                       int HW, int C, int groups, int rows_per_chunk) {
  using V8 = typename Vec<T>::v8;
  // Deterministic block reduction (no float atomics: results are bit-reproducible run to run).  A thread owns at most
  // 2 vector columns (C <= 4096) and a vector of 8 channels spans at most 3 groups (channels per group >= 4), so every
  // thread publishes up to 2 x 3 (group, sum, sumsq) runs, which are then added in a fixed order.
  __shared__ float s_rs[6][256], s_rq[6][256];
  __shared__ int s_rg[6][256];
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, img = blockIdx.y, nchunks = gridDim.x;
  const int vpr = C / 8;
  const int cpg = C / groups;
#pragma unroll
  for (int r = 0; r < 6; ++r) s_rg[r][tid] = -1;
  const int r_begin = chunk * rows_per_chunk;
  const int r_end = min(HW, r_begin + rows_per_chunk);
  // vector columns are distributed over threads; threads beyond a multiple of vpr take extra rows
  const int tpr = vpr <= 256 ? vpr : 256;          // threads that span one row
  const int row_lanes = 256 / tpr;                 // rows processed concurrently
  const int my_row = tid / tpr, my_v0 = tid % tpr;
  int nrun = 0;
  if (my_row < row_lanes) {
    for (int v = my_v0; v < vpr; v += tpr) {
      float a[8], q[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { a[e] = 0.0f; q[e] = 0.0f; }
      const GnSrc<T> src = gn_src<T>(x, x2, C, C1, (long)img * HW, v * 8);
      const T* base = src.base;
      const long P = src.pitch;
      int r = r_begin + my_row;
      // 4 independent 16-byte loads in flight per thread (the loop is latency-bound otherwise)
      for (; r + 3 * row_lanes < r_end; r += 4 * row_lanes) {
        V8 v0 = ld8<T>(base + (long)r * P);
        V8 v1 = ld8<T>(base + (long)(r + row_lanes) * P);
        V8 v2 = ld8<T>(base + (long)(r + 2 * row_lanes) * P);
        V8 v3 = ld8<T>(base + (long)(r + 3 * row_lanes) * P);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f0 = to_f32(v0[e]), f1 = to_f32(v1[e]), f2 = to_f32(v2[e]), f3 = to_f32(v3[e]);
          a[e] += (f0 + f1) + (f2 + f3);
          q[e] += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
        }
      }
      for (; r < r_end; r += row_lanes) {
        V8 val = ld8<T>(base + (long)r * P);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = to_f32(val[e]); a[e] += f; q[e] += f * f; }
      }
      // the 8 channels of a vector are consecutive: runs of equal group index are combined before publishing
      // vector k of this thread owns slots 3k .. (a thread with ONE vector may use all 6: channels per group >= 2;
      // two vectors per thread only occur for C > 2048, i.e. >= 64 channels per group and <= 2 runs each)
      nrun = 3 * ((v - my_v0) / tpr);
      int g_run = (v * 8) / cpg;
      float sa = 0.0f, sq = 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = (v * 8 + e) / cpg;
        if (g != g_run) {
          if (nrun < 6) { s_rg[nrun][tid] = g_run; s_rs[nrun][tid] = sa; s_rq[nrun][tid] = sq; }
          ++nrun;
          sa = 0.0f; sq = 0.0f; g_run = g;
        }
        sa += a[e];
        sq += q[e];
      }
      if (nrun < 6) { s_rg[nrun][tid] = g_run; s_rs[nrun][tid] = sa; s_rq[nrun][tid] = sq; }
      ++nrun;
    }
  }
  __syncthreads();
  // stage 1: the row lanes of a vector column hold runs of the same groups -> row lane 0 adds them in lane order
  if (row_lanes > 1 && my_row == 0) {
    for (int rl = 1; rl < row_lanes; ++rl) {
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        if (s_rg[r][my_v0] >= 0) { s_rs[r][my_v0] += s_rs[r][rl * tpr + my_v0]; s_rq[r][my_v0] += s_rq[r][rl * tpr + my_v0]; }
      }
    }
  }
  __syncthreads();
  // stage 2: one thread per group walks the vector columns that overlap its channels, in column order
  if (tid < groups) {
    const int g = tid;
    const int v_lo = (g * cpg) >> 3, v_hi = ((g + 1) * cpg - 1) >> 3;
    float sacc = 0.0f, qacc = 0.0f;
    for (int v = v_lo; v <= v_hi; ++v) {
      const int t = v % tpr;                            // column v is held by thread v % tpr ...
      const int r_lo = vpr > tpr ? (v / tpr) * 3 : 0, r_hi = vpr > tpr ? r_lo + 3 : 6;   // ... in these slots
      for (int r = r_lo; r < r_hi; ++r) {
        if (s_rg[r][t] == g) { sacc += s_rs[r][t]; qacc += s_rq[r][t]; }
      }
    }
    float* o = ws + (((long)img * nchunks + chunk) * groups + g) * 2;
    o[0] = sacc;
    o[1] = qacc;
  }
}

// GroupNorm, pass 2: finalise the statistics (fp64 combine) and apply scale/shift (+SiLU).
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const T* __restrict__ x2, int C1, T* __restrict__ y,
                                                       const T* __restrict__ gamma, const T* __restrict__ beta,
                                                       const float* __restrict__ ws, int HW, int C, int groups,
                                                       int nchunks, int rows_per_block, float eps, int silu) {
  using V8 = typename Vec<T>::v8;
  __shared__ float s_mean[GN_MAX_GROUPS], s_rstd[GN_MAX_GROUPS];
  __shared__ double s_part[8][GN_MAX_GROUPS][2];
  const int tid = threadIdx.x;
  const int img = blockIdx.y;
  const int cpg = C / groups;
  {
    // 8 partial sums per group in parallel, then one thread per group combines them (fp64)
    const int g = tid & 31, part = tid >> 5;
    double s = 0.0, q = 0.0;
    if (g < groups) {
      for (int c = part; c < nchunks; c += 8) {
        const float* w = ws + (((long)img * nchunks + c) * groups + g) * 2;
        s += (double)w[0];
        q += (double)w[1];
      }
    }
    s_part[part][g][0] = s;
    s_part[part][g][1] = q;
  }
  __syncthreads();
  if (tid < groups) {
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s += s_part[k][tid][0]; q += s_part[k][tid][1]; }
    const double n = (double)HW * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    s_mean[tid] = (float)mean;
    s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  // thread -> (row lane, channel vector): gamma/beta/mean/rstd of a vector are folded into y = x * sc + sh ONCE,
  // the row loop is load, 8 FMA (+SiLU), store
  const int vpr = C / 8;
  const int tpr = vpr <= 256 ? vpr : 256;
  const int row_lanes = 256 / tpr;
  const int my_row = tid / tpr, my_v0 = tid % tpr;
  const long r_begin = (long)blockIdx.x * rows_per_block;
  const long r_end = min((long)HW, r_begin + rows_per_block);
  if (my_row < row_lanes) {
    for (int v = my_v0; v < vpr; v += tpr) {
      const int c0 = v * 8;
      V8 g8 = ld8<T>(gamma + c0), b8 = ld8<T>(beta + c0);
      float sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = (c0 + e) / cpg;
        sc[e] = s_rstd[g] * to_f32(g8[e]);
        sh[e] = to_f32(b8[e]) - s_mean[g] * sc[e];
      }
      const GnSrc<T> src = gn_src<T>(x, x2, C, C1, (long)img * HW, c0);
      const T* xb = src.base;
      T* yb = y + ((long)img * HW) * C + c0;
      for (long r = r_begin + my_row; r < r_end; r += row_lanes) {
        V8 val = ld8<T>(xb + r * src.pitch);
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float f = __builtin_fmaf(to_f32(val[e]), sc[e], sh[e]);
          if (silu) f = silu_f(f);
          o[e] = from_f32<T>(f);
        }
        st8<T>(yb + r * C, o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// GroupNorm for small feature maps (H*W <= 1024: the 32x32 / 16x16 / 8x8 levels), ONE launch: a workgroup owns a
// channel slice of whole groups of one image, sums it (pass 1), reduces in a fixed order, then re-reads the slice
// (L2-resident: <= 160 KB) and writes the normalised values (pass 2).  The two-kernel path above costs ~20 us on these
// shapes whatever their size (two launches, the fp64 combine in every apply block); slices of one image are placed
// on one XCD (xcd_remap) so that the 128-byte lines shared by neighbouring slices are fetched once.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_fused_kernel(const T* __restrict__ x, const T* __restrict__ x2, int C1, T* __restrict__ y,
                                                       const T* __restrict__ gamma, const T* __restrict__ beta,
                                                       int HW, int C, int cpg, int cb, float eps, int silu) {
  using V8 = typename Vec<T>::v8;
  __shared__ float s_rs[3][256], s_rq[3][256];
  __shared__ int s_rg[3][256];
  __shared__ float s_cs[3][16], s_cq[3][16];      // per (slot, vector column) sums over the row lanes
  __shared__ float s_mean[16], s_rstd[16];
  const int tid = threadIdx.x;
  const int nslices = C / cb;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int img = bid / nslices, slice = bid - img * nslices;
  const int vps = cb / 8;                           // vector columns of the slice (<= 16)
  const int lanes = 256 / vps;                      // row lanes
  const int vi = tid % vps, rl = tid / vps;
  const bool act = rl < lanes;
  const int c0 = slice * cb + vi * 8;               // first channel of this thread's vector
  const int gpb = cb / cpg;                         // groups in the slice
  const GnSrc<T> src = gn_src<T>(x, x2, C, C1, (long)img * HW, act ? c0 : 0);
  const T* xb = src.base;
  const long P = src.pitch;                          // row pitch of this thread's source
  T* yb = y + ((long)img * HW) * C + c0;

  float a[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = 0.0f; q[e] = 0.0f; }
  if (act) {
    int r = rl;
    for (; r + 3 * lanes < HW; r += 4 * lanes) {
      const V8 v0 = ld8<T>(xb + (long)r * P), v1 = ld8<T>(xb + (long)(r + lanes) * P);
      const V8 v2 = ld8<T>(xb + (long)(r + 2 * lanes) * P), v3 = ld8<T>(xb + (long)(r + 3 * lanes) * P);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f0 = to_f32(v0[e]), f1 = to_f32(v1[e]), f2 = to_f32(v2[e]), f3 = to_f32(v3[e]);
        a[e] += (f0 + f1) + (f2 + f3);
        q[e] += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
      }
    }
    for (; r < HW; r += lanes) {
      const V8 v0 = ld8<T>(xb + (long)r * P);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = to_f32(v0[e]); a[e] += f; q[e] += f * f; }
    }
  }
  // publish <= 3 runs of equal (slice-local) group index per thread (channels per group >= 4)
  s_rg[0][tid] = -1; s_rg[1][tid] = -1; s_rg[2][tid] = -1;
  if (act) {
    int nrun = 0, g_run = (vi * 8) / cpg;
    float sa = 0.0f, sq = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (vi * 8 + e) / cpg;
      if (g != g_run) {
        if (nrun < 3) { s_rg[nrun][tid] = g_run; s_rs[nrun][tid] = sa; s_rq[nrun][tid] = sq; }
        ++nrun; sa = 0.0f; sq = 0.0f; g_run = g;
      }
      sa += a[e];
      sq += q[e];
    }
    if (nrun < 3) { s_rg[nrun][tid] = g_run; s_rs[nrun][tid] = sa; s_rq[nrun][tid] = sq; }
  }
  __syncthreads();
  if (tid < 3 * vps) {                               // (slot, column) reducers: row lanes added in lane order
    const int slot = tid / vps, col = tid - slot * vps;
    float sa = 0.0f, sq = 0.0f;
    if (s_rg[slot][col] >= 0) {
      for (int l = 0; l < lanes; ++l) { sa += s_rs[slot][l * vps + col]; sq += s_rq[slot][l * vps + col]; }
    }
    s_cs[slot][col] = sa;
    s_cq[slot][col] = sq;
  }
  __syncthreads();
  if (tid < gpb) {                                   // one thread per group: columns in order
    const int g = tid;
    const int v_lo = (g * cpg) >> 3, v_hi = ((g + 1) * cpg - 1) >> 3;
    double sa = 0.0, sq = 0.0;
    for (int v = v_lo; v <= v_hi; ++v) {
#pragma unroll
      for (int slot = 0; slot < 3; ++slot) {
        if (s_rg[slot][v] == g) { sa += (double)s_cs[slot][v]; sq += (double)s_cq[slot][v]; }
      }
    }
    const double n = (double)HW * cpg;
    const double mean = sa / n;
    double var = sq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    s_mean[g] = (float)mean;
    s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  if (!act) return;
  float sc[8], sh[8];
  {
    const V8 g8 = ld8<T>(gamma + c0), b8 = ld8<T>(beta + c0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (vi * 8 + e) / cpg;
      sc[e] = s_rstd[g] * to_f32(g8[e]);
      sh[e] = to_f32(b8[e]) - s_mean[g] * sc[e];
    }
  }
  auto apply8 = [&](V8 val) {
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = __builtin_fmaf(to_f32(val[e]), sc[e], sh[e]);
      if (silu) f = silu_f(f);
      o[e] = from_f32<T>(f);
    }
    return o;
  };
  int r = rl;
  for (; r + 3 * lanes < HW; r += 4 * lanes) {
    const V8 v0 = ld8<T>(xb + (long)r * P), v1 = ld8<T>(xb + (long)(r + lanes) * P);
    const V8 v2 = ld8<T>(xb + (long)(r + 2 * lanes) * P), v3 = ld8<T>(xb + (long)(r + 3 * lanes) * P);
    st8<T>(yb + (long)r * C, apply8(v0));
    st8<T>(yb + (long)(r + lanes) * C, apply8(v1));
    st8<T>(yb + (long)(r + 2 * lanes) * C, apply8(v2));
    st8<T>(yb + (long)(r + 3 * lanes) * C, apply8(v3));
  }
  for (; r < HW; r += lanes) st8<T>(yb + (long)r * C, apply8(ld8<T>(xb + (long)r * P)));
}

// ------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (C <= 8*64*MAXV), two-pass statistics.
// ------------------------------------------------------------------------------------------
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                        const T* __restrict__ gamma, const T* __restrict__ beta,
                                                        const float* __restrict__ pe, long rows, int C, float eps,
                                                        int pe_rpp, int pe_len) {
  using V8 = typename Vec<T>::v8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int vpr = C / 8;
  const T* xr = x + row * C;
  float v[MAXV][8];
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = lane + 64 * i;
    if (vi < vpr) {
      V8 t = ld8<T>(xr + vi * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = to_f32(t[e]); sum += v[i][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.0f;
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = lane + 64 * i;
    if (vi < vpr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
  const float* per = pe ? pe + (long)((row / pe_rpp) % pe_len) * C : nullptr;
  T* yr = y + row * C;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = lane + 64 * i;
    if (vi < vpr) {
      V8 g8 = ld8<T>(gamma + vi * 8), b8 = ld8<T>(beta + vi * 8);
      float pe8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (per) {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(per + vi * 8), p1 = *reinterpret_cast<const f32x4*>(per + vi * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { pe8[e] = p0[e]; pe8[4 + e] = p1[e]; }
      }
      V8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (v[i][e] - mean) * rstd * to_f32(g8[e]) + to_f32(b8[e]);
        if (per) {
          // the reference adds the PE to the already-rounded LayerNorm output (motion_module.py:459)
          f = to_f32(from_f32<T>(f)) + pe8[e];
        }
        o[e] = from_f32<T>(f);
      }
      st8<T>(yr + vi * 8, o);
    }
  }
}

// LayerNorm for C = 40 * LPR (320 / 640 / 1280: every LayerNorm width of the UNets): LPR = 8 / 16 / 32 lanes share a
// row, 5 x 16-byte vectors per lane, so all 64 lanes of a wave load (the one-wave-per-row kernel above runs 40 of 64
// lanes at C = 320) and a wave keeps 5 independent loads per lane in flight over 64 / LPR rows.
template <typename T, int LPR>
__global__ __launch_bounds__(256) void layernorm_sub_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                            const T* __restrict__ gamma, const T* __restrict__ beta,
                                                            const float* __restrict__ pe, long rows, int C, float eps,
                                                            int pe_rpp, int pe_len) {
  using V8 = typename Vec<T>::v8;
  constexpr int RPW = 64 / LPR;             // rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (LPR - 1);
  const long row = ((long)blockIdx.x * 4 + wave) * RPW + lane / LPR;
  const bool live = row < rows;
  const T* xr = x + (live ? row : 0) * C;
  float v[5][8];
  float sum = 0.0f;
  V8 t[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) t[i] = ld8<T>(xr + (sub + LPR * i) * 8);
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[i][e] = to_f32(t[i][e]); sum += v[i][e]; }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)C;
  float sq = 0.0f;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = rsqrtf(sq / (float)C + eps);
  if (!live) return;
  const float* per = pe ? pe + (long)((row / pe_rpp) % pe_len) * C : nullptr;
  T* yr = y + row * C;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int c0 = (sub + LPR * i) * 8;
    V8 g8 = ld8<T>(gamma + c0), b8 = ld8<T>(beta + c0);
    float pe8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (per) {
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(per + c0), p1 = *reinterpret_cast<const f32x4*>(per + c0 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { pe8[e] = p0[e]; pe8[4 + e] = p1[e]; }
    }
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = (v[i][e] - mean) * rstd * to_f32(g8[e]) + to_f32(b8[e]);
      if (per) f = to_f32(from_f32<T>(f)) + pe8[e];   // PE is added to the already-rounded LayerNorm output (motion_module.py:459)
      o[e] = from_f32<T>(f);
    }
    st8<T>(yr + c0, o);
  }
}

// Row statistics only (mean, rstd) -- LayerNorm whose affine is folded into the consuming GEMM (hallo_gemm ln_colsum /
// ln_stats): LPR lanes share a row, VPL 16-byte vectors per lane, two-pass statistics in registers, one 8-byte store.
template <typename T, int LPR, int VPL>
__global__ __launch_bounds__(256) void row_stats_kernel(const T* __restrict__ x, float2* __restrict__ st, long rows, int C,
                                                        float eps) {
  using V8 = typename Vec<T>::v8;
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (LPR - 1);
  const long row = ((long)blockIdx.x * 4 + wave) * RPW + lane / LPR;
  const bool live = row < rows;
  const T* xr = x + (live ? row : 0) * C;
  const int vpr = C / 8;
  float v[VPL][8];
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vi = sub + LPR * i;
    if (vi < vpr) {
      const V8 t = ld8<T>(xr + vi * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = to_f32(t[e]); sum += v[i][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.0f;
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)C;
  float sq = 0.0f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    if (sub + LPR * i < vpr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  if (live && sub == 0) st[row] = make_float2(mean, rsqrtf(sq / (float)C + eps));
}

// ------------------------------------------------------------------------------------------
// Row softmax (fp32 in, T out): one workgroup per row.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, T* __restrict__ y,
                                                           int cols, float scale_log2e) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (long)blockIdx.x * cols;
  T* yr = y + (long)blockIdx.x * cols;
  float mx = -1e30f;
  for (int c = tid * 4; c < cols; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale_log2e;
  __syncthreads();
  float sum = 0.0f;
  for (int c = tid * 4; c < cols; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    sum += exp2f(v.x * scale_log2e - mx) + exp2f(v.y * scale_log2e - mx) + exp2f(v.z * scale_log2e - mx) +
           exp2f(v.w * scale_log2e - mx);
  }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int c = tid * 4; c < cols; c += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    typename Vec<T>::v4 o;
    o[0] = from_f32<T>(exp2f(v.x * scale_log2e - mx) * inv);
    o[1] = from_f32<T>(exp2f(v.y * scale_log2e - mx) * inv);
    o[2] = from_f32<T>(exp2f(v.z * scale_log2e - mx) * inv);
    o[3] = from_f32<T>(exp2f(v.w * scale_log2e - mx) * inv);
    *reinterpret_cast<typename Vec<T>::v4*>(yr + c) = o;
  }
}

// ------------------------------------------------------------------------------------------
// Strided 2-D copy in 16-byte vectors (skip concat / motion-frame concat).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy2d_kernel(const uint4* __restrict__ src, long src_pitch_v,
                                                     uint4* __restrict__ dst, long dst_pitch_v, long rows,
                                                     int width_v) {
  const long total = rows * width_v;
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
    const long r = u / width_v;
    const int c = (int)(u - r * width_v);
    dst[r * dst_pitch_v + c] = src[r * src_pitch_v + c];
  }
}

// ------------------------------------------------------------------------------------------
// Boundary layout conversion.
// ------------------------------------------------------------------------------------------
template <typename T, typename S>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const S* __restrict__ x, T* __restrict__ y, int C,
                                                           int HW, int Cpad) {
  // grid (ceil(HW/256), n); small C (3, 4, 8): each thread writes one token's Cpad channels
  const int n = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  T* o = y + ((long)n * HW + p) * Cpad;
  for (int c = 0; c < Cpad; ++c) {
    const float f = c < C ? (float)x[((long)n * C + c) * HW + p] : 0.0f;
    o[c] = from_f32<T>(f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_f32_kernel(const T* __restrict__ x, float* __restrict__ y,
                                                               int C, int HW, long ldx, float mul, float add,
                                                               float lo, float hi) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const T* i = x + ((long)n * HW + p) * ldx;
  for (int c = 0; c < C; ++c) {
    float f = to_f32(i[c]) * mul + add;
    f = fminf(fmaxf(f, lo), hi);
    y[((long)n * C + c) * HW + p] = f;
  }
}

// ------------------------------------------------------------------------------------------
// Output path: planar fp32 frames [F, 3, H*W] in [0, 1] -> interleaved uint8 frames [F, H*W, 3]
// = np.clip(x * 255, 0, 255).astype(np.uint8) of tensor_to_video (hallo/utils/util.py:308-312): fp32 multiply,
// clamp, truncation toward zero -- bit-exact with the numpy expression on the same fp32 input.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frames_to_uint8_kernel(const float* __restrict__ x, uint8_t* __restrict__ y,
                                                              int C, long HW) {
  const long f = blockIdx.y;
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  for (int c = 0; c < C; ++c) {
    float v = __fmul_rn(x[(f * C + c) * HW + p], 255.0f);
    v = fminf(fmaxf(v, 0.0f), 255.0f);          // NaN -> 0 like np.clip + astype on this platform is undefined; inputs are clamped
    y[(f * HW + p) * C + c] = (uint8_t)v;       // float -> integer conversion truncates toward zero
  }
}

// ------------------------------------------------------------------------------------------
// Timestep embedding: [cos | sin] (flip_sin_to_cos=True, downscale_freq_shift=0).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void timestep_embedding_kernel(const float* __restrict__ t, T* __restrict__ out, int dim) {
  const int b = blockIdx.x;
  const int half = dim / 2;
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    const float freq = expf(-logf(10000.0f) * (float)i / (float)half);
    const float a = t[b] * freq;
    out[(long)b * dim + i] = from_f32<T>(cosf(a));
    out[(long)b * dim + half + i] = from_f32<T>(sinf(a));
  }
}

// ------------------------------------------------------------------------------------------
// CFG combine + DDIM step (eta = 0; v / epsilon / sample prediction, optional clip_sample), fp32 latents updated in place.
// `flags`: bit 0 CFG, bit 1 epsilon prediction, bit 2 sample prediction (neither: v-prediction), bit 3 clip x0 to [-1, 1].
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cfg_ddim_kernel(const T* __restrict__ mo, long ldm, float* __restrict__ lat,
                                                       T* __restrict__ nxt, long ldn, long rows, int C, int flags,
                                                       float gs, float sa_t, float sb_t, float sa_p, float sb_p) {
  const long total = rows * C;
  const int cfg = flags & 1;
  for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
    const long r = u / C;
    const int c = (int)(u - r * C);
    float v;
    if (cfg) {
      const float vu = to_f32(mo[r * ldm + c]);
      const float vc = to_f32(mo[(rows + r) * ldm + c]);
      v = vu + gs * (vc - vu);
    } else {
      v = to_f32(mo[r * ldm + c]);
    }
    const float xs = lat[u];
    float x0, ep;
    if (flags & 2) { ep = v; x0 = (xs - sb_t * v) / sa_t; }            // epsilon prediction
    else if (flags & 4) { x0 = v; ep = (xs - sa_t * v) / sb_t; }       // sample prediction
    else { x0 = sa_t * xs - sb_t * v; ep = sa_t * v + sb_t * xs; }     // v-prediction
    if (flags & 8) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);                 // clip_sample (range 1.0); eps is NOT recomputed
    const float xp = sa_p * x0 + sb_p * ep;
    lat[u] = xp;
    if (nxt) {
      const T o = from_f32<T>(xp);
      nxt[r * ldn + c] = o;
      if (cfg) nxt[(rows + r) * ldn + c] = o;
    }
  }
}

}  // namespace hallo

using namespace hallo;

extern "C" int hallo_abi_version(void) { return 9; }   // v9 (round 6): hallo_gemm_desc.kv_out (head-major K / V), hallo_attn_desc.kv1_hs / kv2_hs; v8 (round 6): hallo_temporal_attention_lead; v7 (round 5): hallo_gemm_desc.row_parts / ln_parts, hallo_face_xattn_stats

static int g_gn_fused = 1;   // hallo_set_option("gn_fused", 0 | 1): single-launch GroupNorm for small feature maps

extern "C" int hallo_set_option_attn(const char* name, int value);   // attention.hip

extern "C" int hallo_set_option_norm(const char* name, int value) {
  if (name && !strcmp(name, "gn_fused")) { if (value < 0 || value > 1) return -22; g_gn_fused = value; return 0; }
  return hallo_set_option_attn(name, value);
}

extern "C" int hallo_get_option_xattn(const char* name);            // fused_xattn.hip

extern "C" int hallo_get_option_norm(const char* name) {
  if (name && !strcmp(name, "gn_fused")) return g_gn_fused;
  return hallo_get_option_xattn(name);
}

extern "C" int hallo_groupnorm_chunks(int HW) {
  int c = (HW + 63) / 64;   // 64 rows per partial-statistics block: >= 1024 blocks at 16 x 64x64 frames
  if (c > 64) c = 64;
  if (c < 1) c = 1;
  return c;
}

template <typename T>
static int launch_groupnorm(const void* x, const void* x2, int C1, void* y, const void* gamma, const void* beta, float* ws, int n_img,
                            int HW, int C, int groups, float eps, int silu, hipStream_t st) {
  if (g_gn_fused && HW <= 1024) {
    // channel slice per workgroup: whole groups, a multiple of 8 channels, 40..128 channels
    const int cpg = C / groups;
    int cb = 0;
    for (int k = 1; cpg >= 4 && k * cpg <= 128; ++k) {
      if ((k * cpg) % 8 == 0 && groups % k == 0 && k * cpg >= 40) { cb = k * cpg; break; }
    }
    if (cb > 0 && (long)(C / cb) * n_img >= 16) {
      hipLaunchKernelGGL((gn_fused_kernel<T>), dim3((unsigned)((C / cb) * n_img)), dim3(256), 0, st,
                         reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(x2), C1, reinterpret_cast<T*>(y), reinterpret_cast<const T*>(gamma),
                         reinterpret_cast<const T*>(beta), HW, C, cpg, cb, eps, silu);
      HALLO_CHECK_LAUNCH();
      return 0;
    }
  }
  const int nchunks = hallo_groupnorm_chunks(HW);
  const int rpc = (HW + nchunks - 1) / nchunks;
  hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(nchunks, n_img), dim3(256), 0, st, reinterpret_cast<const T*>(x),
                     reinterpret_cast<const T*>(x2), C1, ws, HW, C, groups, rpc);
  // apply: ~16 KB of data per 256-thread block (measured: 32-64 KB blocks with 4 loads in flight are 25 % slower --
  // fewer, longer blocks lose more to the tail than the per-block statistics prologue costs)
  int rows_per_block = (8192 * 2) / (C * 2);
  if (rows_per_block < 1) rows_per_block = 1;
  const int nb = (HW + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(nb, n_img), dim3(256), 0, st, reinterpret_cast<const T*>(x),
                     reinterpret_cast<const T*>(x2), C1, reinterpret_cast<T*>(y), reinterpret_cast<const T*>(gamma), reinterpret_cast<const T*>(beta), ws,
                     HW, C, groups, nchunks, rows_per_block, eps, silu);
  HALLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int hallo_groupnorm_nhwc2(const void* x, int C1, const void* x2, void* y, const void* gamma, const void* beta, float* workspace,
                                     int n_img, int HW, int C, int groups, float eps, int silu, int dtype, void* stream);

extern "C" int hallo_groupnorm_nhwc(const void* x, void* y, const void* gamma, const void* beta, float* workspace,
                                    int n_img, int HW, int C, int groups, float eps, int silu, int dtype,
                                    void* stream) {
  return hallo_groupnorm_nhwc2(x, C, nullptr, y, gamma, beta, workspace, n_img, HW, C, groups, eps, silu, dtype, stream);
}

extern "C" int hallo_groupnorm_nhwc2(const void* x, int C1, const void* x2, void* y, const void* gamma, const void* beta, float* workspace,
                                     int n_img, int HW, int C, int groups, float eps, int silu, int dtype, void* stream) {
  if (!x || !y || !gamma || !beta || !workspace) return -22;
  if (C1 <= 0 || C1 > C || (C1 & 7) || (C1 < C && !x2)) return -22;
  if (C1 == C) x2 = nullptr;
  if (n_img <= 0 || HW <= 0 || C <= 0 || (C & 7) || groups <= 0 || groups > GN_MAX_GROUPS || C % groups) return -22;
  if (n_img > 65535) return -22;
  // the deterministic reduction gives a thread 6 (group, sum, sumsq) slots: one vector of 8 channels spanning <= 5
  // groups (channels per group >= 2), or two vectors (C > 2048) of <= 3 groups each
  if (C > 4096 || C / groups < 2) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DT_F16) return launch_groupnorm<_Float16>(x, x2, C1, y, gamma, beta, workspace, n_img, HW, C, groups, eps, silu, st);
  if (dtype == DT_BF16) return launch_groupnorm<__bf16>(x, x2, C1, y, gamma, beta, workspace, n_img, HW, C, groups, eps, silu, st);
  return -22;
}

template <typename T>
static int launch_layernorm(const void* x, void* y, const void* g, const void* b, const float* pe, long rows, int C,
                            float eps, int rpp, int plen, hipStream_t st) {
  const int vpr = C / 8;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const T* xx = reinterpret_cast<const T*>(x);
  T* yy = reinterpret_cast<T*>(y);
  const T* gg = reinterpret_cast<const T*>(g);
  const T* bb = reinterpret_cast<const T*>(b);
  if (vpr == 40 || vpr == 80 || vpr == 160) {
    const int lpr = vpr / 5, rpb = 4 * (64 / lpr);
    dim3 g2((unsigned)((rows + rpb - 1) / rpb));
    if (lpr == 8) hipLaunchKernelGGL((layernorm_sub_kernel<T, 8>), g2, block, 0, st, xx, yy, gg, bb, pe, rows, C, eps, rpp, plen);
    else if (lpr == 16) hipLaunchKernelGGL((layernorm_sub_kernel<T, 16>), g2, block, 0, st, xx, yy, gg, bb, pe, rows, C, eps, rpp, plen);
    else hipLaunchKernelGGL((layernorm_sub_kernel<T, 32>), g2, block, 0, st, xx, yy, gg, bb, pe, rows, C, eps, rpp, plen);
    HALLO_CHECK_LAUNCH();
    return 0;
  }
  if (vpr <= 64) hipLaunchKernelGGL((layernorm_kernel<T, 1>), grid, block, 0, st, xx, yy, gg, bb, pe, rows, C, eps, rpp, plen);
  else if (vpr <= 128) hipLaunchKernelGGL((layernorm_kernel<T, 2>), grid, block, 0, st, xx, yy, gg, bb, pe, rows, C, eps, rpp, plen);
  else if (vpr <= 192) hipLaunchKernelGGL((layernorm_kernel<T, 3>), grid, block, 0, st, xx, yy, gg, bb, pe, rows, C, eps, rpp, plen);
  else return -22;
  HALLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int hallo_layernorm(const void* x, void* y, const void* gamma, const void* beta, const float* pe, int rows,
                               int C, float eps, int pe_rows_per_pos, int pe_len, int dtype, void* stream) {
  if (!x || !y || !gamma || !beta || rows <= 0 || C <= 0 || (C & 7) || C > 1536) return -22;
  if (pe && (pe_rows_per_pos <= 0 || pe_len <= 0)) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rpp = pe ? pe_rows_per_pos : 1, plen = pe ? pe_len : 1;
  if (dtype == DT_F16) return launch_layernorm<_Float16>(x, y, gamma, beta, pe, rows, C, eps, rpp, plen, st);
  if (dtype == DT_BF16) return launch_layernorm<__bf16>(x, y, gamma, beta, pe, rows, C, eps, rpp, plen, st);
  return -22;
}

extern "C" int hallo_frames_to_uint8(const float* x, uint8_t* y, int frames, int channels, int64_t hw, void* stream) {
  if (!x || !y || frames <= 0 || channels <= 0 || channels > 4 || hw <= 0) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(frames_to_uint8_kernel, dim3((unsigned)((hw + 255) / 256), frames), dim3(256), 0, st, x, y, channels, (long)hw);
  HALLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int hallo_row_stats(const void* x, float* stats, int64_t rows, int C, float eps, int dtype, void* stream) {
  if (!x || !stats || rows <= 0 || C <= 0 || (C & 7) || C > 1536) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int vpr = C / 8;
  float2* o = reinterpret_cast<float2*>(stats);
#define HALLO_RS(TT, LPRv, VPLv) hipLaunchKernelGGL((row_stats_kernel<TT, LPRv, VPLv>), dim3((unsigned)((rows + 4 * (64 / LPRv) - 1) / (4 * (64 / LPRv)))), \
    dim3(256), 0, st, reinterpret_cast<const TT*>(x), o, (long)rows, C, eps)
#define HALLO_RS_T(TT)                                                   \
  do {                                                                   \
    if (vpr <= 40) HALLO_RS(TT, 8, 5);          /* C <= 320 */           \
    else if (vpr <= 80) HALLO_RS(TT, 16, 5);    /* C <= 640 */           \
    else if (vpr <= 160) HALLO_RS(TT, 32, 5);   /* C <= 1280 */          \
    else HALLO_RS(TT, 64, 3);                   /* C <= 1536 */          \
  } while (0)
  if (dtype == DT_F16) HALLO_RS_T(_Float16);
  else if (dtype == DT_BF16) HALLO_RS_T(__bf16);
  else return -22;
#undef HALLO_RS_T
#undef HALLO_RS
  HALLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int hallo_softmax_rows(const float* x, void* y, int rows, int cols, float scale, int dtype, void* stream) {
  if (!x || !y || rows <= 0 || cols <= 0 || (cols & 3)) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float sl = scale * 1.4426950408889634f;
  if (dtype == DT_F16)
    hipLaunchKernelGGL((softmax_rows_kernel<_Float16>), dim3(rows), dim3(256), 0, st, x, reinterpret_cast<_Float16*>(y), cols, sl);
  else if (dtype == DT_BF16)
    hipLaunchKernelGGL((softmax_rows_kernel<__bf16>), dim3(rows), dim3(256), 0, st, x, reinterpret_cast<__bf16*>(y), cols, sl);
  else return -22;
  HALLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int hallo_copy2d(const void* src, int64_t src_pitch, void* dst, int64_t dst_pitch, int64_t rows, int width,
                            int dtype, void* stream) {
  (void)dtype;  // both storage types are 2 bytes
  if (!src || !dst || rows <= 0 || width <= 0) return -22;
  if ((width & 7) || (src_pitch & 7) || (dst_pitch & 7)) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long total = rows * (width / 8);
  long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const uint4*>(src),
                     (long)(src_pitch / 8), reinterpret_cast<uint4*>(dst), (long)(dst_pitch / 8), (long)rows, width / 8);
  HALLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int hallo_nchw_to_nhwc(const void* x, void* y, int n, int C, int HW, int Cpad, int src_f32, int dtype,
                                  void* stream) {
  if (!x || !y || n <= 0 || C <= 0 || HW <= 0 || Cpad < C || n > 65535) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((HW + 255) / 256, n), block(256);
  if (dtype == DT_F16) {
    if (src_f32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<_Float16, float>), grid, block, 0, st, reinterpret_cast<const float*>(x), reinterpret_cast<_Float16*>(y), C, HW, Cpad);
    else hipLaunchKernelGGL((nchw_to_nhwc_kernel<_Float16, _Float16>), grid, block, 0, st, reinterpret_cast<const _Float16*>(x), reinterpret_cast<_Float16*>(y), C, HW, Cpad);
  } else if (dtype == DT_BF16) {
    if (src_f32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<__bf16, float>), grid, block, 0, st, reinterpret_cast<const float*>(x), reinterpret_cast<__bf16*>(y), C, HW, Cpad);
    else hipLaunchKernelGGL((nchw_to_nhwc_kernel<__bf16, __bf16>), grid, block, 0, st, reinterpret_cast<const __bf16*>(x), reinterpret_cast<__bf16*>(y), C, HW, Cpad);
  } else return -22;
  HALLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int hallo_nhwc_to_nchw_f32(const void* x, float* y, int n, int C, int HW, int64_t ldx, float mul, float add,
                                      float lo, float hi, int dtype, void* stream) {
  if (!x || !y || n <= 0 || C <= 0 || HW <= 0 || ldx < C || n > 65535) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((HW + 255) / 256, n), block(256);
  if (dtype == DT_F16) hipLaunchKernelGGL((nhwc_to_nchw_f32_kernel<_Float16>), grid, block, 0, st, reinterpret_cast<const _Float16*>(x), y, C, HW, (long)ldx, mul, add, lo, hi);
  else if (dtype == DT_BF16) hipLaunchKernelGGL((nhwc_to_nchw_f32_kernel<__bf16>), grid, block, 0, st, reinterpret_cast<const __bf16*>(x), y, C, HW, (long)ldx, mul, add, lo, hi);
  else return -22;
  HALLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int hallo_timestep_embedding(const float* t, void* out, int batch, int dim, int dtype, void* stream) {
  if (!t || !out || batch <= 0 || dim <= 0 || (dim & 1)) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DT_F16) hipLaunchKernelGGL((timestep_embedding_kernel<_Float16>), dim3(batch), dim3(256), 0, st, t, reinterpret_cast<_Float16*>(out), dim);
  else if (dtype == DT_BF16) hipLaunchKernelGGL((timestep_embedding_kernel<__bf16>), dim3(batch), dim3(256), 0, st, t, reinterpret_cast<__bf16*>(out), dim);
  else return -22;
  HALLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int hallo_cfg_ddim_step(const void* model_out, int64_t ldm, float* latents, void* next_in, int64_t ldn,
                                   int rows, int C, int cfg, float guidance_scale, float alpha_t, float alpha_prev,
                                   int dtype, void* stream) {
  if (!model_out || !latents || rows <= 0 || C <= 0 || ldm < C || (next_in && ldn < C)) return -22;
  if (cfg < 0 || cfg > 15 || (cfg & 6) == 6) return -22;
  if (alpha_t < 0.0f || alpha_t > 1.0f || alpha_prev < 0.0f || alpha_prev > 1.0f) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float sa_t = sqrtf(alpha_t), sb_t = sqrtf(1.0f - alpha_t);
  const float sa_p = sqrtf(alpha_prev), sb_p = sqrtf(1.0f - alpha_prev);
  const long total = (long)rows * C;
  long nb = (total + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (dtype == DT_F16)
    hipLaunchKernelGGL((cfg_ddim_kernel<_Float16>), dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const _Float16*>(model_out), (long)ldm, latents, reinterpret_cast<_Float16*>(next_in), (long)ldn, (long)rows, C, cfg, guidance_scale, sa_t, sb_t, sa_p, sb_p);
  else if (dtype == DT_BF16)
    hipLaunchKernelGGL((cfg_ddim_kernel<__bf16>), dim3((unsigned)nb), dim3(256), 0, st, reinterpret_cast<const __bf16*>(model_out), (long)ldm, latents, reinterpret_cast<__bf16*>(next_in), (long)ldn, (long)rows, C, cfg, guidance_scale, sa_t, sb_t, sa_p, sb_p);
  else return -22;
  HALLO_CHECK_LAUNCH();
  return 0;
}
