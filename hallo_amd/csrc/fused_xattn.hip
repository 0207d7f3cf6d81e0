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
#include <string.h>

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
  float* stats_out;      // optional [rows][2] fp32: (mean, rstd) of the OUTPUT rows, eps = stats_eps (the next nn.LayerNorm's statistics)
  float stats_eps;
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


// ------------------------------------------------------------------------------------------------------------------------
// Round 3: the same operator for C = 320 / 640 / 1280 with the activations staged through LDS.
//
// The kernel above reads x in the MFMA operand form: a lane owns a ROW and fetches 16 bytes of it per k slice, so one wave
// instruction touches 32 rows x 32 bytes -- 32 partial cache lines -- and the output leaves in 8-byte pieces at a row stride
// (the write pattern that cost attention40 1.4x its output bytes before its epilogue was changed).  Timed cold it moves 84 MB in
// 51 us at C = 320 (1.6 TB/s, a device copy does the same bytes in 18) and 21 MB in 37 us at C = 1280.
//
// Here a workgroup walks 32-row blocks (persistent: blocks b, b + grid, ...).  Per block:
//   1. x[32][C] -> LDS by 16-byte loads, consecutive lanes = consecutive 16-byte pieces of a row (whole 128-byte lines), rows at
//      a pitch of 2C + 16 bytes (the 16 rows of a ds_read_b128 lane group fall on 16 distinct 16-byte bank groups);
//   2. scores^T = Sg . X^T with X fragments from LDS; Sg (and further down OwP) fragments live in REGISTERS for the whole kernel
//      -- their row-per-lane loads are issued once per workgroup, not once per 32 rows;
//   3. softmax as above; O^T = OwP . P^T + bias + residual (the residual comes from the LDS tile), written back INTO the tile;
//   4. tile -> y by 16-byte stores of whole lines.
// k slices and channel blocks are dealt to the 4 waves exactly as above and partial sums are combined in the same order, so the
// two kernels agree bit for bit (tests/test_ops_gpu.py compares them).
// ------------------------------------------------------------------------------------------------------------------------
// PF (round 5): the next block's x loads are issued right after this block went to LDS and fly under its MFMA / softmax work.
template <typename T, int C, bool PF>
__global__ __launch_bounds__(256, C == 320 ? (PF ? 2 : 3) : (C == 640 ? 2 : 1)) void face_xattn_tiled_kernel(const XattnArgs p, int nblocks) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  constexpr int PITCH = C * 2 + 16;              // bytes
  constexpr int NKS = C / 16, NKW = NKS / 4;     // k slices, per wave (C / 64: 5, 10, 20)
  constexpr int NCB = C / 32, NCW = (NCB + 3) / 4;
  constexpr int CPR = C / 8;                     // 16-byte pieces per row
  constexpr int NLD = 32 * CPR / 256;            // pieces per thread (C / 64)
  static_assert(NKS % 4 == 0 && (32 * CPR) % 256 == 0, "C must be a multiple of 64");
  __shared__ __attribute__((aligned(16))) unsigned char tile[32 * PITCH];
  __shared__ float s_part[4][16][64];
  __shared__ float s_stat[4][2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const T* __restrict__ bo = reinterpret_cast<const T*>(p.bo);
  const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

  V8 sgf[NKW];               // Sg[(h,t) = l31][ks * 16 + hi * 8 ..], ks = wave + 4 j
  V8 ow0[NCW], ow1[NCW];     // OwP[c = cb * 32 + l31][hi * 8 ..] and [16 + hi * 8 ..], cb = wave + 4 j
  __shared__ float s_gb[2][32];                                // G / B of the current batch (LDS: 32 registers fewer than copies per lane)
  __shared__ float s_ostat[4][2][64];                          // stats_out: per-wave partial (sum, sum of squares) of the output rows
  long cur_b = -1;
  constexpr bool SPLIT_STAGE = !PF && C <= 640;      // stage a block in two halves (see step 1)
  V8 st[SPLIT_STAGE ? (NLD + 1) / 2 : NLD];
  auto load_block = [&](int blk_) {
    const T* xb = reinterpret_cast<const T*>(p.x);
    const long r0 = (long)blk_ * 32;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = tid + 256 * i, r = q / CPR, cc = q - r * CPR;
      st[i] = ld8<T>(xb + min(r0 + r, p.rows - 1) * C + cc * 8);
    }
  };
  if (PF && (int)blockIdx.x < nblocks) load_block(blockIdx.x);

  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const long row0 = (long)blk * 32;
    const long bidx = row0 / p.rows_per_batch;                 // block-uniform: rows_per_batch % 32 == 0
    if (bidx != cur_b) {                                       // (re)load the per-batch constants: once per workgroup unless it crosses batches
      cur_b = bidx;
      const T* sg = reinterpret_cast<const T*>(p.sg) + (bidx * 32 + l31) * C;
      const T* owp = reinterpret_cast<const T*>(p.owp) + bidx * C * 32;
#pragma unroll
      for (int j = 0; j < NKW; ++j) sgf[j] = ld8<T>(sg + (wave + 4 * j) * 16 + hi * 8);
#pragma unroll
      for (int j = 0; j < NCW; ++j) {
        const int cb = wave + 4 * j;
        if (cb < NCB) {
          const T* wrow = owp + (long)(cb * 32 + l31) * 32 + hi * 8;
          ow0[j] = ld8<T>(wrow);
          ow1[j] = ld8<T>(wrow + 16);
        }
      }
      __syncthreads();                                         // previous block's readers of s_gb are done (block-uniform branch)
      if (threadIdx.x < 32) { s_gb[0][threadIdx.x] = p.g[bidx * 32 + threadIdx.x]; s_gb[1][threadIdx.x] = p.b[bidx * 32 + threadIdx.x]; }
    }

    // ---- 1. x block -> LDS ----
    if (!PF && SPLIT_STAGE) {
      // C = 320 without prefetch / C = 640: the NLD staging registers on top of the resident Sg / OwP fragments do not fit the
      // register budget of the occupancy cap (13-15 VGPRs spilled to scratch in round 5, hipcc -S).  Two half-blocks, each
      // with all of ITS loads in flight: no scratch, and the block is still in flight as two groups of NLD / 2 16-byte loads.
      const T* xb = reinterpret_cast<const T*>(p.x);
      constexpr int H0 = NLD / 2;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int i0 = half ? H0 : 0, i1 = half ? NLD : H0;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
          const int q = tid + 256 * i, r = q / CPR, cc = q - r * CPR;
          st[i - i0] = ld8<T>(xb + min(row0 + r, p.rows - 1) * C + cc * 8);
        }
#pragma unroll
        for (int i = i0; i < i1; ++i) {
          const int q = tid + 256 * i, r = q / CPR, cc = q - r * CPR;
          *reinterpret_cast<V8*>(tile + r * PITCH + cc * 16) = st[i - i0];
        }
      }
    } else {
      if (!PF) load_block(blk);
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int q = tid + 256 * i, r = q / CPR, cc = q - r * CPR;
        *reinterpret_cast<V8*>(tile + r * PITCH + cc * 16) = st[i];
      }
      if (PF && blk + (int)gridDim.x < nblocks) load_block(blk + gridDim.x);
    }
    __syncthreads();

    // ---- 2. partial scores^T and row statistics over this wave's k slices ----
    f32x16 s = zero16;
    float sum = 0.0f, sq = 0.0f;
    const unsigned char* xrow = tile + l31 * PITCH;
#pragma unroll
    for (int j = 0; j < NKW; ++j) {
      const V8 xf = *reinterpret_cast<const V8*>(xrow + ((wave + 4 * j) * 16 + hi * 8) * 2);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = to_f32(xf[e]); sum += f; sq = __builtin_fmaf(f, f, sq); }
      s = Vec<T>::mfma32(sgf[j], xf, s);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s_part[wave][r][lane] = s[r];
    s_stat[wave][0][lane] = sum;
    s_stat[wave][1][lane] = sq;
    __syncthreads();
    sum = (s_stat[0][0][lane] + s_stat[1][0][lane]) + (s_stat[2][0][lane] + s_stat[3][0][lane]);
    sq = (s_stat[0][1][lane] + s_stat[1][1][lane]) + (s_stat[2][1][lane] + s_stat[3][1][lane]);
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = (s_part[0][r][lane] + s_part[1][r][lane]) + (s_part[2][r][lane] + s_part[3][r][lane]);
    sum += __shfl_xor(sum, 32, 64);
    sq += __shfl_xor(sq, 32, 64);
    const float mean = sum / (float)C;
    const float var = fmaxf(sq / (float)C - mean * mean, 0.0f);
    const float rstd = rsqrtf(var + p.eps);

    // ---- 3. softmax over the 4 tokens of each head; O^T blocks + bias + residual, back into the tile ----
    V8 pf[2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float sc[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) sc[jj] = __builtin_fmaf(rstd, s[4 * g + jj] - mean * s_gb[0][8 * g + 4 * hi + jj], s_gb[1][8 * g + 4 * hi + jj]);
      const float m = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
      float e4[4], l = 0.0f;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) { e4[jj] = __builtin_amdgcn_exp2f(sc[jj] - m); l += e4[jj]; }
      const float inv = __builtin_amdgcn_rcpf(l);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) pf[g >> 1][(g & 1) * 4 + jj] = from_f32<T>(e4[jj] * inv);
    }
    unsigned char* yrow = tile + l31 * PITCH;
    float e_s = 0.0f, e_q = 0.0f;      // stats_out: this lane's share of its row's (sum, sum of squares) of the ROUNDED outputs
#pragma unroll
    for (int j = 0; j < NCW; ++j) {
      const int cb = wave + 4 * j;
      if (cb < NCB) {
        f32x16 o = Vec<T>::mfma32(ow0[j], pf[0], zero16);
        o = Vec<T>::mfma32(ow1[j], pf[1], o);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = cb * 32 + 8 * g + 4 * hi;
          const V4 res = *reinterpret_cast<const V4*>(yrow + c * 2);
          const V4 bb = *reinterpret_cast<const V4*>(bo + c);
          V4 out;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            out[jj] = from_f32<T>(o[4 * g + jj] + to_f32(bb[jj]) + to_f32(res[jj]));
            const float f = to_f32(out[jj]);
            e_s += f; e_q = __builtin_fmaf(f, f, e_q);
          }
          *reinterpret_cast<V4*>(yrow + c * 2) = out;
        }
      }
    }
    if (p.stats_out) { s_ostat[wave][0][lane] = e_s; s_ostat[wave][1][lane] = e_q; }
    __syncthreads();
    if (p.stats_out && tid < 32 && row0 + tid < p.rows) {
      // row tid: both halves (lanes tid, tid + 32) of the four waves, in a fixed order
      float sm = 0.0f, sq2 = 0.0f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        sm += s_ostat[w][0][tid] + s_ostat[w][0][tid + 32];
        sq2 += s_ostat[w][1][tid] + s_ostat[w][1][tid + 32];
      }
      const float om = sm / (float)C;
      *reinterpret_cast<float2*>(p.stats_out + 2 * (row0 + tid)) = float2{om, rsqrtf(fmaxf(sq2 / (float)C - om * om, 0.0f) + p.stats_eps)};
    }

    // ---- 4. tile -> y ----
    {
      T* yb = reinterpret_cast<T*>(p.y);
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int q = tid + 256 * i, r = q / CPR, cc = q - r * CPR;
        const V8 v = *reinterpret_cast<const V8*>(tile + r * PITCH + cc * 16);
        if (row0 + r < p.rows) st8<T>(yb + (row0 + r) * C + cc * 8, v);
      }
    }
    __syncthreads();       // the next block's loads overwrite the tile
  }
}

}  // namespace hallo

using namespace hallo;

// hallo_set_option("xattn_tiled", 0 | 1 | 2): LDS-staged face cross-attention for C = 320 / 640 / 1280; 2 (default, round 5): next-block prefetch at
// C = 320 (65536 rows: 29.8-32.0 -> 24.5-25.6 us, 73728 rows: 31.8 -> 26.4; profiles/r5_prefetch_ab.json)
static int g_xattn_tiled = 2;
static int g_xattn_cap = 0;      // hallo_set_option("xattn_cap", n): resident-workgroup cap of the tiled kernel at C = 640 / 1280 (0 = 512; A/B)
extern "C" int hallo_set_option_xattn(const char* name, int value) {
  if (name && !strcmp(name, "xattn_tiled")) { if (value < 0 || value > 2) return -22; g_xattn_tiled = value; return 0; }
  if (name && !strcmp(name, "xattn_cap")) { if (value < 0 || value > 4096) return -22; g_xattn_cap = value; return 0; }
  return -2;
}

extern "C" int hallo_get_option_fp8(const char* name);

extern "C" int hallo_get_option_xattn(const char* name) {
  if (name && !strcmp(name, "xattn_tiled")) return g_xattn_tiled;
  if (name && !strcmp(name, "xattn_cap")) return g_xattn_cap;
  return hallo_get_option_fp8(name);       // fp8.hip
}

extern "C" int hallo_row_stats(const void* x, float* stats, int64_t rows, int C, float eps, int dtype, void* stream);   // norm_elementwise.hip

extern "C" int hallo_face_xattn_stats(const void* x, void* y, const void* sg, const float* g, const float* b, const void* owp,
                                      const void* bo, int64_t rows, int C, int64_t rows_per_batch, float eps, float* stats_out,
                                      float stats_eps, int dtype, void* stream) {
  if (!x || !y || !sg || !g || !b || !owp || !bo || rows <= 0 || C <= 0 || (C & 31)) return -22;
  if (rows_per_batch <= 0 || (rows_per_batch & 31)) return -22;
  XattnArgs a;
  a.x = x; a.y = y; a.sg = sg; a.g = g; a.b = b; a.owp = owp; a.bo = bo;
  a.rows = rows; a.C = C; a.rows_per_batch = rows_per_batch; a.eps = eps;
  a.stats_out = stats_out; a.stats_eps = stats_eps;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((unsigned)((rows + 31) / 32)), block(256);
  if (g_xattn_tiled && (C == 320 || C == 640 || C == 1280) && (dtype == DT_F16 || dtype == DT_BF16)) {
    // persistent: the workgroups a CU keeps resident walk the 32-row blocks (the per-batch constants load once per workgroup)
    const int nblocks = (int)grid.x;
    const int cap = C == 320 ? 768 : (g_xattn_cap > 0 ? g_xattn_cap : 512);   // resident workgroups: 3 per CU at C = 320 (160 registers, 39 KB), 2 at 640, 1 at 1280
    const dim3 pg((unsigned)(nblocks < cap ? nblocks : cap));
    // the prefetching form only where a workgroup walks several blocks (C = 320 at 64 x 64: 2048 blocks over <= 512 workgroups of
    // 226 registers, two per SIMD); at C = 640 / 1280 every workgroup has one block and the extra registers would spill
#define HALLO_XT(TT, CC) do { if (g_xattn_tiled == 2 && CC == 320) hipLaunchKernelGGL((face_xattn_tiled_kernel<TT, 320, true>), dim3((unsigned)(nblocks < 512 ? nblocks : 512)), block, 0, st, a, nblocks); \
                              else hipLaunchKernelGGL((face_xattn_tiled_kernel<TT, CC, false>), pg, block, 0, st, a, nblocks); } while (0)
    if (dtype == DT_F16) { if (C == 320) HALLO_XT(_Float16, 320); else if (C == 640) HALLO_XT(_Float16, 640); else HALLO_XT(_Float16, 1280); }
    else { if (C == 320) HALLO_XT(__bf16, 320); else if (C == 640) HALLO_XT(__bf16, 640); else HALLO_XT(__bf16, 1280); }
#undef HALLO_XT
    HALLO_CHECK_LAUNCH();
    return 0;
  }
  if (dtype == DT_F16) hipLaunchKernelGGL((face_xattn_kernel<_Float16>), grid, block, 0, st, a);
  else if (dtype == DT_BF16) hipLaunchKernelGGL((face_xattn_kernel<__bf16>), grid, block, 0, st, a);
  else return -22;
  HALLO_CHECK_LAUNCH();
  if (stats_out) return hallo_row_stats(y, stats_out, rows, C, stats_eps, dtype, stream);     // the untiled kernel does not emit: one pass over y
  return 0;
}

extern "C" int hallo_face_xattn(const void* x, void* y, const void* sg, const float* g, const float* b, const void* owp,
                                const void* bo, int64_t rows, int C, int64_t rows_per_batch, float eps, int dtype,
                                void* stream) {
  return hallo_face_xattn_stats(x, y, sg, g, b, owp, bo, rows, C, rows_per_batch, eps, nullptr, 0.0f, dtype, stream);
}
