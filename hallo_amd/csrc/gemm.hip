// MFMA GEMM / implicit-GEMM conv3x3 for gfx950.
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// Replaces, on the Hallo denoising path, every torch Linear / 1x1 conv / 3x3 conv call:
//   diffusers Attention.to_q/to_k/to_v/to_out (hallo/models/attention.py:22-23),
//   FeedForward/GEGLU (attention.py:601,905; motion_module.py:420),
//   Transformer3DModel.proj_in/proj_out (transformer_3d.py:199,242),
//   InflatedConv3d (resnet.py:50-66) in ResnetBlock3D (resnet.py:388,405,408),
//   Upsample3D / Downsample3D convs (resnet.py:183,250), zero_conv_* (attention.py:865,876,889),
//   AutoencoderKL convs (face_animate.py:237-240,333-335).
//
// Tile: 128x128x64 per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 2x2 MFMA
// 32x32x16 tiles, fp32 accumulators). Operands are staged global -> VGPR -> LDS with the
// next K-tile's global loads in flight under the current tile's MFMAs. LDS rows are padded
// to 144 B so the ds_read_b128 fragment reads are bank-conflict free.
//
// Activations are token-major ("NHWC"): A rows are (frame, y, x), K is the channel axis.
// In conv mode the A operand is gathered on the fly: K index = (ky*3+kx)*Cin + c, zero padding,
// optional stride 2, optional nearest-2x upsample folded into the gather (resnet.py:166-168).
#include "common.h"
#include "../../include/hallo_amd.h"

namespace hallo {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_LD = BK + 8;  // elements per LDS row (144 bytes)

struct GemmArgs {
  const void* A; const void* B; void* C;
  int M, N, K;
  long lda, ldb, ldc;
  long sA, sB, sC;          // batch strides (elements)
  const void* bias;         // [N] (or [M] if bias_per_row) or null; for GEGLU: [2N]
  int bias_per_row;
  const void* bias2;        // [M/bias2_rpg, N] or null
  int bias2_rpg;
  long bias2_ld;
  const float* rowscale;    // [M] fp32 or null
  const void* residual;     // [M,N] (ld = ldr) or null
  long ldr, sR;
  float alpha;
  int act;
  int out_f32;
  int tiles_n, tiles_m;
  // conv gather
  int H, W, Cin, OH, OW, stride, pad_t, pad_l, upsample;
};

template <typename T, bool CONV, bool GEGLU>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
  using V8 = typename Vec<T>::v8;
  __shared__ __attribute__((aligned(16))) T sA[BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) T sB[BN * LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
  const int m0 = tile_m * BM;
  // GEGLU: a block produces 64 output columns from 64 "value" rows and 64 "gate" rows of W.
  const int n0 = GEGLU ? tile_n * 64 : tile_n * BN;
  const long zb = blockIdx.z;

  const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + zb * p.sA;
  const T* __restrict__ B = reinterpret_cast<const T*>(p.B) + zb * p.sB;

  // ---- loader mapping: thread owns k-chunk kc (8 elements) of rows r0 + 32*i ----
  const int kc = tid & 7;
  const int r0 = tid >> 3;

  // B (weights) row indices for this thread's 4 rows
  long b_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = r0 + 32 * i;  // row within the 128-row B tile
    int wrow;
    if (GEGLU) {
      // tile rows: [wn(2)][t(2: value,gate)][32]
      const int jw = j >> 6, t = (j >> 5) & 1, c = j & 31;
      int col = n0 + jw * 32 + c;
      col = col < p.N ? col : p.N - 1;
      wrow = t * p.N + col;
    } else {
      wrow = n0 + j;
      wrow = wrow < p.N ? wrow : p.N - 1;
    }
    b_off[i] = (long)wrow * p.ldb;
  }

  // A row info
  long a_off[4];
  int a_iy[4], a_ix[4];
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + r0 + 32 * i;
    a_ok[i] = true;
    if (m >= p.M) { m = p.M - 1; }
    if (CONV) {
      const int hw = p.OH * p.OW;
      const int img = m / hw, rem = m - img * hw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_iy[i] = oy * p.stride - p.pad_t;
      a_ix[i] = ox * p.stride - p.pad_l;
      a_off[i] = (long)img * p.H * p.W * p.Cin;
    } else {
      a_off[i] = (long)m * p.lda;
      a_iy[i] = a_ix[i] = 0;
    }
  }
  // conv k-state: tap index and channel offset of this thread's chunk in the current K-tile
  int c_tap = 0, c_ch = 0;
  if (CONV) {
    c_tap = (kc * 8) / p.Cin;
    c_ch = (kc * 8) - c_tap * p.Cin;
  }
  const int VH = CONV ? (p.upsample ? 2 * p.H : p.H) : 0;
  const int VW = CONV ? (p.upsample ? 2 * p.W : p.W) : 0;

  V8 ra[4], rb[4];
  auto load_tile = [&](int kt) {
    const int kk = kt * BK + kc * 8;
    const bool kin = kk < p.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = kin ? ld8<T>(B + b_off[i] + kk) : zero8<T>();
    if (CONV) {
      const int ky = c_tap / 3, kx = c_tap - ky * 3;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int vy = a_iy[i] + ky, vx = a_ix[i] + kx;
        const bool inb = kin && vy >= 0 && vy < VH && vx >= 0 && vx < VW;
        const int sy = p.upsample ? (vy >> 1) : vy, sx = p.upsample ? (vx >> 1) : vx;
        ra[i] = inb ? ld8<T>(A + a_off[i] + ((long)sy * p.W + sx) * p.Cin + c_ch) : zero8<T>();
      }
      c_ch += BK;
      while (c_ch >= p.Cin) { c_ch -= p.Cin; ++c_tap; }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) ra[i] = kin ? ld8<T>(A + a_off[i] + kk) : zero8<T>();
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      st8<T>(&sA[(r0 + 32 * i) * LDS_LD + kc * 8], ra[i]);
      st8<T>(&sB[(r0 + 32 * i) * LDS_LD + kc * 8], rb[i]);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nk = (p.K + BK - 1) / BK;
  load_tile(0);
  store_tile();
  __syncthreads();

  const T* fa = &sA[(wm * 64 + l31) * LDS_LD + hi * 8];
  const T* fb = &sB[(wn * 64 + l31) * LDS_LD + hi * 8];

  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      V8 a0 = ld8<T>(fa + ks * 16);
      V8 a1 = ld8<T>(fa + 32 * LDS_LD + ks * 16);
      V8 b0 = ld8<T>(fb + ks * 16);
      V8 b1 = ld8<T>(fb + 32 * LDS_LD + ks * 16);
      acc[0][0] = Vec<T>::mfma32(a0, b0, acc[0][0]);
      acc[0][1] = Vec<T>::mfma32(a0, b1, acc[0][1]);
      acc[1][0] = Vec<T>::mfma32(a1, b0, acc[1][0]);
      acc[1][1] = Vec<T>::mfma32(a1, b1, acc[1][1]);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      store_tile();
      __syncthreads();
    }
  }

  // ---- epilogue ----
  const T* bias = reinterpret_cast<const T*>(p.bias);
  const T* bias2 = reinterpret_cast<const T*>(p.bias2);
  const T* res = p.residual ? reinterpret_cast<const T*>(p.residual) + zb * p.sR : nullptr;
  T* C = reinterpret_cast<T*>(p.C) + zb * p.sC;
  float* Cf = reinterpret_cast<float*>(p.C) + zb * p.sC;

  if (GEGLU) {
    const int n = n0 + wn * 32 + l31;
    if (n < p.N) {
      const float bh = bias ? to_f32(bias[n]) : 0.0f;
      const float bg = bias ? to_f32(bias[p.N + n]) : 0.0f;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (m < p.M) {
            const float h = acc[tm][0][r] + bh;
            const float g = acc[tm][1][r] + bg;
            C[(long)m * p.ldc + n] = from_f32<T>(h * gelu_erf_f(g));
          }
        }
      }
    }
    return;
  }

#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int n = n0 + wn * 64 + tn * 32 + l31;
    if (n >= p.N) continue;
    const float bcol = (bias && !p.bias_per_row) ? to_f32(bias[n]) : 0.0f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (m >= p.M) continue;
        float v = acc[tm][tn][r] + bcol;
        if (bias && p.bias_per_row) v += to_f32(bias[m]);
        if (bias2) v += to_f32(bias2[(long)(m / p.bias2_rpg) * p.bias2_ld + n]);
        if (p.rowscale) v *= p.rowscale[m];
        v *= p.alpha;
        if (res) v += to_f32(res[(long)m * p.ldr + n]);
        if (p.act == ACT_SILU) v = silu_f(v);
        else if (p.act == ACT_RELU) v = fmaxf(v, 0.0f);
        if (p.out_f32) Cf[(long)m * p.ldc + n] = v;
        else C[(long)m * p.ldc + n] = from_f32<T>(v);
      }
    }
  }
}

template <typename T>
static int launch_gemm(const GemmArgs& a, bool conv, bool geglu, int batch, hipStream_t st) {
  dim3 grid(a.tiles_m * a.tiles_n, 1, batch), block(256);
  if (geglu) hipLaunchKernelGGL((gemm_kernel<T, false, true>), grid, block, 0, st, a);
  else if (conv) hipLaunchKernelGGL((gemm_kernel<T, true, false>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, block, 0, st, a);
  HALLO_CHECK_LAUNCH();
  return 0;
}

}  // namespace hallo

using namespace hallo;

extern "C" int hallo_gemm(const hallo_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->B || !d->C) return -22;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || (d->K & 7)) return -22;
  if (d->batch < 1) return -22;
  if ((d->lda & 7) || (d->ldb & 7)) return -22;
  if (d->geglu && (d->rowscale || d->residual || d->bias2 || d->out_f32 || d->bias_per_row)) return -22;
  GemmArgs a;
  a.A = d->A; a.B = d->B; a.C = d->C;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
  a.sA = d->stride_a; a.sB = d->stride_b; a.sC = d->stride_c;
  a.bias = d->bias; a.bias_per_row = d->bias_per_row;
  a.bias2 = d->bias2; a.bias2_rpg = d->bias2_rows_per_group > 0 ? d->bias2_rows_per_group : 1;
  a.bias2_ld = d->bias2_ld > 0 ? d->bias2_ld : d->N;
  a.rowscale = d->rowscale;
  a.residual = d->residual; a.ldr = d->ldr; a.sR = d->stride_r;
  a.alpha = d->alpha; a.act = d->act; a.out_f32 = d->out_f32;
  a.tiles_m = (d->M + BM - 1) / BM;
  a.tiles_n = d->geglu ? (d->N + 63) / 64 : (d->N + BN - 1) / BN;
  a.H = a.W = a.Cin = a.OH = a.OW = a.stride = a.pad_t = a.pad_l = a.upsample = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == DT_F16) return launch_gemm<_Float16>(a, false, d->geglu != 0, d->batch, st);
  if (d->dtype == DT_BF16) return launch_gemm<__bf16>(a, false, d->geglu != 0, d->batch, st);
  return -22;
}

extern "C" int hallo_conv3x3_nhwc(const hallo_conv_desc* d, void* stream) {
  if (!d || !d->x || !d->w || !d->y) return -22;
  if (d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || (d->Cin & 7) || d->Cout <= 0) return -22;
  if (d->stride != 1 && d->stride != 2) return -22;
  if (d->OH <= 0 || d->OW <= 0) return -22;
  GemmArgs a;
  a.A = d->x; a.B = d->w; a.C = d->y;
  a.M = d->n_img * d->OH * d->OW; a.N = d->Cout; a.K = 9 * d->Cin;
  a.lda = d->Cin; a.ldb = a.K; a.ldc = d->ldy > 0 ? d->ldy : d->Cout;
  a.sA = a.sB = a.sC = 0;
  a.bias = d->bias; a.bias_per_row = 0;
  a.bias2 = d->bias2; a.bias2_rpg = d->bias2_rows_per_group > 0 ? d->bias2_rows_per_group : 1;
  a.bias2_ld = d->bias2_ld > 0 ? d->bias2_ld : d->Cout;
  a.rowscale = nullptr;
  a.residual = d->residual; a.ldr = d->ldr > 0 ? d->ldr : d->Cout; a.sR = 0;
  a.alpha = d->alpha; a.act = d->act; a.out_f32 = 0;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.OH = d->OH; a.OW = d->OW;
  a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.upsample = d->upsample;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == DT_F16) return launch_gemm<_Float16>(a, true, false, 1, st);
  if (d->dtype == DT_BF16) return launch_gemm<__bf16>(a, true, false, 1, st);
  return -22;
}
