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
#include "gemm_args.h"
#include "../../include/hallo_amd.h"
#include <string.h>

namespace hallo {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_LD = BK + 8;  // elements per LDS row (144 bytes)


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
        v *= (n < p.lead_cols) ? p.alpha * p.lead_alpha : p.alpha;
        if (p.act == ACT_GELU_PRE) v = gelu_erf_f(v);
        if (res) v += to_f32(res[(long)m * p.ldr + n]);
        if (p.act == ACT_SILU) v = silu_f(v);
        else if (p.act == ACT_RELU) v = fmaxf(v, 0.0f);
        else if (p.act == ACT_GELU) v = gelu_erf_f(v);
        if (p.out_f32) Cf[(long)m * p.ldc + n] = v;
        else C[(long)m * p.ldc + n] = from_f32<T>(v);
      }
    }
  }
}


// =============================================================================================
// v2: direct-to-LDS staging (global_load_lds, 16 B per lane), swizzled through the SOURCE address,
// swapped-operand MFMA so that every lane owns one output row.
//
//  * Each operand tile is [128 rows][64 k] (128 B per row).  One global_load_lds_dwordx4 of a wave writes
//    1024 contiguous LDS bytes = 8 rows; lane l lands at row (l>>3), physical 16-B chunk (l&7), and fetches
//    the LOGICAL chunk (l&7) ^ ((row>>1)&7) of that row from global memory (same 128-B line: still coalesced).
//    Fragment reads apply the same XOR.  With 128-B rows two rows share one 256-B bank row, and
//    (row&1, (row>>1)&7) is distinct over every 16-lane group ds_read_b128 is serviced in
//    ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...), so the fragment reads are bank-conflict free.
//  * No VGPR staging, no ds_write pass: the K loop is {issue 8 LDS-DMA loads, wait, barrier, 16 ds_read_b128
//    + 16 MFMA per wave, barrier}.  STAGES = 2 keeps the next tile's loads in flight under the current
//    tile's MFMAs with a counted s_waitcnt vmcnt(8) (raw s_barrier: __syncthreads would drain the DMA queue).
//  * Out-of-range K chunks and the zero padding of the 3x3 conv gather read a 16-byte zero page.
//  * MFMA operands are swapped (A = weight rows, B = activation rows): D[n][m], lane owns column m and four
//    consecutive n per register group -> 8-byte C stores / residual loads instead of 2-byte scattered ones.
// =============================================================================================
__device__ uint4 g_zero_page[2];

constexpr int TILE_ELEMS = 128 * 64;

// (sum, sum of squares) of 8 storage-type values added to (s, q): packed dot products with fp32 accumulation (v_dot2c_f32_*), no
// conversions -- the same four steps in the same order wherever row statistics of ROUNDED outputs are taken (EMIT epilogue of
// gemm2_kernel, row_parts_kernel), so that both give the same bits.
template <typename T>
__device__ __forceinline__ void row_sums8(typename Vec<T>::v8 w, float& s, float& q) {
  typedef __attribute__((ext_vector_type(2))) T V2t;
  const V2t one2 = {from_f32<T>(1.0f), from_f32<T>(1.0f)};
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const V2t x2 = {w[e], w[e + 1]};
    s = dot2(x2, one2, s);
    q = dot2(x2, x2, q);
  }
}

// LNF: LayerNorm fused into the GEMM (see GemmArgs::ln_colsum).  The A fragments that feed the MFMAs are also
// reduced to per-row sum / sum of squares with packed dot2 instructions (16 VALU ops per 16-deep slice and lane, in the
// shadow of 4 MFMAs), so nn.LayerNorm costs no pass over HBM at all; the epilogue applies
// rstd * (acc - mean * colsum[n]) + bias.  A lane owns output row l31 of each 32-row block in the swapped MFMA
// form, which is the row its A fragments belong to: the statistics are lane-local up to one lane^32 exchange.
// LNF = 1 computes the row statistics in the K loop as described; LNF = 2 reads them from GemmArgs::ln_stats
// (hallo_row_stats: one read pass over A) -- cheaper whenever several N tiles share a row block, because every tile
// would otherwise redo the statistics of the same rows (measured: the in-loop form makes the N = 2560 GEGLU GEMMs
// 25 % slower, more than the LayerNorm launch it replaces).
// EMIT (round 5, MODE 0 / LNF 0 only): the epilogue also writes GemmArgs::row_parts -- (sum, sum of squares) of the ROUNDED
// output values of every row over this wave's 64 columns, slot (n0 + 64 wn) / 64 of [M][ceil(N / 64)][2] -- so that an
// nn.LayerNorm over C's rows (the next projection's LNF = 2 epilogue) needs no pass over C (hallo_row_stats): 8 lanes hold a
// row's 64 columns in the read phase, sum8_dpp reduces them.  A separate instantiation: the plain kernel sits at
// the 128-VGPR boundary.
// LNF = 2 with GemmArgs::ln_parts > 0: ln_stats holds such partial sums over the K columns of A; behind the K loop the tile's rows
// are staged through the idle LDS and reduced to (mean, rstd) per row in slot order (deterministic).
template <typename T, int MODE /*0 gemm, 1 conv3x3, 2 geglu, 3 gemm with a GELU epilogue*/, int STAGES, int LNF, int EMIT = 0>
__global__ __launch_bounds__(256, STAGES == 2 ? 2 : (MODE == 1 || MODE == 3 || EMIT == 2 ? 3 : 4)) void gemm2_kernel(const GemmArgs p) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  static_assert(!EMIT || (MODE == 0 && LNF == 0), "row_parts emission: plain GEMM epilogue only");
  // MODE 3 is a separate instantiation so that the erf polynomial's registers never burden the hot MODE 0 kernel
  // (it sits exactly at the 128-VGPR / 4-waves-per-SIMD boundary)
  constexpr bool CONV = MODE == 1, GEGLU = MODE == 2, GELU = MODE == 3;
  __shared__ __attribute__((aligned(16))) T smem[STAGES * 2 * TILE_ELEMS];
  __shared__ float s_ln[LNF == 2 ? 2 * BM : 2];     // (mean, rstd) of this tile's rows when ln_stats holds partial sums

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  const int nwg = p.tiles_m * p.tiles_n;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = GEGLU ? tile_n * 64 : tile_n * BN;
  const long zb = blockIdx.z;

  const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + zb * p.sA;
  const T* __restrict__ B = reinterpret_cast<const T*>(p.B) + zb * p.sB;

  // ---- loader: LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... offen lds) ----
  // Thread owns physical chunk lp of rows r_j = wave*32 + j*8 + lrow (j = 0..3) of both operand tiles.  The per-lane
  // byte offset of (row, swizzled chunk) is computed ONCE; the K advance is the instruction's SCALAR offset, so the
  // steady-state K loop issues its 8 loads with no vector ALU work at all.  Rows past M / N, the K tail and the
  // zero padding of the 3x3 gather use voffset = 0xFFFFFFFF: out of the descriptor's range -> the DMA writes zeros.
  constexpr unsigned OOB = 0xFFFFFFFFu;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int lrow = lane >> 3, lp = lane & 7;
  const int cA = lp ^ (lane >> 4);   // logical chunk of slabs j = 0, 2   ((r_j >> 1) & 7 == (lane >> 4) + 4*(j&1))
  const int cB = cA ^ 4;             // logical chunk of slabs j = 1, 3
  auto clamp32 = [](long bytes) { return (int)(bytes > 0xFFFFFFFFL ? 0xFFFFFFFFL : bytes); };

  const T* Bbase = GEGLU ? B : B + (long)n0 * p.ldb;
  const long b_rows = GEGLU ? 2L * p.N : (long)(p.N - n0);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(Bbase), 0, clamp32(((b_rows - 1) * p.ldb + p.K) * 2), 0x00020000);
  unsigned b_voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 32 + j * 8 + lrow;
    const int c = (j & 1) ? cB : cA;
    long wrow;
    bool ok;
    if (GEGLU) {
      const int jw = r >> 6, t = (r >> 5) & 1, cc = r & 31;   // tile rows: [wn(2)][value, gate][32]
      const int col = n0 + jw * 32 + cc;
      ok = col < p.N;
      wrow = (long)t * p.N + col;
    } else {
      ok = n0 + r < p.N;
      wrow = r;
    }
    b_voff[j] = ok ? (unsigned)((wrow * p.ldb + c * 8) * 2) : OOB;
  }

  const int nk_all = (p.K + BK - 1) / BK;
  const int kt_begin = (p.splits > 1) ? blockIdx.y * p.nk_per_split : 0;
  const int kt_end = (p.splits > 1) ? min(nk_all, kt_begin + p.nk_per_split) : nk_all;

  unsigned a_voff[4];
  unsigned a_mask[4] = {0, 0, 0, 0};     // conv fast path: bit t set <=> tap t of this row is inside the image
  int a_iy[4] = {0, 0, 0, 0}, a_ix[4] = {0, 0, 0, 0};
  const T* Abase;
  long a_bytes;
  const int VH = CONV ? (p.upsample ? 2 * p.H : p.H) : 0;
  const int VW = CONV ? (p.upsample ? 2 * p.W : p.W) : 0;
  const bool cfast = CONV && p.conv_fast;
  if (CONV) {
    const int hw = p.OH * p.OW;
    const int img0 = m0 / hw, n_img = p.M / hw;
    const long img_elems = (long)p.H * p.W * p.Cin;
    const long shift = cfast ? ((long)p.pad_t * p.W + p.pad_l) * p.Cin : 0;   // taps are addressed from (-pad_t, -pad_l)
    Abase = A + img0 * img_elems - shift;
    a_bytes = ((long)(n_img - img0) * img_elems + shift) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wave * 32 + j * 8 + lrow;
      const int c = (j & 1) ? cB : cA;
      const int img = m / hw, rem = m - img * hw;
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      a_iy[j] = oy * p.stride - p.pad_t;
      a_ix[j] = ox * p.stride - p.pad_l;
      const unsigned img_off = (unsigned)((img - img0) * img_elems * 2);
      if (cfast) {
        a_voff[j] = (m < p.M) ? img_off + (unsigned)((((long)oy * p.stride * p.W + ox * p.stride) * p.Cin + c * 8) * 2) : OOB;
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int vy = a_iy[j] + t / 3, vx = a_ix[j] + t % 3;
          if (m < p.M && vy >= 0 && vy < p.H && vx >= 0 && vx < p.W) mk |= 1u << t;
        }
        a_mask[j] = mk;
      } else {
        a_voff[j] = (m < p.M) ? img_off : OOB;     // image base only; the pixel / channel part is per K tile
      }
    }
  } else {
    Abase = A + (long)m0 * p.lda;
    const int rows = min(p.M - m0, BM);
    a_bytes = ((long)(rows - 1) * p.lda + p.K) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = wave * 32 + j * 8 + lrow;
      const int c = (j & 1) ? cB : cA;
      a_voff[j] = (r < rows) ? (unsigned)(((long)r * p.lda + c * 8) * 2) : OOB;
    }
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Abase), 0, clamp32(a_bytes), 0x00020000);

  // conv K position.  Fast path (Cin % 64 == 0): one tap per K tile, tracked in scalars.  General path: the two
  // logical chunks of a thread may sit in different taps; tracked per thread.
  int tap_u = 0, ch_u = 0, tapA = 0, chA = 0, tapB = 0, chB = 0;
  if (CONV) {
    const int k0 = kt_begin * BK;
    tap_u = k0 / p.Cin; ch_u = k0 - tap_u * p.Cin;
    tapA = (k0 + cA * 8) / p.Cin; chA = k0 + cA * 8 - tapA * p.Cin;
    tapB = (k0 + cB * 8) / p.Cin; chB = k0 + cB * 8 - tapB * p.Cin;
  }

#define HALLO_BLOAD(rs, ldsptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ldsptr), 16, (int)(voff), (int)(soff), 0, 0)

  auto stage = [&](int kt, int buf) {
    T* sA = smem + buf * 2 * TILE_ELEMS + wave_u * 32 * BK;
    T* sB = sA + TILE_ELEMS;
    const bool tail = (kt + 1) * BK > p.K;        // wave-uniform; only the last tile of a K % 64 != 0 problem
    const int soff = kt * BK * 2;
    if (!tail) {
#pragma unroll
      for (int j = 0; j < 4; ++j) HALLO_BLOAD(rsB, sB + j * 8 * BK, b_voff[j], soff);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = kt * BK + ((j & 1) ? cB : cA) * 8;
        HALLO_BLOAD(rsB, sB + j * 8 * BK, (kk < p.K && b_voff[j] != OOB) ? b_voff[j] + soff : OOB, 0);
      }
    }
    if (!CONV) {
      if (!tail) {
#pragma unroll
        for (int j = 0; j < 4; ++j) HALLO_BLOAD(rsA, sA + j * 8 * BK, a_voff[j], soff);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kk = kt * BK + ((j & 1) ? cB : cA) * 8;
          HALLO_BLOAD(rsA, sA + j * 8 * BK, (kk < p.K && a_voff[j] != OOB) ? a_voff[j] + soff : OOB, 0);
        }
      }
    } else if (cfast) {
      const int ky = tap_u / 3, kx = tap_u - ky * 3;
      const int soffA = ((ky * p.W + kx) * p.Cin + ch_u) * 2;
      const unsigned bit = 1u << tap_u;
#pragma unroll
      for (int j = 0; j < 4; ++j) HALLO_BLOAD(rsA, sA + j * 8 * BK, (a_mask[j] & bit) ? a_voff[j] : OOB, soffA);
      ch_u += BK;
      if (ch_u >= p.Cin) { ch_u = 0; ++tap_u; }
    } else {
      const int kA = kt * BK + cA * 8, kB = kt * BK + cB * 8;
      const int kyA = tapA / 3, kxA = tapA - kyA * 3, kyB = tapB / 3, kxB = tapB - kyB * 3;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = (j & 1) ? kB : kA;
        const int ky = (j & 1) ? kyB : kyA, kx = (j & 1) ? kxB : kxA, ch = (j & 1) ? chB : chA;
        const int vy = a_iy[j] + ky, vx = a_ix[j] + kx;
        const bool inb = kk < p.K && a_voff[j] != OOB && vy >= 0 && vy < VH && vx >= 0 && vx < VW;
        const int sy = p.upsample ? (vy >> 1) : vy, sx = p.upsample ? (vx >> 1) : vx;
        HALLO_BLOAD(rsA, sA + j * 8 * BK, inb ? a_voff[j] + (unsigned)(((sy * p.W + sx) * p.Cin + ch) * 2) : OOB, 0);
      }
      chA += BK; while (chA >= p.Cin) { chA -= p.Cin; ++tapA; }
      chB += BK; while (chB >= p.Cin) { chB -= p.Cin; ++tapB; }
    }
  };
#undef HALLO_BLOAD

  f32x16 acc[2][2];   // [tn][tm]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // fragment read offsets: row = base + l31, logical chunk ks*2 + hi -> physical (ks*2) ^ (hi ^ ((l31>>1)&7))
  const int xsw = hi ^ ((l31 >> 1) & 7);
  const int fa_row = (wm * 64 + l31) * BK;
  const int fb_row = (wn * 64 + l31) * BK;

  float ln_s[2] = {0.0f, 0.0f}, ln_q[2] = {0.0f, 0.0f};      // LNF: partial sum / sum of squares of rows l31, l31 + 32
  auto row_stats = [&](V8 a, float& sm, float& sq) {
    typedef __attribute__((ext_vector_type(2))) T V2t;
    const V2t one2 = {from_f32<T>(1.0f), from_f32<T>(1.0f)};
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const V2t x2 = {a[e], a[e + 1]};
      sm = dot2(x2, one2, sm);
      sq = dot2(x2, x2, sq);
    }
  };
  auto compute = [&](int buf) {
    const T* sA = smem + buf * 2 * TILE_ELEMS;
    const T* sB = sA + TILE_ELEMS;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int co = ((ks * 2) ^ xsw) * 8;
      V8 a0 = ld8<T>(sA + fa_row + co);
      V8 a1 = ld8<T>(sA + fa_row + 32 * BK + co);
      V8 w0 = ld8<T>(sB + fb_row + co);
      V8 w1 = ld8<T>(sB + fb_row + 32 * BK + co);
      if (LNF == 1) { row_stats(a0, ln_s[0], ln_q[0]); row_stats(a1, ln_s[1], ln_q[1]); }
      acc[0][0] = Vec<T>::mfma32(w0, a0, acc[0][0]);
      acc[0][1] = Vec<T>::mfma32(w0, a1, acc[0][1]);
      acc[1][0] = Vec<T>::mfma32(w1, a0, acc[1][0]);
      acc[1][1] = Vec<T>::mfma32(w1, a1, acc[1][1]);
    }
  };

  if (STAGES == 1) {
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      stage(kt, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      compute(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
    if (kt_begin < kt_end) stage(kt_begin, kt_begin & 1);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      if (kt + 1 < kt_end) {
        stage(kt + 1, (kt + 1) & 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile kt landed, tile kt+1 (8 DMA loads) in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      compute(kt & 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // fragment reads done before the buffer is released
      __builtin_amdgcn_s_barrier();                          // buffer kt&1 is free for tile kt+2
    }
  }

  if (LNF == 2 && p.ln_parts > 0) {
    // ln_stats holds the producer's partial (sum, sum of squares) per 64-column block of A's rows: [M][P][2].  The K loop ended on
    // a barrier, the staging LDS is idle: this tile's rows arrive there by coalesced 16-byte loads (all in flight at once -- one L2
    // round trip; a per-row loop of dependent loads cost 8-13 us per launch), then thread r reduces row r in slot order.
    const int P2 = p.ln_parts * 2;
    const int rows = min(BM, p.M - m0);
    const int nflt = rows * P2, nvec = nflt >> 2;           // whole 16-byte pieces, then the (<= 3 floats) tail: nothing is read past the rows' partials
    const f32x4* src = reinterpret_cast<const f32x4*>(p.ln_stats + (long)m0 * P2);
    f32x4* stg = reinterpret_cast<f32x4*>(smem);
    for (int i = tid; i < nvec; i += 256) stg[i] = src[i];
    if (tid < (nflt & 3)) reinterpret_cast<float*>(smem)[nvec * 4 + tid] = p.ln_stats[(long)m0 * P2 + nvec * 4 + tid];
    __syncthreads();
    if (tid < BM) {
      const float* pr = reinterpret_cast<const float*>(smem) + min(tid, rows - 1) * P2;
      float sm = 0.0f, sq = 0.0f;
      for (int i = 0; i < P2; i += 2) { sm += pr[i]; sq += pr[i + 1]; }
      const float mean = sm / (float)p.K;
      s_ln[2 * tid] = mean;
      s_ln[2 * tid + 1] = rsqrtf(fmaxf(sq / (float)p.K - mean * mean, 0.0f) + p.ln_eps);
    }
    __syncthreads();                                       // the epilogue's transposition scratch reuses the same LDS
  }

  // ---- epilogue ----
  // The accumulators hold D[n][m] (lane: column m = l31, rows n = 8g + 4hi + 0..3): stored directly, every lane
  // would write 8 bytes at a row stride -- 32 partial lines per store instruction.  Instead each wave transposes
  // its 32 x 64 fp32 half-tile through its own 8 KB slice of the (now idle) staging LDS, XOR-swizzled in 16-byte
  // chunks (chunk ^ (row & 15): conflict-free ds_write_b128 and ds_read_b128), and then every lane owns 8
  // consecutive n of one row: bias / residual loads and the C store are 16 bytes per lane, 128 contiguous
  // bytes per row, 8 rows per instruction.
  const T* bias = reinterpret_cast<const T*>(p.bias);
  const T* bias2 = reinterpret_cast<const T*>(p.bias2);
  const T* res = p.residual ? reinterpret_cast<const T*>(p.residual) + zb * p.sR : nullptr;
  T* C = reinterpret_cast<T*>(p.C) + zb * p.sC;
  float* Cf = reinterpret_cast<float*>(p.C) + zb * p.sC;
  float* tile = reinterpret_cast<float*>(smem) + wave * (32 * 64);   // [32 rows][16 chunks of 4 fp32]

  auto ld8f = [&](const T* ptr, bool aligned, float* o) {   // 8 consecutive elements -> fp32
    if (aligned) {
      V8 v = ld8<T>(ptr);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = to_f32(v[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = to_f32(ptr[j]);
    }
  };

  const int rr0 = lane >> 3;            // read phase: rows rr0 + 8 i
  const int rc = lane & 7;              // read phase: 8-column (GEGLU: 4-column) group
  // per-column bias of this lane's columns, loaded once
  float bcol[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (bias && !p.bias_per_row) {
    if (GEGLU) {
      const int n = n0 + wn * 32 + rc * 4;
      if (n < p.N) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { bcol[j] = to_f32(bias[n + j]); bcol[4 + j] = to_f32(bias[p.N + n + j]); }
      }
    } else {
      const int n = n0 + wn * 64 + rc * 8;
      if (n < p.N) ld8f(bias + n, p.bias_vec_ok, bcol);
    }
  }
  // (the LayerNorm-fused projections have no residual on the path: no prefetch registers for it in that variant)
  const bool use_res = !GEGLU && !LNF && res && p.res_vec_ok && p.splits <= 1;
  // LNF: row statistics of rows l31 (tm = 0) and l31 + 32 (tm = 1) of this wave's 64-row block, and the fp32 column
  // sums of this lane's output columns
  float ln_mean[2] = {0.0f, 0.0f}, ln_rstd[2] = {1.0f, 1.0f};
  if (LNF == 1) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float sm = ln_s[t] + __shfl_xor(ln_s[t], 32, 64);      // the partner lane holds the other k chunks of the row
      const float sq = ln_q[t] + __shfl_xor(ln_q[t], 32, 64);
      const float mean = sm / (float)p.K;
      ln_mean[t] = mean;
      ln_rstd[t] = rsqrtf(fmaxf(sq / (float)p.K - mean * mean, 0.0f) + p.ln_eps);
    }
  }
  // column sums of this lane's output columns: loaded where they are used (L2-resident; keeping 8 more values live
  // through the epilogue pushes the kernel over 128 VGPRs = 4 workgroups per CU)
  auto load_gcol = [&](float* gc) {
    if (GEGLU) {
      const int n = n0 + wn * 32 + rc * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) gc[j] = 0.0f;
      if (n < p.N) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.ln_colsum + n), b = *reinterpret_cast<const f32x4*>(p.ln_colsum + p.N + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) { gc[j] = a[j]; gc[4 + j] = b[j]; }
      }
    } else {
      const int n = n0 + wn * 64 + rc * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) gc[j] = 0.0f;
      if (n < p.N) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.ln_colsum + n), b = *reinterpret_cast<const f32x4*>(p.ln_colsum + n + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { gc[j] = a[j]; gc[4 + j] = b[j]; }
      }
    }
  };
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    // residual rows of this half (4 row groups x 8 columns per lane): issued before the LDS transposition so
    // that their latency overlaps it instead of serialising the read phase
    V8 rpre[4];
    if (use_res) {
      const int n = n0 + wn * 64 + rc * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + tm * 32 + rr0 + 8 * i;
        rpre[i] = (m < p.M && n < p.N) ? ld8<T>(res + (long)m * p.ldr + n) : zero8<T>();
      }
    }
    // ---- write phase: acc[tn][tm] -> tile[row = l31][col = tn*32 + 8g + 4hi + 0..3] ----
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int chunk = (tn * 8 + 2 * g + hi) ^ (l31 & 15);
        *reinterpret_cast<f32x4*>(tile + l31 * 64 + chunk * 4) =
            f32x4{acc[tn][tm][g * 4], acc[tn][tm][g * 4 + 1], acc[tn][tm][g * 4 + 2], acc[tn][tm][g * 4 + 3]};
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- read phase ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = rr0 + 8 * i;
      const int m = m0 + wm * 64 + tm * 32 + rr;
      // LNF: statistics of row rr live in lane rr (either half)
      float r_mean = 0.0f, r_rstd = 1.0f;
      if (LNF == 1) {
        r_mean = __shfl(ln_mean[tm], rr, 64);
        r_rstd = __shfl(ln_rstd[tm], rr, 64);
      } else if (LNF == 2) {
        if (p.ln_parts > 0) {
          r_mean = s_ln[2 * (wm * 64 + tm * 32 + rr)];
          r_rstd = s_ln[2 * (wm * 64 + tm * 32 + rr) + 1];
        } else {
          const float2 st2 = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (long)min(m, p.M - 1));
          r_mean = st2.x;
          r_rstd = st2.y;
        }
      }
      float gcol[8];
      if (LNF) load_gcol(gcol);
      float e_s = 0.0f, e_q = 0.0f;      // EMIT: this lane's share of the row's (sum, sum of squares)
      if (GEGLU) {
        const int n = n0 + wn * 32 + rc * 4;
        f32x4 hv = *reinterpret_cast<const f32x4*>(tile + rr * 64 + ((rc ^ (rr & 15)) * 4));
        f32x4 gv = *reinterpret_cast<const f32x4*>(tile + rr * 64 + (((8 + rc) ^ (rr & 15)) * 4));
        if (LNF) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            hv[j] = r_rstd * (hv[j] - r_mean * gcol[j]);
            gv[j] = r_rstd * (gv[j] - r_mean * gcol[4 + j]);
          }
        }
        if (m < p.M && n < p.N) {
          V4 w;
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = from_f32<T>((hv[j] + bcol[j]) * gelu_erf_f(gv[j] + bcol[4 + j]));
          *reinterpret_cast<V4*>(C + (long)m * p.ldc + n) = w;
        }
      } else {
        const int n = n0 + wn * 64 + rc * 8;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(tile + rr * 64 + (((2 * rc) ^ (rr & 15)) * 4));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(tile + rr * 64 + (((2 * rc + 1) ^ (rr & 15)) * 4));
        if (p.splits > 1) {
          if (m < p.M && n < p.N) {
            float* sp = p.slab + ((long)blockIdx.y * p.M + m) * p.N + n;
            if (p.slab_nt) {
              __builtin_nontemporal_store(v0, reinterpret_cast<f32x4*>(sp));
              __builtin_nontemporal_store(v1, reinterpret_cast<f32x4*>(sp + 4));
            } else {
              *reinterpret_cast<f32x4*>(sp) = v0;
              *reinterpret_cast<f32x4*>(sp + 4) = v1;
            }
          }
        } else if (m < p.M && n < p.N) {
          float o[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          float t8[8];
          if (LNF) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = r_rstd * (o[j] - r_mean * gcol[j]);
          }
          const float br = (bias && p.bias_per_row) ? to_f32(bias[m]) : 0.0f;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += bcol[j] + br;
          if (bias2) {
            ld8f(bias2 + (long)(m / p.bias2_rpg) * p.bias2_ld + n, p.bias2_vec_ok, t8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += t8[j];
          }
          const float rs = (p.rowscale ? p.rowscale[m] * p.alpha : p.alpha) * ((n < p.lead_cols) ? p.lead_alpha : 1.0f);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] *= rs;
          if (GELU && p.act == ACT_GELU_PRE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = gelu_erf_f(o[j]);
          }
          if (use_res) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += to_f32(rpre[i][j]);
          } else if (res) {
            ld8f(res + (long)m * p.ldr + n, false, t8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += t8[j];
          }
          if (p.act == ACT_SILU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = silu_f(o[j]);
          } else if (p.act == ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], 0.0f);
          } else if (GELU && p.act == ACT_GELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = gelu_erf_f(o[j]);
          }
          if (p.out_f32) {
            float* cp = Cf + (long)m * p.ldc + n;
            *reinterpret_cast<f32x4*>(cp) = f32x4{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4*>(cp + 4) = f32x4{o[4], o[5], o[6], o[7]};
          } else {
            V8 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = from_f32<T>(o[j]);
            st8<T>(C + (long)m * p.ldc + n, w);
            if (EMIT) row_sums8<T>(w, e_s, e_q);
          }
        }
        if (EMIT) {
          // the 8 lanes rc = 0..7 of a row group hold this row's 64 columns of the wave (lanes past N hold zeros): sum them in a
          // fixed tree, lane rc = 0 writes the slot.  Wave-uniform control flow: every lane takes part in the exchanges.
          e_s = sum8_dpp(e_s);
          e_q = sum8_dpp(e_q);
          const int slot = (n0 + wn * 64) >> 6;
          if (rc == 0 && m < p.M && n0 + wn * 64 < p.N)
            *reinterpret_cast<float2*>(p.row_parts + ((long)m * ((p.N + 63) >> 6) + slot) * 2) = float2{e_s, e_q};
        }
      }
    }
    __builtin_amdgcn_wave_barrier();   // this wave's reads are issued (LDS is in-order per wave) before the next half's writes
  }
  // shapes without 16-byte aligned rows (N % 8 != 0, e.g. the 3-channel image conv) are routed to the v1 kernel by the host
}


// Split-K second pass: out[m, n..n+7] = epilogue( sum_s slab[s][m][n..] ) in a fixed split order (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
  using V8 = typename Vec<T>::v8;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int n8 = p.N / 8;
  if (idx >= (long)p.M * n8) return;
  const int m = (int)(idx / n8), n = (int)(idx - (long)m * n8) * 8;
  float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int sidx = 0; sidx < p.splits; ++sidx) {
    const float* sp = p.slab + ((long)sidx * p.M + m) * p.N + n;
    f32x4 a, b;
    if (p.slab_nt >= 2) {
      a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp));
      b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + 4));
    } else {
      a = *reinterpret_cast<const f32x4*>(sp); b = *reinterpret_cast<const f32x4*>(sp + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] += a[j]; o[4 + j] += b[j]; }
  }
  const T* bias = reinterpret_cast<const T*>(p.bias);
  const T* bias2 = reinterpret_cast<const T*>(p.bias2);
  const T* res = reinterpret_cast<const T*>(p.residual);
  if (bias) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] += p.bias_per_row ? to_f32(bias[m]) : to_f32(bias[n + j]);
  }
  if (bias2) {
    const T* b2 = bias2 + (long)(m / p.bias2_rpg) * p.bias2_ld + n;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] += to_f32(b2[j]);
  }
  const float rs = (p.rowscale ? p.rowscale[m] * p.alpha : p.alpha) * ((n < p.lead_cols) ? p.lead_alpha : 1.0f);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] *= rs;
  if (p.act == ACT_GELU_PRE) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = gelu_erf_f(o[j]);
  }
  if (res) {
    const T* rp = res + (long)m * p.ldr + n;
    if (p.res_vec_ok) {
      V8 r8 = ld8<T>(rp);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += to_f32(r8[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += to_f32(rp[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (p.act == ACT_SILU) o[j] = silu_f(o[j]);
    else if (p.act == ACT_RELU) o[j] = fmaxf(o[j], 0.0f);
    else if (p.act == ACT_GELU) o[j] = gelu_erf_f(o[j]);
  }
  if (p.out_f32) {
    float* cp = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
    *reinterpret_cast<f32x4*>(cp) = f32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(cp + 4) = f32x4{o[4], o[5], o[6], o[7]};
  } else {
    V8 w;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = from_f32<T>(o[j]);
    st8<T>(reinterpret_cast<T*>(p.C) + (long)m * p.ldc + n, w);
  }
}

// GemmArgs::row_parts for the routing targets whose epilogue does not emit it (big tile, split-K, exact-fit, row-stationary kernels):
// one pass over the finished C -- 8 lanes per (row, 64-column block), 16 bytes per lane -- in the same layout and the same
// summation tree as gemm2_kernel's EMIT epilogue (lane-local sum of 8 values, then sum8_dpp).
template <typename T>
__global__ __launch_bounds__(256) void row_parts_kernel(const T* __restrict__ C, long ldc, float* __restrict__ parts, int M, int N) {
  using V8 = typename Vec<T>::v8;
  const int P = (N + 63) >> 6;
  const long g = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;       // (row, slot) group of 8 lanes
  const int rc = threadIdx.x & 7;
  const bool on = g < (long)M * P;
  const int m = on ? (int)(g / P) : 0, slot = on ? (int)(g - (long)m * P) : 0;
  const int n = slot * 64 + rc * 8;
  float e_s = 0.0f, e_q = 0.0f;
  if (on && n < N) row_sums8<T>(ld8<T>(C + (long)m * ldc + n), e_s, e_q);
  e_s = sum8_dpp(e_s);
  e_q = sum8_dpp(e_q);
  if (on && rc == 0) *reinterpret_cast<float2*>(parts + ((long)m * P + slot) * 2) = float2{e_s, e_q};
}

// gemm_variant: 0 v1 (register-staged 128x128), 1 / 2 v2 (LDS-DMA 128x128) with 1 / 2 LDS stages, 3 auto among v2 only,
// 4 / 5 force the 256x320 / 128x320 kernel of gemm3.hip wherever it is applicable, 6 auto over everything (default)
static int g_gemm_variant = 6;
static int g_split_k = 1;        // 0: never split K, 1: auto
static int g_conv_fast = 1;      // 0: always use the general (per-thread tap) conv gather
static int g_v3_min_tiles = 192; // auto: smallest grid (workgroups, 1 per CU) worth the big-tile kernel
// Kernel picked by the last hallo_gemm / hallo_conv3x3_nhwc call, for per-symbol profiling (bench.py):
// 1000 * LNF (gemm2: 0 none, 1 in-loop LayerNorm statistics, 2 external) + 100 * kernel (1 gemm_kernel, 2 gemm2_kernel,
// 3 gemm3_kernel) + 10 * mode (0 gemm, 1 conv, 2 geglu) + stages / TM + 10000 * EMIT (gemm2's row_parts epilogue form: 1 / 2)
static int g_last_kernel = 0;
static int g_splitk_nt = 0;      // hallo_set_option("splitk_nt", 0 | 1 | 2): non-temporal split-K slab stores (+ loads): A/B of the slab's cache footprint with several clips in flight
static int g_split_max = 16;         // hallo_set_option("split_k_max", n): cap of the split-K factor (slab traffic grows with it; under concurrency fewer, longer workgroups cost less than they do alone)
static int g_stage_min_tiles = 640;  // hallo_set_option("gemm_stage_min_tiles", n): grids of >= n tiles take the 1-stage 128x128 kernel (4 workgroups per CU), smaller ones the 2-stage form
static int g_gemm4_min_nk = 40;  // hallo_set_option("gemm4_min_nk", n): shortest K loop (64-deep steps) the auto rule gives to gemm4.hip (A/B)
static int g_gemm4 = 1;          // hallo_set_option("gemm4", 0 off | 1 auto rule | 2 every problem gemm4.hip covers): exact-fit / stream-K kernel
static int g_last_splits = 1;    // hallo_get_option("last_gemm_splits"): split-K factor of the last launch (gemm4: 1000 + parts of a tail tile, 1 = none)
static int g_gemm_rs = 2;        // hallo_set_option("gemm_rs", 0 | 1 | 2): row-stationary kernels for eligible K = 320 / 640 shapes (1: gemm_rs.hip only, 2: gemm_rs2.hip at K = 320)


static int g_producer_stats = 1; // hallo_set_option("producer_stats", 0 | 1): read by the HOST side (hallo_amd/models): ask producers for row_parts / stats at all (A/B)
static int g_row_parts = 1;      // hallo_set_option("row_parts", 0 | 1 | 2): 0 = GemmArgs::row_parts always by the extra pass (A/B of the EMIT epilogue); 2 = always the four-workgroups-per-CU form of the 1-stage EMIT kernel

template <typename T>
static int launch_gemm_impl(GemmArgs a, bool conv, bool geglu, int batch, void* ws, int64_t ws_bytes, hipStream_t st, bool* emitted) {
  const int nk = (a.K + BK - 1) / BK;
  a.splits = 1; a.nk_per_split = 0; a.slab = nullptr; a.slab_nt = g_splitk_nt;
  int v = a.vec_ok ? g_gemm_variant : 0;
  const bool lnf = a.ln_colsum != nullptr;      // fused LayerNorm: 128x128 LDS-DMA kernel only, no split-K
  const bool gelu = a.act >= ACT_GELU;          // GELU epilogues (wav2vec2 front-end): own instantiation of the 128x128 kernel
  g_last_splits = 1;                            // (before the row-stationary early returns: they never split)
  if (g_gemm_rs >= 2 && v >= 3 && a.vec_ok && gemm_rs2_eligible(a, conv, geglu, batch)) {
    // 5xx: gemm_rs2_kernel (K = 320, pipelined epilogue): + 10 * mode (0 gemm, 2 geglu) + 1, + 1000 with LayerNorm
    g_last_kernel = 500 + (geglu ? 20 : 0) + 1 + (lnf ? 1000 : 0);
    return launch_gemm_rs2<T>(a, geglu, st);
  }
  if (a.kv_out) return -22;       // only gemm_rs2.hip writes head-major K / V (hallo_gemm_kv_split_ok)
  if (g_gemm_rs && v >= 3 && a.vec_ok && gemm_rs_eligible(a, conv, geglu, batch)) {
    // 4xx: gemm_rs_kernel (A rows in registers, W streamed): + 10 * mode (0 gemm, 2 geglu) + (1: K = 320, 2: K = 640), + 1000 with LayerNorm
    g_last_kernel = 400 + (geglu ? 20 : 0) + (a.K == 320 ? 1 : 2) + (lnf ? 1000 : 0);
    return launch_gemm_rs<T>(a, geglu, st);
  }
  g_last_splits = 1;
  // ---- gemm4.hip: 128 x 160 tiles, one persistent workgroup per CU, stream-K tail (mid-size problems of the 32x32 ... 8x8 levels) ----
  if (g_gemm4 && v >= 3 && !conv && !geglu && !gelu && batch == 1 && a.vec_ok && !a.bias_per_row && a.act <= ACT_RELU && ws &&
      a.ln_parts == 0 && (!lnf || (a.ln_stats != nullptr && !(reinterpret_cast<uintptr_t>(a.ln_colsum) & 15) && !(a.N & 7)))) {
    G4Sched gs;
    // the last 64 KB of the workspace hold the arrival counters (zero between launches): split-K slabs stay below them
    if (gemm4_plan(a, ws_bytes, &gs)) {
      const int t4 = gs.tiles_m * gs.tiles_n;
      // Auto rule, from tools/cbench/g4.sh (hot, per shape against the kernels it replaces: profiles/r4_gemm4_ab.txt) AND an end-to-end
      // A/B on one box (bench.py --set-option gemm4=0 against the rule "every one-round problem with K >= 1280":
      // 13.92 against 13.81 frames/s, profiles/r4_gemm4_e2e_ab.json).  Hot, the kernel wins where its tiles fill the chip in ONE round:
      // 4096 x 1280 x 5120 + residual 63 us against 82 (gemm3 split 4 + reduce), 4096 x 1280 x 1280 (+ residual) 21-23 against 25.
      // Inside the step (operands and residual cold) only the long-K case keeps its margin (74 against 87 us); at K = 1280 one
      // workgroup per CU exposes the cold prologue, the residual round trip and the store drain that the 128x128 kernel's
      // co-resident workgroups hide (37 against 35 us).  With two and more rounds per workgroup (4096 x 3840 x 1280: 65 against 61
      // hot) the exposed epilogues cost more than the quantisation they remove, and a K-split tail loses to the partly filled
      // round it replaces.  So: one round, K >= 2560.
      const bool one_round = (gs.dp == 1 && gs.R == 0) || (gs.dp == 0 && gs.parts == 1 && t4 >= 192);
      const bool take = (g_gemm4 == 2 || (one_round && nk >= g_gemm4_min_nk && a.N % 160 == 0)) &&
                        (gs.parts == 1 || a.ws_zeroed);      // a K-split tail counts arrivals in the workspace's counter tail: zero or no split
      if (take) {
        g_last_kernel = 600 + (lnf ? 2000 : 0);
        g_last_splits = gs.parts > 1 ? 1000 + gs.parts : 1;
        launch_gemm4<T>(a, gs, ws, ws_bytes, st);
        HALLO_CHECK_LAUNCH();
        return 0;
      }
    }
  }
  if (ws_bytes > 65536) ws_bytes -= 65536;      // (the counters of gemm4.hip)
  if (gelu) {
    if (conv || geglu || lnf) return -22;
    if (v >= 4) v = 3;
  }
  // LayerNorm-fused problems run on the 128x128 kernel -- or on the big tile (whose epilogue applies rstd * (acc - mean * G[n]))
  // when the caller supplies the row statistics and the column sums are 16-byte aligned; split-K is off for them
  const bool lnf3 = lnf && a.ln_stats != nullptr && !(reinterpret_cast<uintptr_t>(a.ln_colsum) & 15) && !(a.N & 3) && !conv;
  if (lnf) {
    if (!a.vec_ok || conv) return -22;
    if (!(lnf3 && v >= 4)) v = (v == 1 || v == 2) ? v : 3;
  }

  // ---- big-tile kernel (gemm3.hip): 256x320 or 128x320 output tiles, one workgroup per CU ----
  if (v >= 4 && (!lnf || lnf3)) {
    const bool ok3 = (a.K % 64 == 0) && (!conv || a.conv_fast) && a.N >= 160;
    const int tn3 = geglu ? (a.N + 159) / 160 : (a.N + 319) / 320;
    const float n_eff = (float)a.N / (float)(tn3 * (geglu ? 160 : 320));
    const int t256 = ((a.M + 255) / 256) * tn3 * batch, t128 = ((a.M + 127) / 128) * tn3 * batch;
    int tm = 0, sp = 1;
    // Split-K factor that brings a grid of `tiles` workgroups to ~one per CU (long K only; fp32 slabs + reduce pass).
    auto split_for = [&](int tiles) {
      if (tiles >= g_v3_min_tiles || geglu || lnf || batch != 1 || !g_split_k || !ws || nk < 16) return 1;
      int f = (256 + tiles - 1) / tiles;
      if (f > nk / 8) f = nk / 8;
      if (f > 8) f = 8;
      if (f > g_split_max) f = g_split_max;
      while (f > 1 && (int64_t)f * a.M * a.N * 4 > ws_bytes) --f;
      return f < 1 ? 1 : f;
    };
    if (ok3) {
      if (v == 4) { tm = 2; sp = split_for(t256); }
      else if (v == 5) { tm = 1; sp = split_for(t128); }
      else if (n_eff > 0.8f) {
        // Auto rule, from tools/kernel_bench.py on MI355X (profiles/r1_kernel_bench_gemm_variants.txt): the big tile pays
        // when the K loop is long enough to amortise its prologue / 6-pass epilogue with one workgroup per CU:
        //   conv3x3 (K = 9 Cin >= 576): always; 256-row tiles when the grid (after split-K) keeps >= 64 K steps per
        //   workgroup, else 128-row tiles;  GEMM / GEGLU: K >= 1280 and a grid that fills the chip.
        // One workgroup per CU: a grid of g workgroups runs in ceil(g / 256) rounds, so e.g. 288 tiles cost two full
        // rounds (56 % efficiency).  Require >= 85 % of the last round to be filled.
        auto fills = [&](int g) { const int rounds = (g + 255) / 256; return g >= g_v3_min_tiles && g * 100 >= rounds * 256 * 84; };
        const int sp256 = split_for(t256), sp128 = split_for(t128);
        if (conv && a.stride == 1) {
          if (fills(t256 * sp256) && (sp256 == 1 || nk / sp256 >= 40)) { tm = 2; sp = sp256; }
          else if (t128 <= 64 && fills(t128 * sp128)) { tm = 1; sp = sp128; }     // 8x8 feature maps: 128-row tiles, deep split
        } else if (!conv && geglu) {
          // GEGLU (r2 A/B with the LayerNorm epilogue, tools/cbench/g3.sh): the 4 x wider W stream per output column pays from
          // K = 640 on, and a multi-round grid wins even at 75 % fill of its last round; 128-row tiles for the 8 x 8 maps
          if (nk >= 10 && (fills(t256) || t256 >= 512)) { tm = 2; sp = 1; }
          else if (nk >= 20 && t128 >= 224 && t128 <= 256) { tm = 1; sp = 1; }
        } else if (!conv && nk >= 20) {
          if (fills(t256)) { tm = 2; sp = 1; }
          else if (nk >= 40 && batch == 1 && !a.out_f32 && a.tiles_m * a.tiles_n < 384) {
            // long-K projections of the 16x16 / 8x8 levels (ff.net[2] at K = 2560 / 5120), where the 128x128 kernel would split K
            // as well: cold, the split big tile wins by 7-10 us (profiles/r3_gemm_cold_bench.json: 77 vs 87, 38 vs 46, 40 vs 48 us)
            if (sp256 > 1 && fills(t256 * sp256) && nk / sp256 >= 20) { tm = 2; sp = sp256; }
            else if (sp128 > 1 && fills(t128 * sp128) && nk / sp128 >= 5) { tm = 1; sp = sp128; }
          }
        }
      }
    }
    if (tm) {
      a.tiles_m = (a.M + 128 * tm - 1) / (128 * tm);
      a.tiles_n = tn3;
      if (sp > 1) {
        a.nk_per_split = (nk + sp - 1) / sp;
        a.splits = (nk + a.nk_per_split - 1) / a.nk_per_split;
        a.slab = reinterpret_cast<float*>(ws);
      }
      g_last_splits = a.splits;
      g_last_kernel = 300 + 10 * (geglu ? 2 : (conv ? 1 : 0)) + tm + (lnf ? 2000 : 0);
      launch_gemm3<T>(a, geglu ? 2 : (conv ? 1 : 0), tm, batch, st);
      HALLO_CHECK_LAUNCH();
      if (a.splits > 1) {
        const long n = (long)a.M * (a.N / 8);
        hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
        HALLO_CHECK_LAUNCH();
      }
      return 0;
    }
    v = 3;
  }

  const int tiles = a.tiles_m * a.tiles_n;
  // auto: one LDS stage (4 workgroups per CU hide each other's load latency) when the grid fills the chip several
  // times over, two stages (in-workgroup prefetch) for small grids
  if (v == 3) v = (tiles * batch >= g_stage_min_tiles) ? 1 : 2;
  if (v != 0 && !geglu && !lnf && batch == 1 && g_split_k && ws && tiles < 384 && nk >= 32) {
    // small grids with a long K loop (8x8 / 16x16 feature maps, K up to 23040): split K so that >= ~768 workgroups
    // are in flight; partial sums go to an fp32 slab and a second pass applies the epilogue in a fixed order
    int sp = (768 + tiles - 1) / tiles;
    if (sp > nk / 4) sp = nk / 4;
    if (sp > 16) sp = 16;
    if (sp > g_split_max) sp = g_split_max;
    while (sp > 1 && (int64_t)sp * a.M * a.N * 4 > ws_bytes) --sp;
    if (sp > 1) {
      a.nk_per_split = (nk + sp - 1) / sp;
      a.splits = (nk + a.nk_per_split - 1) / a.nk_per_split;
      a.slab = reinterpret_cast<float*>(ws);
    }
  }
  dim3 grid(tiles, a.splits, batch), block(256);
  // the 128 x 128 kernel's own row_parts epilogue: plain GEMM, one K pass, 16-bit output
  const bool emit2 = g_row_parts && a.row_parts && v != 0 && a.splits == 1 && batch == 1 && !a.out_f32 && !conv && !geglu && !gelu && !lnf;
  g_last_splits = a.splits;
  g_last_kernel = (v == 0 ? 100 : 200) + 10 * (geglu ? 2 : (conv ? 1 : (gelu && v != 0 ? 3 : 0))) + (v == 0 ? 0 : v) +
                  1000 * (v != 0 && lnf ? (a.ln_stats ? 2 : 1) : 0);
  if (v == 0) {
    if (geglu) hipLaunchKernelGGL((gemm_kernel<T, false, true>), grid, block, 0, st, a);
    else if (conv) hipLaunchKernelGGL((gemm_kernel<T, true, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, block, 0, st, a);
  } else if (v == 1) {
    if (lnf && a.ln_stats && geglu) hipLaunchKernelGGL((gemm2_kernel<T, 2, 1, 2>), grid, block, 0, st, a);
    else if (lnf && a.ln_stats) hipLaunchKernelGGL((gemm2_kernel<T, 0, 1, 2>), grid, block, 0, st, a);
    else if (lnf && geglu) hipLaunchKernelGGL((gemm2_kernel<T, 2, 1, 1>), grid, block, 0, st, a);
    else if (lnf) hipLaunchKernelGGL((gemm2_kernel<T, 0, 1, 1>), grid, block, 0, st, a);
    else if (geglu) hipLaunchKernelGGL((gemm2_kernel<T, 2, 1, 0>), grid, block, 0, st, a);
    else if (gelu) hipLaunchKernelGGL((gemm2_kernel<T, 3, 1, 0>), grid, block, 0, st, a);
    else if (conv) hipLaunchKernelGGL((gemm2_kernel<T, 1, 1, 0>), grid, block, 0, st, a);
    else if (emit2) {
      const int r4 = (tiles + 1023) / 1024, r3 = (tiles + 767) / 768;          // rounds at four / three workgroups per CU
      if (r4 < r3 || g_row_parts == 2) { hipLaunchKernelGGL((gemm2_kernel<T, 0, 1, 0, 1>), grid, block, 0, st, a); g_last_kernel += 10000; }
      else { hipLaunchKernelGGL((gemm2_kernel<T, 0, 1, 0, 2>), grid, block, 0, st, a); g_last_kernel += 20000; }
      *emitted = true;
    }
    else hipLaunchKernelGGL((gemm2_kernel<T, 0, 1, 0>), grid, block, 0, st, a);
  } else {
    if (lnf && a.ln_stats && geglu) hipLaunchKernelGGL((gemm2_kernel<T, 2, 2, 2>), grid, block, 0, st, a);
    else if (lnf && a.ln_stats) hipLaunchKernelGGL((gemm2_kernel<T, 0, 2, 2>), grid, block, 0, st, a);
    else if (lnf && geglu) hipLaunchKernelGGL((gemm2_kernel<T, 2, 2, 1>), grid, block, 0, st, a);
    else if (lnf) hipLaunchKernelGGL((gemm2_kernel<T, 0, 2, 1>), grid, block, 0, st, a);
    else if (geglu) hipLaunchKernelGGL((gemm2_kernel<T, 2, 2, 0>), grid, block, 0, st, a);
    else if (gelu) hipLaunchKernelGGL((gemm2_kernel<T, 3, 2, 0>), grid, block, 0, st, a);
    else if (conv) hipLaunchKernelGGL((gemm2_kernel<T, 1, 2, 0>), grid, block, 0, st, a);
    else if (emit2) { hipLaunchKernelGGL((gemm2_kernel<T, 0, 2, 0, 1>), grid, block, 0, st, a); g_last_kernel += 10000; *emitted = true; }
    else hipLaunchKernelGGL((gemm2_kernel<T, 0, 2, 0>), grid, block, 0, st, a);
  }
  HALLO_CHECK_LAUNCH();
  if (a.splits > 1) {
    const long n = (long)a.M * (a.N / 8);
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((n + 255) / 256)), block, 0, st, a);
    HALLO_CHECK_LAUNCH();
  }
  return 0;
}

template <typename T>
static int launch_gemm(GemmArgs a, bool conv, bool geglu, int batch, void* ws, int64_t ws_bytes, hipStream_t st) {
  bool emitted = false;
  const int rc = launch_gemm_impl<T>(a, conv, geglu, batch, ws, ws_bytes, st, &emitted);
  if (rc != 0 || !a.row_parts || emitted) return rc;
  // whatever kernel took the problem (incl. its split-K reduce pass) has written C on this stream: one pass over it
  const long groups = (long)a.M * ((a.N + 63) >> 6);
  hipLaunchKernelGGL((row_parts_kernel<T>), dim3((unsigned)((groups * 8 + 255) / 256)), dim3(256), 0, st,
                     reinterpret_cast<const T*>(a.C), a.ldc, a.row_parts, a.M, a.N);
  HALLO_CHECK_LAUNCH();
  return 0;
}

static void set_vec_flags(GemmArgs& a, int esz_out) {
  // C rows: 4 consecutive elements per store -> ldc % 4 and an 8-byte (fp32: 16-byte) aligned base
  const uintptr_t cp = reinterpret_cast<uintptr_t>(a.C);
  a.vec_ok = (a.N % 8 == 0) && (a.ldc % 8 == 0) && (cp % 16 == 0) && ((a.sC * esz_out) % 16 == 0);
  a.bias_vec_ok = a.bias && reinterpret_cast<uintptr_t>(a.bias) % 16 == 0;
  a.bias2_vec_ok = a.bias2 && reinterpret_cast<uintptr_t>(a.bias2) % 16 == 0 && a.bias2_ld % 8 == 0;
  const uintptr_t rp = reinterpret_cast<uintptr_t>(a.residual);
  a.res_vec_ok = a.residual && (a.ldr % 8 == 0) && (rp % 16 == 0) && ((a.sR * 2) % 16 == 0);
}

}  // namespace hallo

using namespace hallo;

extern "C" int hallo_gemm(const hallo_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->B || !d->C) return -22;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || (d->K & 7)) return -22;
  if (d->batch < 1) return -22;
  if (d->act < ACT_NONE || d->act > ACT_GELU_PRE) return -22;
  if ((d->lda & 7) || (d->ldb & 7)) return -22;
  if (d->geglu && (d->rowscale || d->residual || d->bias2 || d->out_f32 || d->bias_per_row || d->lead_cols > 0)) return -22;
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
  a.lead_cols = d->lead_cols > 0 ? d->lead_cols : 0; a.lead_alpha = d->lead_cols > 0 ? d->lead_alpha : 1.0f;
  if (a.lead_cols & 7) return -22;
  a.ln_colsum = d->ln_colsum; a.ln_eps = d->ln_eps; a.ln_stats = d->ln_colsum ? d->ln_stats : nullptr;
  if (a.ln_colsum && (d->batch != 1 || d->out_f32 || d->bias_per_row)) return -22;
  a.ln_parts = (a.ln_colsum && a.ln_stats && d->ln_parts > 0) ? d->ln_parts : 0;
  if (d->ln_parts < 0 || (d->ln_parts > 0 && (!(d->ln_colsum && d->ln_stats) || (reinterpret_cast<uintptr_t>(d->ln_stats) & 15)))) return -22;
  // the partial sums are one (sum, sum of squares) pair per 64-column block of A's K columns -- nothing else makes mean = sum / K
  // right -- and a tile's rows x ln_parts pairs are staged in the (idle) operand LDS behind the K loop: 128 rows x 32 pairs fill the
  // 32 KB of the 1-stage 128 x 128 kernel (K <= 2048; the widest LayerNorm on the path is 1280).  Larger K: hallo_row_stats.
  if (d->ln_parts > 0 && (d->ln_parts != (d->K + 63) / 64 || d->ln_parts > 32)) return -22;
  a.ws_zeroed = d->workspace_zeroed != 0;
  a.row_parts = d->row_parts;
  a.kv_out = d->kv_out; a.kv_col0 = d->kv_col0; a.kv_L = d->kv_rows_per_image; a.kv_tstride = d->kv_tensor_stride;
  if (a.kv_out) {
    // K and V of 8 heads x 40 behind the q columns, whole images of kv_rows_per_image rows, LayerNorm-fused dtype output, nothing
    // else in the epilogue; 16-byte stores
    if (d->geglu || d->batch != 1 || d->out_f32 || d->residual || d->rowscale || d->row_parts || !d->ln_colsum || d->ln_stats) return -22;
    if (d->kv_col0 < 0 || (d->kv_col0 & 31) || d->N - d->kv_col0 != 640 || d->ldc < d->kv_col0 || d->lead_cols > d->kv_col0) return -22;
    if (d->kv_rows_per_image <= 0 || d->M % d->kv_rows_per_image || (d->kv_tensor_stride & 7) || (reinterpret_cast<uintptr_t>(d->kv_out) & 15)) return -22;
  }
  if (a.row_parts && (d->batch != 1 || d->out_f32 || d->geglu || (d->N & 7) || (reinterpret_cast<uintptr_t>(d->row_parts) & 15))) return -22;
  a.tiles_m = (d->M + BM - 1) / BM;
  a.tiles_n = d->geglu ? (d->N + 63) / 64 : (d->N + BN - 1) / BN;
  a.H = a.W = a.Cin = a.OH = a.OW = a.stride = a.pad_t = a.pad_l = a.upsample = 0;
  a.conv_fast = 0;
  set_vec_flags(a, d->out_f32 ? 4 : 2);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == DT_F16) return launch_gemm<_Float16>(a, false, d->geglu != 0, d->batch, d->workspace, d->workspace_bytes, st);
  if (d->dtype == DT_BF16) return launch_gemm<__bf16>(a, false, d->geglu != 0, d->batch, d->workspace, d->workspace_bytes, st);
  return -22;
}

extern "C" int hallo_gemm4_schedule(int M, int N, int K, int64_t workspace_bytes, int force_parts, int* sched) {
  // The schedule csrc/gemm4.hip would run a plain M x N x K problem with (pure function of its arguments and of the device's CU
  // count; 256 without a device): sched[0..7] = tiles_m, tiles_n, K steps, workgroups, data-parallel rounds, tail tiles, parts per
  // tail tile, K steps per part.  Returns 1 if the kernel covers the problem, 0 if not, -22 on bad arguments.
  if (M <= 0 || N <= 0 || K <= 0 || !sched) return -22;
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.M = M; a.N = N; a.K = K; a.bias2_rpg = 1; a.res_vec_ok = 1;
  G4Sched gs;
  if (!gemm4_plan(a, workspace_bytes, &gs)) return 0;
  (void)force_parts;
  sched[0] = gs.tiles_m; sched[1] = gs.tiles_n; sched[2] = gs.nk; sched[3] = gs.G; sched[4] = gs.dp; sched[5] = gs.R;
  sched[6] = gs.parts; sched[7] = gs.per;
  return 1;
}

extern "C" int hallo_gemm_fuses_row_stats(int M, int N, int K, int geglu, int bias2_rows_per_group, int lead_cols) {
  if (!g_gemm_rs || g_gemm_variant < 3) return 0;
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N;
  a.bias2 = bias2_rows_per_group > 0 ? reinterpret_cast<const void*>(16) : nullptr;
  a.bias2_rpg = bias2_rows_per_group > 0 ? bias2_rows_per_group : 1;
  a.lead_cols = lead_cols; a.splits = 1; a.act = ACT_NONE;
  a.A = a.C = reinterpret_cast<void*>(16);      // alignment checks of the eligibility rules: hallo_gemm requires 16-byte pointers anyway
  if (g_gemm_rs >= 2 && gemm_rs2_eligible(a, false, geglu != 0, 1)) return 1;
  return gemm_rs_eligible(a, false, geglu != 0, 1) ? 1 : 0;
}

extern "C" int hallo_gemm_kv_split_ok(int M, int N, int K, int kv_col0) {
  if (g_gemm_rs < 2 || g_gemm_variant < 3 || N - kv_col0 != 640 || (kv_col0 & 31) || kv_col0 < 0) return 0;
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = kv_col0 > 0 ? kv_col0 : 8;
  a.bias2_rpg = 1; a.lead_cols = kv_col0; a.splits = 1; a.act = ACT_NONE;
  a.ln_colsum = reinterpret_cast<const float*>(16);
  a.A = a.C = reinterpret_cast<void*>(16);
  return gemm_rs2_eligible(a, false, false, 1) ? 1 : 0;
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
  a.lead_cols = 0; a.lead_alpha = 1.0f;
  a.kv_out = nullptr; a.kv_col0 = 0; a.kv_L = 0; a.kv_tstride = 0;
  a.ln_colsum = nullptr; a.ln_eps = 0.0f; a.ln_stats = nullptr; a.ln_parts = 0; a.row_parts = nullptr; a.ws_zeroed = 0;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.N + BN - 1) / BN;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.OH = d->OH; a.OW = d->OW;
  a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.upsample = d->upsample;
  a.conv_fast = (d->Cin % 64 == 0) && !d->upsample && g_conv_fast;
  set_vec_flags(a, 2);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->dtype == DT_F16) return launch_gemm<_Float16>(a, true, false, 1, d->workspace, d->workspace_bytes, st);
  if (d->dtype == DT_BF16) return launch_gemm<__bf16>(a, true, false, 1, d->workspace, d->workspace_bytes, st);
  return -22;
}

extern "C" int hallo_set_option_norm(const char* name, int value);   // norm_elementwise.hip

static int g_rs_dbg_value = 0;

extern "C" int hallo_get_option_attn(const char* name);   // attention.hip

extern "C" void hallo_gemm4_debug_buffer(long long* p) { set_gemm4_debug_buffer(p); }   // tools/cbench: s_memtime stamps of gemm4.hip

extern "C" const char* hallo_option_names(void) {
  // every name hallo_set_option accepts (this file, attention.hip, fused_xattn.hip, norm_elementwise.hip); tests/test_abi.py
  // checks that each one round-trips through hallo_get_option / hallo_set_option
  return "gemm_variant,split_k,split_k_max,splitk_nt,v3_min_tiles,conv_fast,row_parts,producer_stats,gemm_rs,gemm4,gemm4_min_nk,"
         "gemm_stage_min_tiles,ff_fused,gemm_rs_dbg,attn40,temporal_mfma,attn_order,tok_attn,xattn_tiled,xattn_cap,gn_fused,fp8_mx";
}

extern "C" int hallo_get_option(const char* name) {
  if (!name) return -22;
  if (!strcmp(name, "gemm_variant")) return g_gemm_variant;
  if (!strcmp(name, "split_k")) return g_split_k;
  if (!strcmp(name, "v3_min_tiles")) return g_v3_min_tiles;
  if (!strcmp(name, "last_gemm_kernel")) return g_last_kernel;
  if (!strcmp(name, "last_gemm_splits")) return g_last_splits;
  if (!strcmp(name, "gemm4")) return g_gemm4;
  if (!strcmp(name, "gemm4_min_nk")) return g_gemm4_min_nk;
  if (!strcmp(name, "gemm_stage_min_tiles")) return g_stage_min_tiles;
  if (!strcmp(name, "split_k_max")) return g_split_max;
  if (!strcmp(name, "splitk_nt")) return g_splitk_nt;
  if (!strcmp(name, "gemm_rs")) return g_gemm_rs;
  if (!strcmp(name, "gemm_rs_dbg")) return g_rs_dbg_value;
  if (!strcmp(name, "ff_fused")) return ff_fused_variant();
  if (!strcmp(name, "conv_fast")) return g_conv_fast;
  if (!strcmp(name, "row_parts")) return g_row_parts;
  if (!strcmp(name, "producer_stats")) return g_producer_stats;
  return hallo_get_option_attn(name);
}

extern "C" int hallo_set_option(const char* name, int value) {
  if (!name) return -22;
  if (!strcmp(name, "gemm_variant")) { if (value < 0 || value > 6) return -22; g_gemm_variant = value; return 0; }
  if (!strcmp(name, "v3_min_tiles")) { if (value < 1) return -22; g_v3_min_tiles = value; return 0; }
  if (!strcmp(name, "conv_fast")) { if (value < 0 || value > 1) return -22; g_conv_fast = value; return 0; }
  if (!strcmp(name, "split_k")) { if (value < 0 || value > 1) return -22; g_split_k = value; return 0; }
  if (!strcmp(name, "row_parts")) { if (value < 0 || value > 2) return -22; g_row_parts = value; return 0; }
  if (!strcmp(name, "producer_stats")) { if (value < 0 || value > 1) return -22; g_producer_stats = value; return 0; }
  if (!strcmp(name, "gemm_rs")) { if (value < 0 || value > 2) return -22; g_gemm_rs = value; return 0; }
  if (!strcmp(name, "gemm4")) { if (value < 0 || value > 2) return -22; g_gemm4 = value; return 0; }
  if (!strcmp(name, "gemm4_min_nk")) { if (value < 4) return -22; g_gemm4_min_nk = value; return 0; }
  if (!strcmp(name, "gemm_stage_min_tiles")) { if (value < 0) return -22; g_stage_min_tiles = value; return 0; }
  if (!strcmp(name, "split_k_max")) { if (value < 1) return -22; g_split_max = value; return 0; }
  if (!strcmp(name, "splitk_nt")) { if (value < 0 || value > 2) return -22; g_splitk_nt = value; return 0; }
  if (!strcmp(name, "ff_fused")) {          // 0 / 1; 2.. = A/B and timing-ablation forms of a -DHALLO_ABLATIONS build
#ifdef HALLO_ABLATIONS
    if (value < 0 || value > 9) return -22;
#else
    if (value < 0 || value > 1) return -22;
#endif
    set_ff_fused_variant(value); return 0;
  }
  if (!strcmp(name, "gemm_rs_dbg")) {
    // timing ablations of the row-stationary kernels (no stores / no epilogue: WRONG results): only a library built with
    // -DHALLO_ABLATIONS (HALLO_ABLATIONS=1 python -m hallo_amd.build, what tools/cbench uses) accepts a non-zero value
#ifndef HALLO_ABLATIONS
    if (value != 0) return -22;
#endif
    if (value < 0 || value > 7) return -22;
    g_rs_dbg_value = value; set_gemm_rs_dbg(value); set_gemm_rs2_dbg(value); return 0;
  }
  return hallo_set_option_norm(name, value);
}
