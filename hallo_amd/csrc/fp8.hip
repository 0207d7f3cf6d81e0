// fp8 (OCP e4m3) projections -- BASELINE.json configs[4]: "fp8 MFMA QKV/out projections with bf16 accumulate".
//
// Replaces, when the model is switched to fp8 projections, the bf16 / fp16 GEMMs behind diffusers `Attention.to_q / to_k /
// to_v / to_out` (reached from hallo/models/mutual_self_attention.py:253-303, hallo/models/attention.py:828-884,
// hallo/models/motion_module.py:553-609):
//   hallo_quant_rows_fp8   activations [rows, C] (optionally through LayerNorm: the norm that feeds to_q|k|v) -> e4m3 bytes +
//                          one fp32 scale per ROW (absmax / 448), one read of x, one write of C bytes per row;
//                          the same kernel quantises weight rows once per model (one scale per OUTPUT CHANNEL);
//   hallo_gemm_fp8         C[M,N] = (Aq . Wq^T) * a_scale[m] * w_scale[n] (+ bias, lead-column scale, alpha, residual) on
//                          v_mfma_f32_32x32x16_fp8_fp8, fp32 accumulate, output in the model's storage type.
// gfx950 has no non-scaled fp8 MFMA with K > 16 per 32x32 instruction (the K = 64 forms are the block-scaled MX ones, whose
// per-32-element scales do not fit per-row / per-channel quantisation), so the matrix pipe runs at the bf16 rate; what fp8
// buys here is half the operand bytes through HBM, L2 and LDS.  The K = 320 / 640 projections are bound by their bf16 OUTPUT
// stream and the L2 -> LDS fabric, so the A/B is roughly neutral; the path exists because the north
// star names it, with its parity measured (tests/test_fp8_gpu.py, profiles/r2_fp8_ab.json).
#include "common.h"
#include <string.h>
#include "../../include/hallo_amd.h"

namespace hallo {

// ------------------------------------------------------------------------------------------
// Row quantisation (optionally LayerNorm first).  LPR lanes share a row, VPL 8-element vectors per lane.
// ------------------------------------------------------------------------------------------
template <typename T, int LPR, int VPL>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const T* __restrict__ x, long ldx, unsigned char* __restrict__ q,
                                                             float* __restrict__ scale, long rows, int C,
                                                             const T* __restrict__ gamma, const T* __restrict__ beta, float eps) {
  using V8 = typename Vec<T>::v8;
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (LPR - 1);
  const long row = ((long)blockIdx.x * 4 + wave) * RPW + lane / LPR;
  const bool live = row < rows;
  const T* xr = x + (live ? row : 0) * ldx;
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
  if (gamma) {      // nn.LayerNorm (two-pass statistics in registers), affine applied, result rounded to T like the module's output
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)C;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
      if (sub + LPR * i < vpr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
      }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = rsqrtf(sq / (float)C + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = sub + LPR * i;
      if (vi < vpr) {
        const V8 g = ld8<T>(gamma + vi * 8), b = ld8<T>(beta + vi * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = to_f32(from_f32<T>((v[i][e] - mean) * rstd * to_f32(g[e]) + to_f32(b[e])));
      }
    }
  }
  float amax = 0.0f;
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  const float sc = amax > 0.0f ? amax * (1.0f / 448.0f) : 1.0f;      // e4m3 max finite = 448
  const float inv = 1.0f / sc;
  if (live) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vi = sub + LPR * i;
      if (vi < vpr) {
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][4] * inv, v[i][5] * inv, w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][6] * inv, v[i][7] * inv, w1, true);
        *reinterpret_cast<int2*>(q + row * C + vi * 8) = make_int2(w0, w1);
      }
    }
    if (sub == 0) scale[row] = sc;
  }
}

// ------------------------------------------------------------------------------------------
// fp8 GEMM: 128 x 128 output tile, 4 waves as 2 x 2 of 64 x 64 (2 x 2 MFMA 32x32x16 blocks each), K step 64 (64 bytes per
// row), two LDS stages filled by LDS-DMA (16-byte pieces; the 16-byte chunk index XOR-ed with (row >> 2) & 3 through the
// SOURCE address so that the 8-byte fragment reads of 16 consecutive rows fall on 16 distinct 16-byte slots), swapped operands
// (D = W . A^T: a lane owns an output row), 16-byte stores after a half swap.
// ------------------------------------------------------------------------------------------
struct Fp8GemmArgs {
  const unsigned char* A; const unsigned char* B; void* C;
  int M, N, K;
  long lda, ldb, ldc;
  const float* a_scale; const float* w_scale;
  const void* bias; const void* residual; long ldr;
  float alpha; int lead_cols; float lead_alpha;
  int tiles_n;
};

// MX (round 6): the contraction on `v_mfma_scale_f32_32x32x64_f8f6f4` -- the block-scaled OCP-MX form, the only fp8 MFMA of gfx950 that runs
// at the fp8 rate (2 x bf16; the non-scaled 32x32x16 form has the bf16 K per instruction and the bf16 rate) -- with UNIT block scales
// (E8M0 127 = 2^0 in every scale byte): the operands stay plain per-row / per-channel-scaled e4m3 and the per-row x per-channel
// scales are applied to the fp32 accumulators in the epilogue exactly as before.  One instruction contracts the whole 64-byte K step
// of a 32 x 32 block: a lane hands over 32 consecutive bytes of its row (k half = lane / 32), read as two 16-byte LDS pieces through
// the same chunk swizzle.  The k order inside an instruction is the same function of (lane, byte) for both operands, so the sum is
// the same set of exact products in another order: results agree with the 32x32x16 form to fp32 summation noise.
template <typename T, bool MX>
__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(const Fp8GemmArgs p) {
  constexpr int BM = 128, BN = 128, BK = 64;
  constexpr int TILE = BM * BK;                                 // bytes per operand stage
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE];     // [stage][A | B]
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* const lds = (lds_u8*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = gridDim.x;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // loaders: thread owns 16-byte piece (row = wave*32 + j*16 + lane/4, physical chunk lane%4) of both operand tiles, j = 0, 1
  auto clamp32 = [](long b) { return (int)(b > 0xFFFFFFFFL ? 0xFFFFFFFFL : b); };
  const int rowsA = min(p.M - m0, BM), rowsB = min(p.N - n0, BN);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.A + (long)m0 * p.lda), 0,
                                                                        clamp32((long)(rowsA - 1) * p.lda + p.K), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.B + (long)n0 * p.ldb), 0,
                                                                        clamp32((long)(rowsB - 1) * p.ldb + p.K), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFFFu;
  unsigned a_voff[2], b_voff[2];
  int kchunk[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wave * 32 + j * 16 + (lane >> 2);
    const int logical = (lane & 3) ^ ((row >> 2) & 3);
    kchunk[j] = logical * 16;
    a_voff[j] = row < rowsA ? (unsigned)((long)row * p.lda + logical * 16) : OOB;
    b_voff[j] = row < rowsB ? (unsigned)((long)row * p.ldb + logical * 16) : OOB;
  }
  auto stage = [&](int kt, int buf) {
    const int dst = buf * 2 * TILE + wave_u * 2048;
    const bool tail = (kt + 1) * BK > p.K;           // wave-uniform; K % 64 != 0 only: pieces past K are written as zeros
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool in = !tail || kt * BK + kchunk[j] < p.K;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(lds + dst + j * 1024), 16, (int)(in ? a_voff[j] : OOB), kt * BK, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(lds + dst + TILE + j * 1024), 16, (int)(in ? b_voff[j] : OOB), kt * BK, 0, 0);
    }
  };

  f32x16 acc[2][2];   // [tn][tm]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // fragment reads: row = base + l31 (+32), logical chunk ks (16 bytes = one k16 step), half hi -> physical ks ^ ((row >> 2) & 3)
  const int sw = (l31 >> 2) & 3;
  const lds_u8* const fa = lds + (wm * 64 + l31) * BK + hi * 8;
  const lds_u8* const fb = lds + TILE + (wn * 64 + l31) * BK + hi * 8;
  typedef const __attribute__((address_space(3))) long* ldsl;
  typedef __attribute__((ext_vector_type(4))) int intx4;
  typedef __attribute__((ext_vector_type(8))) int intx8;
  typedef const __attribute__((address_space(3))) intx4* ldsq;
  // MX fragments: the 32 bytes k = 32 hi .. 32 hi + 31 of row (base + l31) = logical 16-byte chunks 2 hi, 2 hi + 1
  const lds_u8* const fa32 = lds + (wm * 64 + l31) * BK;
  const lds_u8* const fb32 = lds + TILE + (wn * 64 + l31) * BK;
  const int c0 = ((2 * hi) ^ sw) * 16, c1 = ((2 * hi + 1) ^ sw) * 16;
  auto frag32 = [&](const lds_u8* base) {
    const intx4 lo = *(ldsq)(base + c0), hi4 = *(ldsq)(base + c1);
    return __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto compute = [&](int buf) {
    const int bo = buf * 2 * TILE;
    if (MX) {
      constexpr int UNIT = 0x7F7F7F7F;       // E8M0 scale bytes: 2^(127 - 127) = 1 for every 32-element block
      const intx8 a0 = frag32(fa32 + bo), a1 = frag32(fa32 + bo + 32 * BK);
      const intx8 w0 = frag32(fb32 + bo), w1 = frag32(fb32 + bo + 32 * BK);
      acc[0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w0, a0, acc[0][0], 0, 0, 0, UNIT, 0, UNIT);
      acc[0][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w0, a1, acc[0][1], 0, 0, 0, UNIT, 0, UNIT);
      acc[1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w1, a0, acc[1][0], 0, 0, 0, UNIT, 0, UNIT);
      acc[1][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w1, a1, acc[1][1], 0, 0, 0, UNIT, 0, UNIT);
      return;
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int co = (ks ^ sw) * 16;
      const long a0 = *(ldsl)(fa + bo + co), a1 = *(ldsl)(fa + bo + 32 * BK + co);
      const long w0 = *(ldsl)(fb + bo + co), w1 = *(ldsl)(fb + bo + 32 * BK + co);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(w0, a0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(w0, a1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(w1, a0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(w1, a1, acc[1][1], 0, 0, 0);
    }
  };
  const int nk = (p.K + BK - 1) / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      stage(kt + 1, (kt + 1) & 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile kt landed, tile kt+1 (4 DMA loads) in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    compute(kt & 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // stage kt&1 is free for tile kt+2
  }

  // ---- epilogue: lane owns row m (per tm), D rows n = tn*32 + 8g + 4hi + 0..3 ----
  const T* bias = reinterpret_cast<const T*>(p.bias);
  const T* res = reinterpret_cast<const T*>(p.residual);
  T* Cp = reinterpret_cast<T*>(p.C);
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int m = m0 + wm * 64 + tm * 32 + l31;
    const bool mlive = m < p.M;
    const float sa = mlive ? p.a_scale[m] : 0.0f;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      unsigned pk[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + tn * 32 + 8 * g + 4 * hi;
        typename Vec<T>::v4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int nn = min(n + j, p.N - 1);
          float v = acc[tn][tm][g * 4 + j] * sa * p.w_scale[nn];
          if (bias) v += to_f32(bias[nn]);
          v *= p.alpha * (n + j < p.lead_cols ? p.lead_alpha : 1.0f);
          if (res && mlive && n + j < p.N) v += to_f32(res[(long)m * p.ldr + n + j]);
          o[j] = from_f32<T>(v);
        }
        const uint2 u = __builtin_bit_cast(uint2, o);
        pk[g][0] = u.x; pk[g][1] = u.y;
      }
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        const auto x = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
        const auto y = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
        const int nc = n0 + wn * 64 + tn * 32 + 8 * g + 8 * hi;
        if (mlive && nc < p.N) *reinterpret_cast<uint4*>(Cp + (long)m * p.ldc + nc) = make_uint4(x[0], y[0], x[1], y[1]);
      }
    }
  }
}

}  // namespace hallo

using namespace hallo;

static int g_fp8_mx = 1;
extern "C" int hallo_set_option_fp8(const char* name, int value) {
  if (name && !strcmp(name, "fp8_mx")) { if (value < 0 || value > 1) return -22; g_fp8_mx = value; return 0; }
  return -22;
}
extern "C" int hallo_get_option_fp8(const char* name) {
  if (name && !strcmp(name, "fp8_mx")) return g_fp8_mx;
  return -22;
}

extern "C" int hallo_quant_rows_fp8(const void* x, int64_t ldx, void* q, float* scale, int64_t rows, int C, const void* gamma,
                                    const void* beta, float eps, int dtype, void* stream) {
  if (!x || !q || !scale || rows <= 0 || C <= 0 || (C & 7) || C > 1536 || ldx < C || (ldx & 7)) return -22;
  if ((gamma == nullptr) != (beta == nullptr)) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int vpr = C / 8;
  // lanes per row: smallest power of two with <= 5 vectors per lane
  int lpr = 8;
  while (lpr < 64 && (vpr + lpr - 1) / lpr > 5) lpr *= 2;
  if ((vpr + lpr - 1) / lpr > 5) return -22;
#define HALLO_Q(TT, LPRv) hipLaunchKernelGGL((quant_rows_fp8_kernel<TT, LPRv, 5>), dim3((unsigned)((rows + 4 * (64 / LPRv) - 1) / (4 * (64 / LPRv)))), \
    dim3(256), 0, st, reinterpret_cast<const TT*>(x), (long)ldx, reinterpret_cast<unsigned char*>(q), scale, (long)rows, C, \
    reinterpret_cast<const TT*>(gamma), reinterpret_cast<const TT*>(beta), eps)
  if (dtype == DT_F16) { if (lpr == 8) HALLO_Q(_Float16, 8); else if (lpr == 16) HALLO_Q(_Float16, 16); else if (lpr == 32) HALLO_Q(_Float16, 32); else HALLO_Q(_Float16, 64); }
  else if (dtype == DT_BF16) { if (lpr == 8) HALLO_Q(__bf16, 8); else if (lpr == 16) HALLO_Q(__bf16, 16); else if (lpr == 32) HALLO_Q(__bf16, 32); else HALLO_Q(__bf16, 64); }
  else return -22;
#undef HALLO_Q
  HALLO_CHECK_LAUNCH();
  return 0;
}

extern "C" int hallo_gemm_fp8(const hallo_gemm_fp8_desc* d, void* stream) {
  if (!d || !d->A || !d->B || !d->C || !d->a_scale || !d->w_scale) return -22;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || (d->K & 15) || (d->N & 7)) return -22;
  if ((d->lda & 15) || (d->ldb & 15) || (d->ldc & 7) || (reinterpret_cast<uintptr_t>(d->C) & 15)) return -22;
  if ((reinterpret_cast<uintptr_t>(d->A) & 15) || (reinterpret_cast<uintptr_t>(d->B) & 15)) return -22;
  if (d->lead_cols < 0) return -22;
  Fp8GemmArgs a;
  a.A = reinterpret_cast<const unsigned char*>(d->A); a.B = reinterpret_cast<const unsigned char*>(d->B); a.C = d->C;
  a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
  a.a_scale = d->a_scale; a.w_scale = d->w_scale; a.bias = d->bias; a.residual = d->residual; a.ldr = d->ldr;
  a.alpha = d->alpha; a.lead_cols = d->lead_cols; a.lead_alpha = d->lead_cols > 0 ? d->lead_alpha : 1.0f;
  a.tiles_n = (d->N + 127) / 128;
  const int tiles = ((d->M + 127) / 128) * a.tiles_n;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // hallo_set_option("fp8_mx", 1 (default) | 0): the MX-rate contraction (unit block scales) or the non-scaled 32x32x16 form (A/B)
  if (d->dtype == DT_F16) { if (g_fp8_mx) hipLaunchKernelGGL((gemm_fp8_kernel<_Float16, true>), dim3(tiles), dim3(256), 0, st, a);
                            else hipLaunchKernelGGL((gemm_fp8_kernel<_Float16, false>), dim3(tiles), dim3(256), 0, st, a); }
  else if (d->dtype == DT_BF16) { if (g_fp8_mx) hipLaunchKernelGGL((gemm_fp8_kernel<__bf16, true>), dim3(tiles), dim3(256), 0, st, a);
                                  else hipLaunchKernelGGL((gemm_fp8_kernel<__bf16, false>), dim3(tiles), dim3(256), 0, st, a); }
  else return -22;
  HALLO_CHECK_LAUNCH();
  return 0;
}
