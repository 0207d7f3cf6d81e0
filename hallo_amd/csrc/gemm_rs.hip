// Row-stationary GEMM for the short-K projections of the denoising UNet (K = 320 / 640): the A rows of a workgroup live in
// REGISTERS for the whole kernel and the weight matrix streams past them.
//
// Replaces, for these shapes, the 128x128x64 LDS-tiled kernel of gemm.hip on the calls
//   to_q|to_k|to_v with norm1 folded in  (hallo/models/mutual_self_attention.py:253-303, attention.py:828-884 via diffusers
//                                         Attention; motion_module.py:553-609),
//   FeedForward net.0 (GEGLU) with norm3 / ff_norm folded in  (hallo/models/attention.py:601,905; motion_module.py:420).
//
// Why: with K = 320 a 128 x 128 output tile has only 5 K steps between a prologue and an epilogue, every A row block is
// re-fetched by each of the 7.5-20 N tiles of its row (PMC r1: 1.42x the algorithmic read traffic), and a separate
// hallo_row_stats launch reads A once more for the LayerNorm statistics.  Those GEMMs ran at 2.2-2.7 TB/s / 500-680 TFLOP/s.
// Here
//   * a wave owns RB x 32 rows; their K/16 MFMA B-operand fragments (16 bytes per lane each, read straight from global
//     memory: lane = row, 8 consecutive k) stay in 160 VGPRs -- A is read from HBM exactly once and never touches LDS;
//   * the LayerNorm statistics are computed from those registers (packed dot products; mean / rstd per lane = per output
//     row), so no statistics pass and no normalised copy exist at all;
//   * W streams through an 8-slot LDS ring of [128 rows][64 k] chunks by LDS-DMA (XOR-swizzled through the source address,
//     conflict-free ds_read_b128), continuous across the N tiles of the workgroup, 7 chunks in flight behind a counted
//     vmcnt and ONE barrier per chunk (32 MFMAs per wave per barrier at K = 320);
//   * swapped operands (D = W . A^T): a lane owns an output row, the epilogue scalars of the row (mean, rstd) are lane-local,
//     bias / column sums come from LDS, two 8-byte groups are merged with one v_permlane32_swap into 16-byte stores.
// One workgroup (4 waves, one per SIMD, up to 512 registers each) per CU: the A fragments (160 VGPRs) and a 4-block-wide
// accumulator set do not fit the 256-register budget of two waves per SIMD without spilling into the K loop, and with the
// whole register file a wave runs 32 MFMAs between barriers, so the deep DMA ring -- not a second wave -- hides the latency.
// Grid: (M / rows per workgroup) x nsplit; the N tiles of a row block are split over `nsplit` workgroups only when there
// are fewer row blocks than CUs (their A rows are then re-read from L2: neighbours in the XCD-contiguous order).
#include "gemm_args.h"

namespace hallo {

namespace {
constexpr int RS_CHUNK_BYTES = 128 * 64 * 2;      // one W chunk: 128 rows x 64 k
constexpr int RS_RING = 8;
constexpr int RS_MAX_WROWS = 2560;                // rows of W (GEGLU: 2 N) whose bias / column sums fit the LDS constants
constexpr int RS_CONST_FLOATS = RS_MAX_WROWS + 128;   // + one tile of slack: a partial last tile indexes past N
constexpr int RS_OFF_BIAS = RS_RING * RS_CHUNK_BYTES;                    // fp32 [RS_CONST_FLOATS]
constexpr int RS_OFF_CSUM = RS_OFF_BIAS + RS_CONST_FLOATS * 4;           // fp32 [RS_CONST_FLOATS]
constexpr int RS_LDS = RS_OFF_CSUM + RS_CONST_FLOATS * 4;                // 152576 B: one workgroup per CU
}  // namespace

// K16 = K / 16 (20 or 40), RB = 32-row blocks per wave (K16 * RB = 40: 160 fragment VGPRs)
template <typename T, int K16, int RB, bool GEGLU, bool LNF>
__global__ __launch_bounds__(256, 1) void gemm_rs_kernel(const GemmArgs p, const int nsplit) {
  using V8 = typename Vec<T>::v8;
  constexpr int KS = K16 / 4;                     // 64-wide chunks per N tile
  constexpr int WG_ROWS = 4 * RB * 32;
  constexpr int NB = 4;                           // 32-row blocks of W per chunk
  constexpr int TILE_COLS = GEGLU ? 64 : 128;     // output columns per N tile (GEGLU: 64 value rows + 64 gate rows of W)
  constexpr int DMA_PER_CHUNK = 4;                // LDS-DMA instructions per wave per chunk

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* const lds = (lds_u8*)smem;
  float* const sBias = reinterpret_cast<float*>(smem + RS_OFF_BIAS);
  float* const sCsum = reinterpret_cast<float*>(smem + RS_OFF_CSUM);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, l31 = lane & 31;

  const int nwg = gridDim.x;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int rowblk = bid / nsplit, part = bid - rowblk * nsplit;
  const int m0 = rowblk * WG_ROWS + wave * (RB * 32);
  const int ntiles = (p.N + TILE_COLS - 1) / TILE_COLS;
  const int t_begin = (int)((long)ntiles * part / nsplit), t_end = (int)((long)ntiles * (part + 1) / nsplit);
  const int nchunks = (t_end - t_begin) * KS;

  const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(p.B);
  T* __restrict__ C = reinterpret_cast<T*>(p.C);
  const int wrows = GEGLU ? 2 * p.N : p.N;

  // ---- W stream: per-lane source offsets of this wave's four DMA instructions per chunk (pieces w*256 + j*64 + lane).
  // The N-tile advance is part of the per-lane offset so that rows past the end of W (partial last tile) fall outside the
  // descriptor and are written as zeros; the K advance is the scalar offset. ----
  const long w_bytes = (((long)wrows - 1) * p.ldb + p.K) * 2;
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(W), 0, (int)(w_bytes > 0xFFFFFFFFL ? 0xFFFFFFFFL : w_bytes), 0x00020000);
  unsigned w_voff[DMA_PER_CHUNK];
#pragma unroll
  for (int j = 0; j < DMA_PER_CHUNK; ++j) {
    const int row = wave * 32 + j * 8 + (lane >> 3);               // row of the chunk (0..127)
    const int logical = (lane & 7) ^ ((row >> 1) & 7);             // LDS is lane-linear: the swizzle goes on the SOURCE
    const long wrow = GEGLU ? (long)(row & 63) + (row >= 64 ? p.N : 0) : row;
    w_voff[j] = (unsigned)((wrow * p.ldb + logical * 8) * 2);
  }
  const unsigned tile_step = (unsigned)((long)TILE_COLS * p.ldb * 2);
  auto issue_chunk = [&](int c) {      // chunk c of this workgroup's stream: tile t_begin + c / KS, k chunk c % KS
    const int t = t_begin + c / KS, kc = c - (c / KS) * KS;
    const unsigned toff = (unsigned)t * tile_step;
    const int dst = (c & (RS_RING - 1)) * RS_CHUNK_BYTES + wave_u * 4096;
#pragma unroll
    for (int j = 0; j < DMA_PER_CHUNK; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(lds + dst + j * 1024), 16,
                                               (int)(w_voff[j] + toff), kc * 128, 0, 0);
  };
#pragma unroll
  for (int c0 = 0; c0 < RS_RING - 1; ++c0)
    if (c0 < nchunks) issue_chunk(c0);

  // ---- epilogue constants of all W rows -> LDS (fp32): bias (+ the per-row-group bias2 row of this workgroup), column sums ----
  {
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* bias2 = reinterpret_cast<const T*>(p.bias2);
    const long b2row = bias2 ? (long)((rowblk * WG_ROWS) / p.bias2_rpg) * p.bias2_ld : 0;
    for (int n = tid; n < RS_CONST_FLOATS; n += 256) {
      float b = 0.0f, g = 0.0f;
      if (n < wrows) {
        if (bias) b = to_f32(bias[n]);
        if (bias2) b += to_f32(bias2[b2row + n]);
        if (LNF) g = p.ln_colsum[n];
      }
      sBias[n] = b;
      if (LNF) sCsum[n] = g;
    }
  }

  // ---- A fragments: lane holds A[m0 + rb*32 + l31][k16*16 + hi*8 .. +8] for every k16 ----
  V8 af[RB][K16];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int m = min(m0 + rb * 32 + l31, p.M - 1);
    const T* arow = A + (long)m * p.lda + hi * 8;
#pragma unroll
    for (int k = 0; k < K16; ++k) af[rb][k] = ld8<T>(arow + k * 16);
  }
  // LayerNorm statistics of the lane's rows from the fragments (one pass: sum and sum of squares by packed dot products)
  float r_mean[RB], r_rstd[RB];
  if (LNF) {
    typedef __attribute__((ext_vector_type(2))) T V2t;
    const V2t one2 = {from_f32<T>(1.0f), from_f32<T>(1.0f)};
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      float sm = 0.0f, sq = 0.0f;
#pragma unroll
      for (int k = 0; k < K16; ++k)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const V2t x2 = {af[rb][k][e], af[rb][k][e + 1]};
          sm = dot2(x2, one2, sm);
          sq = dot2(x2, x2, sq);
        }
      sm += __shfl_xor(sm, 32, 64);                 // the partner lane holds the other half of every 16-wide k group
      sq += __shfl_xor(sq, 32, 64);
      const float mean = sm / (float)p.K;
      r_mean[rb] = mean;
      r_rstd[rb] = rsqrtf(fmaxf(sq / (float)p.K - mean * mean, 0.0f) + p.ln_eps);
    }
  }
  __syncthreads();      // constants visible (this also drains the chunks in flight once; the counted waits below stay valid)

  // W fragment (A operand): chunk row nb*32 + l31, logical 16-byte chunk k16*2 + hi -> physical ^ ((row >> 1) & 7)
  const int xsw = hi ^ ((l31 >> 1) & 7);
  const lds_u8* const wb = lds + l31 * 128;

  const float lead = p.lead_alpha, alpha = p.alpha;
  int c = 0;
  for (int t = t_begin; t < t_end; ++t) {
    f32x16 acc[RB][NB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][nb][r] = 0.0f;
#pragma unroll
    for (int kc = 0; kc < KS; ++kc, ++c) {
      // chunk c has landed for this wave's own pieces when at most the RING-2 younger chunks are outstanding; loads retire
      // in order, so the count is sufficient whatever the stores of the previous epilogue do
      const int younger = min(nchunks - 1 - c, RS_RING - 2);
      if (younger >= RS_RING - 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (younger > 0) {      // stream tail: wait for everything but the last chunk (or, at the end, for everything)
        if (younger >= 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();              // every wave's pieces of chunk c landed; slot (c-1) % RING is free
      if (c + RS_RING - 1 < nchunks) issue_chunk(c + RS_RING - 1);
      const lds_u8* const sw = wb + (c & (RS_RING - 1)) * RS_CHUNK_BYTES;
#pragma unroll
      for (int k16 = 0; k16 < 4; ++k16) {
        const int co = ((k16 * 2) ^ xsw) * 16;
        typedef const __attribute__((address_space(3))) V8* ldsv8;
        V8 wf[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wf[nb] = *(ldsv8)(sw + nb * 32 * 128 + co);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) acc[rb][nb] = Vec<T>::mfma32(wf[nb], af[rb][kc * 4 + k16], acc[rb][nb]);
      }
    }
    // ---- epilogue of N tile t: lane owns row m, D rows n = nb*32 + 8g + 4hi + 0..3 ----
    const int n0 = t * TILE_COLS;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int m = m0 + rb * 32 + l31;
      T* crow = C + (long)min(m, p.M - 1) * p.ldc;
      const float mean = LNF ? r_mean[rb] : 0.0f, rstd = LNF ? r_rstd[rb] : 1.0f;
#pragma unroll
      for (int ob = 0; ob < (GEGLU ? 2 : NB); ++ob) {
        const float colscale = alpha * ((!GEGLU && n0 + ob * 32 < p.lead_cols) ? lead : 1.0f);
        unsigned pk[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nn = n0 + ob * 32 + 8 * g + 4 * hi;
          typename Vec<T>::v4 o;
          if (GEGLU) {                                           // value row nn, gate row N + nn of W
            const f32x4 bh = *reinterpret_cast<const f32x4*>(sBias + nn), bg = *reinterpret_cast<const f32x4*>(sBias + p.N + nn);
            f32x4 gh = {0, 0, 0, 0}, gg = {0, 0, 0, 0};
            if (LNF) { gh = *reinterpret_cast<const f32x4*>(sCsum + nn); gg = *reinterpret_cast<const f32x4*>(sCsum + p.N + nn); }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float hv = acc[rb][ob][g * 4 + j], gv = acc[rb][2 + ob][g * 4 + j];
              if (LNF) { hv = rstd * (hv - mean * gh[j]); gv = rstd * (gv - mean * gg[j]); }
              o[j] = from_f32<T>((hv + bh[j]) * gelu_erf_f(gv + bg[j]));
            }
          } else {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(sBias + nn);
            f32x4 gc = {0, 0, 0, 0};
            if (LNF) gc = *reinterpret_cast<const f32x4*>(sCsum + nn);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float v = acc[rb][ob][g * 4 + j];
              if (LNF) v = rstd * (v - mean * gc[j]);
              o[j] = from_f32<T>((v + bb[j]) * colscale);
            }
          }
          const uint2 u = __builtin_bit_cast(uint2, o);
          pk[g][0] = u.x; pk[g][1] = u.y;
        }
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          // groups g (cols 8g + 4hi ..) and g+1: after the half swap lanes hi = 0 hold cols 8g .. 8g+7, lanes hi = 1 cols 8g+8 .. 8g+15
          const auto x = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
          const auto y = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
          const int nc = n0 + ob * 32 + 8 * g + 8 * hi;
          if (m < p.M && nc < p.N) *reinterpret_cast<uint4*>(crow + nc) = make_uint4(x[0], y[0], x[1], y[1]);
        }
      }
    }
  }
}

// Eligibility of a problem for the row-stationary kernel (see the header): everything else stays on gemm.hip / gemm3.hip.
bool gemm_rs_eligible(const GemmArgs& a, bool conv, bool geglu, int batch) {
  if (conv || batch != 1 || a.splits > 1) return false;
  if (a.K != 320 && a.K != 640) return false;
  if (a.residual || a.rowscale || a.bias_per_row || a.out_f32 || a.act != ACT_NONE) return false;
  if (a.ln_colsum && a.ln_stats) return false;                 // the caller already paid for a statistics pass
  const int wrows = geglu ? 2 * a.N : a.N;
  if ((a.N & 7) || wrows > RS_MAX_WROWS) return false;
  if (a.lead_cols % 32) return false;
  const int wg_rows = (a.K == 320) ? 256 : 128;
  if (a.bias2 && (a.bias2_rpg % wg_rows)) return false;        // the bias2 row must be constant per workgroup
  if (a.M < 32 * wg_rows) return false;                        // too few row blocks to fill the chip: the tiled kernels do better
  if ((a.lda & 7) || (a.ldb & 7) || (a.ldc & 7) || (reinterpret_cast<uintptr_t>(a.C) & 15) || (reinterpret_cast<uintptr_t>(a.A) & 15)) return false;
  if (a.ln_colsum && (reinterpret_cast<uintptr_t>(a.ln_colsum) & 15)) return false;
  return true;
}

template <typename T, int K16, int RB, bool G, bool L>
static void launch_rs_one(const GemmArgs& a, int nsplit, dim3 grid, hipStream_t st) {
  static bool attr_done = false;       // the kernel needs more than the default 64 KB of dynamic LDS
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rs_kernel<T, K16, RB, G, L>), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_rs_kernel<T, K16, RB, G, L>), grid, dim3(256), RS_LDS, st, a, nsplit);
}

template <typename T>
int launch_gemm_rs(const GemmArgs& a, bool geglu, hipStream_t st) {
  const bool lnf = a.ln_colsum != nullptr;
  const int wg_rows = (a.K == 320) ? 256 : 128;
  const int rowblks = (a.M + wg_rows - 1) / wg_rows;
  const int tile_cols = geglu ? 64 : 128;
  const int ntiles = (a.N + tile_cols - 1) / tile_cols;
  // N split: one workgroup per CU (256 slots).  Pick the split with the fewest "rounds x work per workgroup", where a
  // workgroup costs its A load (~2 N tiles' worth) plus its tiles.
  int best = 1;
  float best_cost = 3.0e38f;
  for (int s = 1; s <= 8 && s <= ntiles; ++s) {
    const int wgs = rowblks * s;
    const float rounds = (float)((wgs + 255) / 256);
    const float cost = rounds * (2.0f + (float)((ntiles + s - 1) / s));
    if (cost < best_cost - 1e-3f) { best_cost = cost; best = s; }
  }
  dim3 grid(rowblks * best);
#define HALLO_RS(K16, RB, G, L) launch_rs_one<T, K16, RB, G, L>(a, best, grid, st)
  if (a.K == 320) {
    if (geglu) { if (lnf) HALLO_RS(20, 2, true, true); else HALLO_RS(20, 2, true, false); }
    else { if (lnf) HALLO_RS(20, 2, false, true); else HALLO_RS(20, 2, false, false); }
  } else {
    if (geglu) { if (lnf) HALLO_RS(40, 1, true, true); else HALLO_RS(40, 1, true, false); }
    else { if (lnf) HALLO_RS(40, 1, false, true); else HALLO_RS(40, 1, false, false); }
  }
#undef HALLO_RS
  HALLO_CHECK_LAUNCH();
  return 0;
}

template int launch_gemm_rs<_Float16>(const GemmArgs&, bool, hipStream_t);
template int launch_gemm_rs<__bf16>(const GemmArgs&, bool, hipStream_t);

}  // namespace hallo
