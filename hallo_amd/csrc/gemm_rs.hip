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
//   * a wave owns 32 rows; their K/16 MFMA B-operand fragments (16 bytes per lane each, read straight from global
//     memory: lane = row, 8 consecutive k) stay in 80 / 160 VGPRs -- A is read from HBM exactly once and never touches LDS;
//   * the LayerNorm statistics are computed from those registers (packed dot products; mean / rstd per lane = per output
//     row), so no statistics pass and no normalised copy exist at all;
//   * W streams through a 4-slot LDS ring of [128 | 64 rows][64 k] chunks by LDS-DMA (XOR-swizzled through the source
//     address, conflict-free ds_read_b128), continuous across the N tiles of the workgroup, 3 chunks in flight behind a
//     counted vmcnt and ONE barrier per chunk (16 / 8 MFMAs per wave per barrier);
//   * swapped operands (D = W . A^T): a lane owns an output row, the epilogue scalars of the row (mean, rstd) are lane-local,
//     bias / column sums come from LDS, two 8-byte groups are merged with one v_permlane32_swap into 16-byte stores.
// (Round-2 correction, tools/cbench `rsdbg` ablations: what held THIS kernel was its epilogue -- q|k|v 74 us full / 43 without
// epilogue / 35 MFMA loop alone, GEGLU 180 / 111 / 86 -- not the fabric as the paragraph below concluded.  gemm_rs2.hip, which
// takes the K = 320 problems now, turns the loops round so that the epilogue interleaves with the next pair's MFMAs; this
// kernel remains for K = 640, where 160 registers of A fragments leave no room for a second accumulator set.)
// Every workgroup streams the WHOLE of W past its rows, so the L2 -> LDS fabric carries (M / rows per workgroup) x |W| bytes.  Measured: a 128-row workgroup (two per CU) ran
// the 65536 x 960 x 320 K loop in 46-49 us = 314 MB at 6.8 TB/s from L2 whatever the ring depth, chunk size or fragment-read
// schedule (MFMA floor 20 us); the tiled 128 x 128 kernel moves 614 MB for the same problem.  So the workgroup is made as
// tall as the register file allows: 8 waves x 32 rows = 256 rows share ONE W stream (157 MB), one workgroup per CU.  The two
// waves of a SIMD are the two HALVES of the workgroup (waves 0-3 / 4-7); the second half runs LAG = 2 chunks behind the
// first, so that one half's VALU epilogue (LayerNorm affine, GELU, pack, store) overlaps the other half's MFMAs.
#include "gemm_args.h"

namespace hallo {

#if defined(__HIP_DEVICE_COMPILE__)
#define RS_KEEP(x) asm volatile("" :: "v"(x))
#else
#define RS_KEEP(x) (void)(x)
#endif

namespace {
constexpr int RS_MAX_WROWS = 2560;                // rows of W (GEGLU: 2 N) whose bias / column sums fit the LDS constants
constexpr int RS_CONST_N = RS_MAX_WROWS + 128;    // + one tile of slack: a partial last tile indexes past N
constexpr int RS_RING = 8, RS_LAG = 2;
constexpr int RS_RING_BYTES = RS_RING * 128 * 64 * 2;   // 8 chunks of 128 rows x 64 k (K = 320); K = 640 uses 64-row chunks
constexpr int RS_OFF_BIAS = RS_RING_BYTES;                               // fp32 [RS_CONST_N]: bias (+ this workgroup's bias2 row)
constexpr int RS_OFF_CSUM = RS_OFF_BIAS + RS_CONST_N * 4;                // fp32 [RS_CONST_N]: LayerNorm column sums
constexpr int RS_LDS = RS_OFF_CSUM + RS_CONST_N * 4;                     // 152576 B
static_assert(RS_LDS <= 160 * 1024, "one workgroup per CU");
}  // namespace

// K16 = K / 16 (20 or 40); NB = 32-row blocks of W per chunk (4 at K = 320, 2 at K = 640)
template <typename T, int K16, int NB, bool GEGLU, bool LNF>
__global__ __launch_bounds__(512, 2) void gemm_rs_kernel(const GemmArgs p, const int dbg) {
  using V8 = typename Vec<T>::v8;
  constexpr int KS = K16 / 4;                     // 64-wide chunks per N tile
  constexpr int WG_ROWS = 256;                    // 8 waves x 32 rows
  constexpr int RING = RS_RING, LAG = RS_LAG;
  constexpr int CHUNK_ROWS = NB * 32;
  constexpr int CHUNK_BYTES = CHUNK_ROWS * 128;
  constexpr int TILE_COLS = GEGLU ? CHUNK_ROWS / 2 : CHUNK_ROWS;   // output columns per N tile (GEGLU: value rows + gate rows of W)
  constexpr int DPC = NB / 2;                     // LDS-DMA instructions per wave per chunk (8 waves)
  static_assert(RING * CHUNK_BYTES <= RS_RING_BYTES, "ring");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* const lds = (lds_u8*)smem;
  float* const sBias = reinterpret_cast<float*>(smem + RS_OFF_BIAS);
  float* const sCsum = reinterpret_cast<float*>(smem + RS_OFF_CSUM);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int half_u = wave_u >> 2;                 // 0: waves 0-3, 1: waves 4-7 (LAG chunks behind)
  const int hi = lane >> 5, l31 = lane & 31;

  const int rowblk = blockIdx.x;
  const int m0 = rowblk * WG_ROWS + wave * 32;
  const int ntiles = (p.N + TILE_COLS - 1) / TILE_COLS;
  const int nchunks = ntiles * KS;

  const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(p.B);
  T* __restrict__ C = reinterpret_cast<T*>(p.C);
  const int wrows = GEGLU ? 2 * p.N : p.N;

  // ---- W stream: per-lane source offsets of this wave's DMA instructions per chunk (pieces w*CHUNK_ROWS*2 + j*64 + lane).
  // The N-tile advance is part of the per-lane offset so that rows past the end of W (partial last tile) fall outside the
  // descriptor and are written as zeros; the K advance is the scalar offset. ----
  const long w_bytes = (((long)wrows - 1) * p.ldb + p.K) * 2;
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(W), 0, (int)(w_bytes > 0xFFFFFFFFL ? 0xFFFFFFFFL : w_bytes), 0x00020000);
  unsigned w_voff[DPC];
#pragma unroll
  for (int j = 0; j < DPC; ++j) {
    const int row = wave * (CHUNK_ROWS / 8) + j * 8 + (lane >> 3); // row of the chunk
    const int logical = (lane & 7) ^ ((row >> 1) & 7);             // LDS is lane-linear: the swizzle goes on the SOURCE
    const long wrow = GEGLU ? (long)(row % (CHUNK_ROWS / 2)) + (row >= CHUNK_ROWS / 2 ? p.N : 0) : row;
    w_voff[j] = (unsigned)((wrow * p.ldb + logical * 8) * 2);
  }
  const unsigned tile_step = (unsigned)((long)TILE_COLS * p.ldb * 2);
  // issue state: chunk index -> (tile offset, k chunk, ring slot), advanced incrementally (no divisions in the loop)
  unsigned is_toff = 0;
  int is_kc = 0, is_slot = 0, issued = 0;
  auto issue_next = [&]() {
    const int dst = is_slot * CHUNK_BYTES + wave_u * (CHUNK_BYTES / 8);
#pragma unroll
    for (int j = 0; j < DPC; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(lds + dst + j * 1024), 16,
                                               (int)(w_voff[j] + is_toff), is_kc * 128, 0, 0);
    ++issued;
    if (++is_kc == KS) { is_kc = 0; is_toff += tile_step; }
    if (++is_slot == RING) is_slot = 0;
  };
#pragma unroll
  for (int c0 = 0; c0 < RING - 1 - LAG; ++c0)
    if (issued < nchunks) issue_next();

  // ---- epilogue constants of all W rows -> LDS (fp32): bias (+ the per-row-group bias2 row of this workgroup), column sums ----
  {
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* bias2 = reinterpret_cast<const T*>(p.bias2);
    const long b2row = bias2 ? (long)((rowblk * WG_ROWS) / p.bias2_rpg) * p.bias2_ld : 0;
    for (int n = tid; n < RS_CONST_N; n += 512) {
      float b = 0.0f, g = 0.0f;
      if (n < wrows) {
        if (bias) b = to_f32(bias[n]);
        if (bias2) b += to_f32(bias2[b2row + n]);
        if (LNF) g = p.ln_colsum[n];
        if (!GEGLU) b *= p.alpha * (n < p.lead_cols ? p.lead_alpha : 1.0f);     // the column scale is folded into the constants
      }
      sBias[n] = b;
      if (LNF) sCsum[n] = g;
    }
  }

  // ---- A fragments: lane holds A[m0 + l31][k16*16 + hi*8 .. +8] for every k16 ----
  V8 af[K16];
  {
    const int m = min(m0 + l31, p.M - 1);
    const T* arow = A + (long)m * p.lda + hi * 8;
#pragma unroll
    for (int k = 0; k < K16; ++k) af[k] = ld8<T>(arow + k * 16);
  }
  // LayerNorm statistics of the lane's row from the fragments (one pass: sum and sum of squares by packed dot products)
  float mean = 0.0f, rstd = 1.0f;
  if (LNF) {
    typedef __attribute__((ext_vector_type(2))) T V2t;
    const V2t one2 = {from_f32<T>(1.0f), from_f32<T>(1.0f)};
    float sm = 0.0f, sq = 0.0f;
#pragma unroll
    for (int k = 0; k < K16; ++k)
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const V2t x2 = {af[k][e], af[k][e + 1]};
        sm = dot2(x2, one2, sm);
        sq = dot2(x2, x2, sq);
      }
    sm += __shfl_xor(sm, 32, 64);                 // the partner lane holds the other half of every 16-wide k group
    sq += __shfl_xor(sq, 32, 64);
    mean = sm / (float)p.K;
    rstd = rsqrtf(fmaxf(sq / (float)p.K - mean * mean, 0.0f) + p.ln_eps);
  }
  __syncthreads();      // constants visible (this also drains the chunks in flight once; the counted waits below stay valid)

  // W fragment (A operand): chunk row nb*32 + l31, logical 16-byte chunk k16*2 + hi -> physical ^ ((row >> 1) & 7)
  const int xsw = hi ^ ((l31 >> 1) & 7);
  const lds_u8* const wb = lds + l31 * 128;
  const int m = m0 + l31;
  T* const crow = C + (long)min(m, p.M - 1) * p.ldc;
  const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  const float lead = p.lead_alpha, alpha = p.alpha;

  // One synchronisation step = one chunk of the first half: wait until this wave's pieces of chunk `step` have landed (at
  // most RING-2-LAG younger chunks outstanding; loads retire in order, so the count is sufficient whatever the stores of an
  // epilogue do), barrier, re-fill the slot the lagging half released.  Every wave runs nchunks + LAG steps.
  int step = 0;
  auto sync_step = [&]() {
    if (dbg & 4) { ++step; return; }              // ablation: no waits, no barrier, no DMA (stale LDS data)
    if (step < nchunks) {
      const int younger = issued - 1 - step;      // 0 .. RING-2-LAG
      if (younger >= 4) { if (DPC == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      else if (younger == 3) { if (DPC == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
      else if (younger == 2) { if (DPC == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
      else if (younger == 1) { if (DPC == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // chunk `step` landed for everybody; both halves are done with chunk step-1-LAG
    if (issued < nchunks) issue_next();
    ++step;
  };
  if (half_u == 1)
    for (int i = 0; i < LAG; ++i) sync_step();

  int slot = 0;
  V8 wa[NB], wc[NB];
  for (int t = 0; t < ntiles; ++t) {
    f32x16 acc[NB];          // initialised by the first MFMA of the tile (C = 0)
#pragma unroll
    for (int kc = 0; kc < KS; ++kc) {
      sync_step();
      const lds_u8* const sw = wb + slot * CHUNK_BYTES;
      slot = (slot + 1) & (RING - 1);
      // W fragments two k16 steps ahead of the MFMAs that consume them, in two named register sets (hipcc otherwise reuses one
      // 8-register pair and exposes the LDS latency before every second MFMA)
      typedef const __attribute__((address_space(3))) V8* ldsv8;
#define RS_LOADW(dst, k16) _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) dst[nb] = *(ldsv8)(sw + nb * 32 * 128 + (((k16) * 2) ^ xsw) * 16)
#define RS_MFMA(src, k16) _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) \
        acc[nb] = Vec<T>::mfma32(src[nb], af[kc * 4 + (k16)], (kc == 0 && (k16) == 0) ? zero16 : acc[nb])
      const bool ld = !(dbg & 8) || (t == 0 && kc == 0);      // (ablation 8: W fragments read once, then reused)
      if (ld) { RS_LOADW(wa, 0); }
      if (ld) { RS_LOADW(wc, 1); }
      __builtin_amdgcn_sched_barrier(0);
      RS_MFMA(wa, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ld) { RS_LOADW(wa, 2); }
      __builtin_amdgcn_sched_barrier(0);
      RS_MFMA(wc, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (ld) { RS_LOADW(wc, 3); }
      __builtin_amdgcn_sched_barrier(0);
      RS_MFMA(wa, 2);
      RS_MFMA(wc, 3);
#undef RS_LOADW
#undef RS_MFMA
    }
    // ---- epilogue of N tile t: lane owns row m, D rows n = nb*32 + 8g + 4hi + 0..3 ----
    const int n0 = t * TILE_COLS;
    if (dbg & 2) {      // ablation: no epilogue at all (accumulators kept alive)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) RS_KEEP(acc[nb]);
      continue;
    }
#pragma unroll
    for (int ob = 0; ob < (GEGLU ? NB / 2 : NB); ++ob) {
      // out = cs * (rstd * (acc - mean * G[n]) + bias[n]) = rs * acc + (c1 * G[n] + cs * bias[n]),  rs = rstd * cs, c1 = -mean * rs:
      // two FMAs per element (cs * bias is what the LDS constants hold)
      const float cs = GEGLU ? 1.0f : alpha * ((n0 + ob * 32 < p.lead_cols) ? lead : 1.0f);
      const float rs = rstd * cs, c1 = -mean * rs;
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
            const float hv = LNF ? __builtin_fmaf(rs, acc[ob][g * 4 + j], __builtin_fmaf(c1, gh[j], bh[j])) : acc[ob][g * 4 + j] + bh[j];
            const float gv = LNF ? __builtin_fmaf(rs, acc[NB / 2 + ob][g * 4 + j], __builtin_fmaf(c1, gg[j], bg[j]))
                                 : acc[NB / 2 + ob][g * 4 + j] + bg[j];
            o[j] = from_f32<T>(hv * gelu_erf_f(gv));
          }
        } else {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(sBias + nn);
          f32x4 gc = {0, 0, 0, 0};
          if (LNF) gc = *reinterpret_cast<const f32x4*>(sCsum + nn);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            o[j] = from_f32<T>(__builtin_fmaf(rs, acc[ob][g * 4 + j], LNF ? __builtin_fmaf(c1, gc[j], bb[j]) : bb[j]));
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
        if (dbg & 1) { RS_KEEP(x); RS_KEEP(y); continue; }      // ablation: no stores
        if (m < p.M && nc < p.N) *reinterpret_cast<uint4*>(crow + nc) = make_uint4(x[0], y[0], x[1], y[1]);
      }
    }
  }
  if (half_u == 0)
    for (int i = 0; i < LAG; ++i) sync_step();
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
  if (a.bias2 && (a.bias2_rpg % 256)) return false;            // the bias2 row must be constant per workgroup
  // 256-row workgroups, one per CU: the grid runs in rounds of 256; require the last round to be >= 85 % full (73728 rows =
  // 288 workgroups would idle 44 % of the chip in their second round) and enough rows to fill the chip at all
  const int wgs = (a.M + 255) / 256;
  const int rounds = (wgs + 255) / 256;
  if (wgs < 192 || wgs * 100 < rounds * 256 * 85) return false;
  if ((a.lda & 7) || (a.ldb & 7) || (a.ldc & 7) || (reinterpret_cast<uintptr_t>(a.C) & 15) || (reinterpret_cast<uintptr_t>(a.A) & 15)) return false;
  if (a.ln_colsum && (reinterpret_cast<uintptr_t>(a.ln_colsum) & 15)) return false;
  return true;
}

static int g_rs_dbg = 0;      // hallo_set_option("gemm_rs_dbg", bits): timing ablations (1: no stores, 2: no epilogue) -- wrong results
void set_gemm_rs_dbg(int v) { g_rs_dbg = v; }

template <typename T, int K16, int NB, bool G, bool L>
static int launch_rs_one(const GemmArgs& a, dim3 grid, hipStream_t st) {
  static bool attr_done[64] = {};    // per device: the opt-in to > 64 KB of dynamic LDS belongs to the device's function
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -19;
  if (!attr_done[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rs_kernel<T, K16, NB, G, L>), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS) != hipSuccess)
      return -12;
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL((gemm_rs_kernel<T, K16, NB, G, L>), grid, dim3(512), RS_LDS, st, a, g_rs_dbg);
  return 0;
}

template <typename T>
int launch_gemm_rs(const GemmArgs& a, bool geglu, hipStream_t st) {
  const bool lnf = a.ln_colsum != nullptr;
  dim3 grid((a.M + 255) / 256);
  int rc = 0;
#define HALLO_RS(K16, NB, G, L) rc = launch_rs_one<T, K16, NB, G, L>(a, grid, st)
  if (a.K == 320) {
    if (geglu) { if (lnf) HALLO_RS(20, 4, true, true); else HALLO_RS(20, 4, true, false); }
    else { if (lnf) HALLO_RS(20, 4, false, true); else HALLO_RS(20, 4, false, false); }
  } else {
    if (geglu) { if (lnf) HALLO_RS(40, 2, true, true); else HALLO_RS(40, 2, true, false); }
    else { if (lnf) HALLO_RS(40, 2, false, true); else HALLO_RS(40, 2, false, false); }
  }
#undef HALLO_RS
  if (rc) return rc;
  HALLO_CHECK_LAUNCH();
  return 0;
}

template int launch_gemm_rs<_Float16>(const GemmArgs&, bool, hipStream_t);
template int launch_gemm_rs<__bf16>(const GemmArgs&, bool, hipStream_t);

}  // namespace hallo
