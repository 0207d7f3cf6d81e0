// Exact-fit MFMA GEMM for the mid-size problems of the 32x32, 16x16 and 8x8 latent levels (gfx950).
//
// Same reference calls as gemm.hip (diffusers Attention.to_out / to_q|k|v at C = 640 / 1280: hallo/models/attention.py:22-23,
// mutual_self_attention.py:253-303; FeedForward net[2]: attention.py:601,905, motion_module.py:420; Transformer3DModel
// proj_in / proj_out: transformer_3d.py:199,242) -- C[M,N] = epilogue(A[M,K] . W[N,K]^T) with M = 1024 ... 18432 rows.
//
// Why a fourth GEMM kernel (round 4; profiles/r4_gemm2_pmc_*.txt, profiles/r4_vendor_ab.json, profiles/r4_gemm4_stamps.txt).
// At these sizes the 128x128 kernel launches 80-480 workgroups on 256 CUs, each with ONE 32 KB K step in flight: PMC of
// 4096 x 1280 x 1280 shows the waves parked 61 % of the time and the matrix pipe 24 % busy -- the K loop is a chain of L2
// round trips, and what the chip gets is 25-35 us for 13 GFLOP (hipBLASLt: 29 us, it quantises the same way).  The big tile
// of gemm3.hip has 64-128 tiles for these outputs, so it splits K, and the fp32 slabs (sp x M x N x 4 bytes written and
// re-read) plus the separate reduction launch cost a third of the time (88 us for 4096 x 1280 x 5120: gemm 59 + reduce 29).
//
// Design.
//   * Tile 128 x 160 x 64.  Every width of the UNets is a multiple of 160 and the token counts are multiples of 128, so a
//     4096 x 1280 output is EXACTLY 256 tiles = one per CU: no wave quantisation, no split-K, no slab.
//   * One persistent workgroup per CU, 8 waves in two ROLES: waves 0-3 multiply (2 x 2, a wave owns 64 x 80 outputs = 4 x 5
//     blocks of v_mfma_f32_16x16x32: 80 accumulator registers), waves 4-7 only issue the LDS-DMA of the operand stream, one
//     per SIMD next to a multiplying wave.  (Measured with s_memtime stamps: with the four multiplying waves issuing their own
//     DMA -- an LDS-DMA instruction holds its wave's issue slot for 60-100 cycles -- a K step took 1280-1800 cycles against
//     640 cycles of MFMAs; with the roles split it takes 1000-1020, which is the LDS: 72 KB of fragment reads + 36 KB of DMA
//     writes per step.  A 64 x 80 wave tile cannot do better; the larger wave tiles that could do not give 256 tiles.)
//   * The workgroup walks a list of segments: `dp` rounds of whole tiles (tile = xcd_remap(round * G + workgroup)), then one
//     TAIL segment for the remaining R < G tiles: either whole tiles again (parts = 1: a partly filled round), or, when the K
//     loop is long enough to pay for it, each tail tile's K range split over `parts` workgroups.  Partial sums are written as
//     fp32 register images with write-through (sc1) 16-byte stores, every wave drains its stores, one lane takes a ticket, and
//     the LAST arriver of the tile (no workgroup ever waits for another) reads the parts back with sc1 loads in K order --
//     fixed order, so the result is bit-reproducible -- and runs the epilogue.  The per-tile counters live in the caller's
//     zero-initialised workspace and are restored to zero by the reducer.
//   * 4-slot LDS ring of K steps (36 KB each: A [128][64] then W [160][64], 128-byte rows XOR-swizzled through the SOURCE
//     address as in gemm2_kernel), filled by LDS-DMA (`buffer_load_dwordx4 ... lds`, 9 pieces of 1 KB per loader wave per
//     step) three to four steps ahead behind a COUNTED vmcnt: 110-140 KB in flight per CU where the 128x128 kernel has 32-40.
//     The ring runs across segment boundaries: the next tile's first steps land while the current tile's epilogue runs.
//   * ONE barrier per K step, placed between the two 32-deep halves of the step.  Multiplying waves: { MFMAs of half 0 |
//     fragment reads of half 1 } -> barrier -> { MFMAs of half 1 | fragment reads of half 0 of step i+1 }.  Loader waves:
//     wait (counted vmcnt) until their pieces of step i+1 have landed -> barrier -> issue step i+4 into the slot of step i,
//     whose last fragment reads every multiplying wave retired before it arrived at that barrier.  A staged slot is read
//     only after the wait + barrier that retire it.
//   * Epilogue (first form: 27 000 cycles per tile, more than a 20-step K loop -- global loads issued after the first C store
//     wait for the stores on gfx9's single vmcnt, and the spill reloads of a 256-register budget did the same; now ~3 000):
//     EVERY global load of the tile -- bias, LayerNorm column sums and row statistics, per-frame bias2 rows, row scales, the
//     residual tile -- is issued up front, the per-column / per-row constants go to a wave-private LDS area, and the four
//     16-row blocks are then transposed through LDS (16 x 80 fp32, the ring slot of the step just consumed) and stored as
//     16-byte vectors with no memory read in between.  Same fused operations and arithmetic order as gemm2_kernel /
//     gemm3_kernel: LayerNorm (rstd * (acc - mean * G[n]) from hallo_row_stats), bias, per-frame bias2, row scale, alpha,
//     leading-column scale, residual, SiLU / ReLU, fp32 output.
//
// Results differ from the 128x128 / big-tile kernels only by the summation order inside the K loop (fp32).
#include "common.h"
#include "gemm_args.h"
#include <type_traits>

namespace hallo {

namespace {
constexpr int G4_BM = 128, G4_BN = 160, G4_BK = 64, G4_ST = 4;
constexpr int G4_A_EL = G4_BM * G4_BK;                    // 8192 elements
constexpr int G4_SLOT_EL = (G4_BM + G4_BN) * G4_BK;       // 18432 elements = 36864 bytes
constexpr unsigned G4_OOB = 0xFFFFFFFFu;
constexpr int G4_PART_FLOATS = G4_BM * G4_BN;             // one partial tile: 80 KB of fp32
// epilogue scratch inside a 36 KB ring slot (floats): 4 x [16 rows][80] transposition tiles, then 4 x 512 constants
constexpr int G4_SCR_T = 16 * 80, G4_SCR_C0 = 4 * G4_SCR_T, G4_SCR_C = 512;
// constants of a wave (float offsets): bias[80] | colsum[80] | bias2 group A [80] | bias2 group B [80] | (mean, rstd)[64] | rowscale*alpha[64]
constexpr int G4_C_BIAS = 0, G4_C_COLSUM = 80, G4_C_B2A = 160, G4_C_B2B = 240, G4_C_STATS = 320, G4_C_ROWS = 448;
static_assert((G4_SCR_C0 + 4 * G4_SCR_C) * 4 <= G4_SLOT_EL * 2, "epilogue scratch must fit one ring slot");
}  // namespace

template <typename T> struct Mfma16;
template <> struct Mfma16<_Float16> {
  static __device__ __forceinline__ f32x4 run(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma16<__bf16> {
  static __device__ __forceinline__ f32x4 run(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

#define G4_BLOAD(rs, ldsptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ldsptr), 16, (int)(voff), (int)(soff), 0, 0)
#define G4_INLINE __attribute__((always_inline))

// A workgroup's segment list, walked one K step at a time (all fields wave-uniform): `dp` whole tiles, then at most one
// tail segment (tile s.dp * G + wl / parts, K steps [q * per, (q + 1) * per) with q = wl % parts).
struct G4Cursor {
  int j;              // segment index: j < dp data-parallel round j, j == dp the tail segment
  int tile, kt, kb, ke;
  bool done;
};

template <typename T>
__global__ __launch_bounds__(512, 2) void gemm4_kernel(const GemmArgs p, const G4Sched s) {
  using V8 = typename Vec<T>::v8;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  __shared__ __attribute__((aligned(16))) T smem[G4_ST * G4_SLOT_EL];      // 147456 bytes: one workgroup per CU

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bool producer = wave_u >= 4;          // waves 4-7: LDS-DMA issue only
  const int cw = wave_u & 3;                   // multiplying wave index / the piece family (cw + 4 i) of a loader wave
  const int wm = cw >> 1, wn = cw & 1;
  const int l15 = lane & 15, lg = lane >> 4;

  const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ B = reinterpret_cast<const T*>(p.B);
  const int G = (int)gridDim.x;
  const int bid = (int)blockIdx.x;
  const int wl = xcd_remap(bid, G);            // logical index in the tail deal: the parts of a tile share an XCD (and its L2)
  const int nk = s.nk;
  const bool in_tail = wl < s.R * s.parts;     // this workgroup has a tail segment
  const int tail_q = wl % s.parts;
  const int tail_kb = tail_q * s.per, tail_ke = min(nk, tail_kb + s.per);
  const int n_seg = s.dp + (in_tail ? 1 : 0);
  const int n_it = s.dp * nk + (in_tail ? tail_ke - tail_kb : 0);
  if (n_it <= 0) return;

  auto load_segment = [&](G4Cursor& c) G4_INLINE {
    if (c.j < s.dp) {
      c.tile = xcd_remap(c.j * G + bid, s.dp * G);
      c.kb = c.kt = 0; c.ke = nk; c.done = false;
    } else if (c.j == s.dp && in_tail) {
      c.tile = s.dp * G + wl / s.parts;
      c.kb = c.kt = tail_kb; c.ke = tail_ke; c.done = false;
    } else {
      c.done = true;
    }
  };

  auto stamp = [&](int k) G4_INLINE {
    if (s.dbg && bid == 0 && tid == 0) s.dbg[k] = (long long)__builtin_amdgcn_s_memtime();
  };

  // =================================================================================================== loader waves
  if (producer) {
    const int lrow = lane >> 3, lp = lane & 7;
    const int lc = lp ^ (((lane >> 4) + 4 * (cw & 1)) & 7);       // logical 16-byte chunk this lane fetches (piece parity = cw parity)
    auto clamp32 = [](long bytes) G4_INLINE { return (int)(bytes > 0xFFFFFFFFL ? 0xFFFFFFFFL : bytes); };
    __amdgpu_buffer_rsrc_t rsA, rsB;
    unsigned a_voff[4], w_voff[5];
    auto setup_loader = [&](int tile) G4_INLINE {
      const int tile_m = tile / s.tiles_n, tile_n = tile - tile_m * s.tiles_n;
      const int m0 = tile_m * G4_BM, n0 = tile_n * G4_BN;
      const int rows = min(p.M - m0, G4_BM), wrows = min(p.N - n0, G4_BN);
      rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(A + (long)m0 * p.lda), 0, clamp32(((long)(rows - 1) * p.lda + p.K) * 2), 0x00020000);
      rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(B + (long)n0 * p.ldb), 0, clamp32(((long)(wrows - 1) * p.ldb + p.K) * 2), 0x00020000);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = (cw + 4 * i) * 8 + lrow;
        a_voff[i] = (r < rows) ? (unsigned)(((long)r * p.lda + lc * 8) * 2) : G4_OOB;
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int r = (cw + 4 * i) * 8 + lrow;
        w_voff[i] = (r < wrows) ? (unsigned)(((long)r * p.ldb + lc * 8) * 2) : G4_OOB;
      }
    };
    G4Cursor dc;
    dc.j = 0;
    load_segment(dc);
    setup_loader(dc.tile);
    int dj = 0;                                           // steps issued so far
    auto issue = [&]() G4_INLINE {                        // the 9 pieces of step dj -> ring slot dj & 3
      T* dA = smem + (dj & 3) * G4_SLOT_EL;
      T* dW = dA + G4_A_EL;
      const int soff = dc.kt * G4_BK * 2;
#pragma unroll
      for (int q = 0; q < 5; ++q) G4_BLOAD(rsB, dW + (cw + 4 * q) * 512, w_voff[q], soff);
#pragma unroll
      for (int q = 0; q < 4; ++q) G4_BLOAD(rsA, dA + (cw + 4 * q) * 512, a_voff[q], soff);
      ++dj;
      if (++dc.kt == dc.ke) {
        ++dc.j;
        load_segment(dc);
        if (!dc.done) setup_loader(dc.tile);
      }
    };
    auto wait_younger = [&](int younger) G4_INLINE {     // whole steps issued behind the one waited for
      if (younger >= 3) asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
      else if (younger == 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    for (int q = 0; q < min(n_it, 4); ++q) issue();
    wait_younger(dj - 1);
    __builtin_amdgcn_s_barrier();                         // [B0] step 0 visible
    // mirror of the multiplying waves' barrier sequence
    int i = 0;
    for (int sg = 0; sg < n_seg; ++sg) {
      const bool tail_seg = sg == s.dp;
      const int steps = tail_seg ? tail_ke - tail_kb : nk;
      for (int q = 0; q < steps; ++q, ++i) {
        if (i + 1 < n_it) wait_younger(dj - (i + 2));     // my pieces of step i + 1 have landed
        __builtin_amdgcn_s_barrier();                     // [mid i]
        if (q + 1 < steps && dj < n_it) issue();          // step i + 4 -> slot i & 3 (not behind a segment's last step: epilogue scratch)
      }
      __builtin_amdgcn_s_barrier();                       // [scratch]
      if (tail_seg && s.parts > 1) {                      // partial tile: the three barriers of the hand-over
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
      }
      if (sg + 1 == n_seg) break;
      __builtin_amdgcn_s_barrier();                       // [restore] the epilogue's scratch reads are retired
      if (dj < n_it && dj == i + 3) issue();              // the step the segment's last K step left out
    }
    return;
  }

  // =================================================================================================== multiplying waves
  // fragment addressing: row = block * 16 + l15, 16-byte chunk (ks * 4 + lg) ^ ((row >> 1) & 7)
  const int xs = (lane >> 1) & 7;
  const int fa_row = (wm * 64 + l15) * G4_BK;                 // + bm * 16 * BK
  const int fw_row = G4_A_EL + (wn * 80 + l15) * G4_BK;       // + bn * 16 * BK
  V8 fa[2][4], fw[2][5];
  auto frag = [&](int slot, int ks, V8* a, V8* w) G4_INLINE {
    const T* base = smem + slot * G4_SLOT_EL;
    const int co = ((ks * 4 + lg) ^ xs) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = ld8<T>(base + fa_row + j * 16 * G4_BK + co);
#pragma unroll
    for (int i = 0; i < 5; ++i) w[i] = ld8<T>(base + fw_row + i * 16 * G4_BK + co);
  };

  f32x4 acc[5][4];       // [bn][bm]: D[n = bn*16 + 4*lg + r][m = bm*16 + l15]
  auto zero_acc = [&]() G4_INLINE {
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  };
  zero_acc();

  const T* bias = reinterpret_cast<const T*>(p.bias);
  const T* bias2 = reinterpret_cast<const T*>(p.bias2);
  const T* res = reinterpret_cast<const T*>(p.residual);
  T* C = reinterpret_cast<T*>(p.C);
  float* Cf = reinterpret_cast<float*>(p.C);
  const bool lnf = p.ln_colsum != nullptr;

  // ---- epilogue of a finished tile (accumulators final); `scr_all` = the ring slot of the step just consumed ----
  auto epilogue = [&](int tile, float* scr_all) G4_INLINE {
    const int tile_m = tile / s.tiles_n, tile_n = tile - tile_m * s.tiles_n;
    const int m0 = tile_m * G4_BM + wm * 64, n0 = tile_n * G4_BN + wn * 80;       // this wave's 64 x 80 corner
    float* scr = scr_all + cw * G4_SCR_T;
    float* cst = scr_all + G4_SCR_C0 + cw * G4_SCR_C;
    // read-phase lane mapping of pass ps (0, 1: 16 rows x 4 column groups of 8; 2: 32 lanes, 16 rows x 2 groups)
    int prow[3], pcol[3];
    bool pon[3];
    prow[0] = prow[1] = lane >> 2; pcol[0] = (lane & 3) * 8; pcol[1] = 32 + (lane & 3) * 8; pon[0] = pon[1] = true;
    prow[2] = (lane >> 1) & 15; pcol[2] = 64 + (lane & 1) * 8; pon[2] = lane < 32;
    // (1) every global load of the tile, all in flight together: per-column constants (lane -> columns lane, 64 + lane), per-row
    //     constants (lane -> row lane), the residual tile (twelve 16-byte vectors per lane)
    const int g_a = m0 / p.bias2_rpg;                      // bias2 row group of the wave's first row (rows/group >= 128: <= 2 groups per wave)
    float c_bias[2] = {0.0f, 0.0f}, c_cs[2] = {0.0f, 0.0f}, c_b2a[2] = {0.0f, 0.0f}, c_b2b[2] = {0.0f, 0.0f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = h * 64 + lane, n = n0 + c;
      if (c < 80 && n < p.N) {
        if (bias) c_bias[h] = to_f32(bias[n]);
        if (lnf) c_cs[h] = p.ln_colsum[n];
        if (bias2) {
          c_b2a[h] = to_f32(bias2[(long)g_a * p.bias2_ld + n]);
          if ((long)(g_a + 1) * p.bias2_rpg < p.M) c_b2b[h] = to_f32(bias2[(long)(g_a + 1) * p.bias2_ld + n]);
        }
      }
    }
    float2 c_st = {0.0f, 1.0f};
    float c_rs = p.alpha;
    {
      const int m = m0 + lane;
      if (m < p.M) {
        if (lnf) c_st = *reinterpret_cast<const float2*>(p.ln_stats + 2 * (long)m);
        if (p.rowscale) c_rs = p.rowscale[m] * p.alpha;
      }
    }
    V8 rpre[4][3];
    if (res) {
#pragma unroll
      for (int bm = 0; bm < 4; ++bm)
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) {
          const int m = m0 + bm * 16 + prow[ps], n = n0 + pcol[ps];
          rpre[bm][ps] = (pon[ps] && m < p.M && n < p.N) ? ld8<T>(res + (long)m * p.ldr + n) : zero8<T>();
        }
    }
    // (2) constants -> this wave's LDS area
    cst[G4_C_BIAS + lane] = c_bias[0]; cst[G4_C_COLSUM + lane] = c_cs[0]; cst[G4_C_B2A + lane] = c_b2a[0]; cst[G4_C_B2B + lane] = c_b2b[0];
    if (lane < 16) {
      cst[G4_C_BIAS + 64 + lane] = c_bias[1]; cst[G4_C_COLSUM + 64 + lane] = c_cs[1]; cst[G4_C_B2A + 64 + lane] = c_b2a[1]; cst[G4_C_B2B + 64 + lane] = c_b2b[1];
    }
    *reinterpret_cast<float2*>(cst + G4_C_STATS + 2 * lane) = c_st;
    cst[G4_C_ROWS + lane] = c_rs;
    stamp(32);
    // (3) four 16-row blocks: transpose through LDS, apply, store.  No memory read from here on.
#pragma unroll
    for (int bm = 0; bm < 4; ++bm) {
      // write phase: acc[bn][bm] -> scr[row = l15][16-byte chunk bn*4 + lg, XORed with (row >> 2) & 3]
#pragma unroll
      for (int bn = 0; bn < 5; ++bn) {
        const int chunk = (bn * 4 + lg) ^ ((l15 >> 2) & 3);
        *reinterpret_cast<f32x4*>(scr + l15 * 80 + chunk * 4) = acc[bn][bm];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ps = 0; ps < 3; ++ps) {
        const int rr = prow[ps], lr = bm * 16 + rr;          // row inside the 16-row block / inside the wave tile
        const int m = m0 + lr, n = n0 + pcol[ps];
        const int c0 = pcol[ps] >> 2;                        // first of the two 16-byte chunks
        const int sw = (rr >> 2) & 3;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(scr + rr * 80 + ((c0 ^ sw) * 4));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(scr + rr * 80 + (((c0 + 1) ^ sw) * 4));
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(cst + G4_C_BIAS + pcol[ps]), b1 = *reinterpret_cast<const f32x4*>(cst + G4_C_BIAS + pcol[ps] + 4);
        float o[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (lnf) {       // fused LayerNorm: rstd * (acc - mean * G[n]) with the statistics of hallo_row_stats
          const float2 st = *reinterpret_cast<const float2*>(cst + G4_C_STATS + 2 * lr);
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(cst + G4_C_COLSUM + pcol[ps]), g1 = *reinterpret_cast<const f32x4*>(cst + G4_C_COLSUM + pcol[ps] + 4);
          const float c1 = -st.x * st.y;
#pragma unroll
          for (int j = 0; j < 4; ++j) { o[j] = __builtin_fmaf(st.y, o[j], c1 * g0[j]); o[4 + j] = __builtin_fmaf(st.y, o[4 + j], c1 * g1[j]); }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] += b0[j]; o[4 + j] += b1[j]; }
        if (bias2) {
          const int off = ((m / p.bias2_rpg) != g_a ? G4_C_B2B : G4_C_B2A) + pcol[ps];
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(cst + off), t1 = *reinterpret_cast<const f32x4*>(cst + off + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { o[j] += t0[j]; o[4 + j] += t1[j]; }
        }
        const float rs = cst[G4_C_ROWS + lr] * ((n < p.lead_cols) ? p.lead_alpha : 1.0f);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] *= rs;
        if (res) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += to_f32(rpre[bm][ps][j]);
        }
        if (p.act == ACT_SILU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = silu_f(o[j]);
        } else if (p.act == ACT_RELU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], 0.0f);
        }
        if (!(pon[ps] && m < p.M && n < p.N)) continue;
        if (p.out_f32) {
          float* cp = Cf + (long)m * p.ldc + n;
          *reinterpret_cast<f32x4*>(cp) = f32x4{o[0], o[1], o[2], o[3]};
          *reinterpret_cast<f32x4*>(cp + 4) = f32x4{o[4], o[5], o[6], o[7]};
        } else {
          V8 wv;
#pragma unroll
          for (int j = 0; j < 8; ++j) wv[j] = from_f32<T>(o[j]);
          st8<T>(C + (long)m * p.ldc + n, wv);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this block's scratch reads retired before the next block's writes
      __builtin_amdgcn_wave_barrier();
      stamp(33 + bm);
    }
  };

  // ---- end of a segment: a whole tile goes straight to the epilogue; a partial tile (tail, parts > 1) is published with
  //      WRITE-THROUGH (sc1) 16-byte stores -- no release fence, which would write back this XCD's whole L2 -- every wave drains
  //      its stores, one lane takes a ticket, and the last arriver of the tile reads the parts back with sc1 loads in K order.
  auto clamp32p = [](long bytes) G4_INLINE { return (int)(bytes > 0xFFFFFFFFL ? 0xFFFFFFFFL : bytes); };
  const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(s.part, 0, clamp32p((long)G * G4_PART_FLOATS * 4), 0x00020000);
  auto finish_segment = [&](int tile, bool partial, float* scr_all) G4_INLINE {
    if (partial) {
      const int tt = tile - s.dp * G;                                   // tail tile index: its parts are workgroups tt * parts + q
      const unsigned lane_off = (unsigned)(((cw * 20 * 64 + lane) * 4) * 4);      // bytes inside a partial tile
      const unsigned mine = (unsigned)(wl * (G4_PART_FLOATS * 4));
#pragma unroll
      for (int bn = 0; bn < 5; ++bn)
#pragma unroll
        for (int bm = 0; bm < 4; ++bm)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[bn][bm]), rsP, lane_off + (bn * 4 + bm) * 1024, mine, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // every wave: its write-through stores have reached memory
      __builtin_amdgcn_s_barrier();
      int* flag = reinterpret_cast<int*>(scr_all);
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(s.cnt + tt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = old == s.parts - 1;
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int last = *reinterpret_cast<volatile int*>(flag);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                     // everybody has read the flag before the scratch is reused
      if (!last) return;
      zero_acc();
      for (int q = 0; q < s.parts; ++q) {
        const unsigned src = (unsigned)((tt * s.parts + q) * (G4_PART_FLOATS * 4));
#pragma unroll
        for (int bn = 0; bn < 5; ++bn)
#pragma unroll
          for (int bm = 0; bm < 4; ++bm) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsP, lane_off + (bn * 4 + bm) * 1024, src, 16));
            acc[bn][bm][0] += v[0]; acc[bn][bm][1] += v[1]; acc[bn][bm][2] += v[2]; acc[bn][bm][3] += v[3];
          }
      }
      if (tid == 0) __hip_atomic_store(s.cnt + tt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
    }
    epilogue(tile, scr_all);
  };

  // ---- the K-step stream ----
  G4Cursor cc;
  cc.j = 0;
  load_segment(cc);
  stamp(0);
  __builtin_amdgcn_s_barrier();                            // [B0]
  stamp(1);
  frag(0, 0, fa[0], fw[0]);
  // One K step; NEXT: the segment goes on (the half-0 fragments of step i + 1 are prefetched in half 1).  A segment's last step
  // prefetches nothing: the epilogue needs the registers, and the next segment's first fragments are read behind it.
  auto kstep = [&](auto next_c, int i) G4_INLINE {
    constexpr bool NEXT = decltype(next_c)::value;
    const int slot = i & 3;
    frag(slot, 1, fa[1], fw[1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int bn = 0; bn < 5; ++bn)
#pragma unroll
      for (int bm = 0; bm < 4; ++bm) acc[bn][bm] = Mfma16<T>::run(fw[0][bn], fa[0][bm], acc[bn][bm]);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my reads of slot i are retired: the loaders may refill it
    __builtin_amdgcn_s_barrier();                            // [mid i] step i + 1 is visible
    if (NEXT) frag((i + 1) & 3, 0, fa[0], fw[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int bn = 0; bn < 5; ++bn)
#pragma unroll
      for (int bm = 0; bm < 4; ++bm) acc[bn][bm] = Mfma16<T>::run(fw[1][bn], fa[1][bm], acc[bn][bm]);
    __builtin_amdgcn_sched_barrier(0);
    ++cc.kt;
    if (i < 24) stamp(2 + i);
  };
  int i = 0;
  for (;;) {
    while (cc.kt + 1 < cc.ke) { kstep(std::true_type{}, i); ++i; }
    kstep(std::false_type{}, i); ++i;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // [scratch] the slot of step i - 1 is the epilogue's scratch now
    stamp(30);
    finish_segment(cc.tile, cc.j == s.dp && s.parts > 1, reinterpret_cast<float*>(smem + ((i - 1) & 3) * G4_SLOT_EL));
    stamp(31);
    if (i >= n_it) break;
    zero_acc();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // [restore]
    ++cc.j;
    load_segment(cc);
    frag(i & 3, 0, fa[0], fw[0]);                            // step i was made visible by the barrier in the middle of step i - 1
  }
}
#undef G4_BLOAD
#undef G4_INLINE

static long long* g_g4_dbg = nullptr;       // hallo_gemm4_debug_buffer: s_memtime stamps of workgroup 0 (tools/cbench)
void set_gemm4_debug_buffer(long long* p) { g_g4_dbg = p; }

static int g4_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

// Workspace of the split tail: one partial tile per workgroup + the per-tile counters (last 64 KB of the caller's buffer).
constexpr int64_t G4_CNT_BYTES = 65536;

bool gemm4_plan(const GemmArgs& a, int64_t ws_bytes, G4Sched* out) {
  if (a.K % G4_BK != 0 || a.K < 4 * G4_BK) return false;
  if (a.bias2 && a.bias2_rpg < G4_BM) return false;               // the epilogue stages at most two bias2 row groups per 64-row wave tile
  if (a.residual && !a.res_vec_ok) return false;
  G4Sched s;
  s.tiles_m = (a.M + G4_BM - 1) / G4_BM;
  s.tiles_n = (a.N + G4_BN - 1) / G4_BN;
  s.nk = a.K / G4_BK;
  const int tiles = s.tiles_m * s.tiles_n;
  const int cus = g4_num_cus();
  s.G = tiles < cus ? tiles : cus;
  s.dp = tiles / s.G;
  s.R = tiles - s.dp * s.G;
  s.parts = 1;
  if (tiles < cus) { s.G = cus; s.dp = 0; s.R = tiles; }           // fewer tiles than CUs: one (possibly split) tail segment
  // Split the tail tiles' K loop when it is long enough to pay for the hand-over through memory.  Cost model from the stamps of
  // tools/cbench (MI355X): a K step ~0.5 us; publishing a partial tile ~3 us, the reducer ~1.2 us per part it reads.
  if (s.R > 0) {
    const int64_t avail = ws_bytes - G4_CNT_BYTES;
    float best = 0.5f * s.nk;
    for (int parts = 2; parts <= 8; parts *= 2) {
      if ((int64_t)s.R * parts > s.G || (s.nk + parts - 1) / parts < 4) break;
      if ((int64_t)s.R * parts * G4_PART_FLOATS * 4 > avail || s.R * 4 > G4_CNT_BYTES) break;
      const float cost = 0.5f * ((s.nk + parts - 1) / parts) + 3.0f + 1.2f * parts;
      if (cost < best) { best = cost; s.parts = parts; }
    }
  }
  s.per = (s.nk + s.parts - 1) / s.parts;
  if ((s.parts - 1) * s.per >= s.nk) return false;                 // (every part must own at least one K step)
  if (tiles < cus && s.parts == 1) s.G = tiles;                    // nothing split: no idle workgroups
  s.part = nullptr; s.cnt = nullptr; s.dbg = nullptr;
  *out = s;
  return true;
}

template <typename T>
int launch_gemm4(const GemmArgs& a, G4Sched s, void* ws, int64_t ws_bytes, hipStream_t st) {
  if (s.parts > 1) {
    s.part = reinterpret_cast<float*>(ws);
    s.cnt = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + ws_bytes - G4_CNT_BYTES);
  }
  s.dbg = g_g4_dbg;
  hipLaunchKernelGGL((gemm4_kernel<T>), dim3((unsigned)s.G), dim3(512), 0, st, a, s);
  return 0;
}
template int launch_gemm4<_Float16>(const GemmArgs&, G4Sched, void*, int64_t, hipStream_t);
template int launch_gemm4<__bf16>(const GemmArgs&, G4Sched, void*, int64_t, hipStream_t);

}  // namespace hallo
