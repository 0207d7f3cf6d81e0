// Row-stationary GEMM for K = 320, second form: the epilogue of one pair of 32-row W blocks is issued, element by element,
// between the MFMAs of the NEXT pair.
//
// Same calls as gemm_rs.hip (to_q|to_k|to_v with norm1 folded in: hallo/models/mutual_self_attention.py:253-303,
// attention.py:828-884, motion_module.py:553-609; FeedForward net.0 / GEGLU with norm3 folded in: attention.py:601,905,
// motion_module.py:420) at the 64 x 64-latent level, where K = 320.
//
// Why a second form.  Ablation of gemm_rs.hip on MI355X (tools/cbench gemm ... rsdbg=N, 65536 rows, bf16):
//     to_q|k|v (N = 960):  full 74 us | no stores 55 | no epilogue 43 | MFMA loop alone 35
//     GEGLU  (N = 2 x 1280): full 180 us | no stores 162 | no epilogue 111 | MFMA loop alone 86
// The epilogue (LayerNorm affine, erf-GELU, pack: ~22 VALU instructions per GEGLU output) costs as much as the MFMAs and
// does NOT overlap them: there the K loop is outermost (a [128 rows][64 k] chunk per barrier), so all accumulators of an N
// tile complete together, the epilogue is one VALU block per 5 chunks, and the per-chunk workgroup barrier stops the other
// waves while one half is inside it.
// Here the loops are turned round: a step = one PAIR of 32-row W blocks (GEGLU: the value block and the gate block of the
// same 32 output columns; otherwise 64 consecutive columns) with the WHOLE K = 320: 40 MFMAs per wave per barrier into two
// fresh accumulators.  The accumulators of the previous pair are then final, and its epilogue is interleaved at element
// granularity with those 40 MFMAs (one or two output elements per k16 step, pinned with sched_barrier): every wave does
// the same MFMA + VALU mix between two barriers, matrix pipe and VALU run concurrently inside a wave and across the two
// waves of a SIMD.
//   * W chunk = 5 sub-tiles [64 rows][64 k] (gemm_rs.hip's swizzled 128-byte-row image), 40 KB, ring of 3: chunk c + 2 is
//     issued behind the barrier of step c (80 KB per CU in flight), one counted vmcnt per step.
//   * A rows, LayerNorm statistics, constants in LDS, swapped operands, half-swap 16-byte stores: as in gemm_rs.hip.
#include "gemm_args.h"
#include <type_traits>

namespace hallo {

#if defined(__HIP_DEVICE_COMPILE__)
#define R2_KEEP(x) asm volatile("" :: "v"(x))
#else
#define R2_KEEP(x) (void)(x)
#endif

namespace {
constexpr int R2_K16 = 20;                                   // K = 320
constexpr int R2_SUB = 5;                                    // [64 rows][64 k] sub-tiles per chunk
constexpr int R2_SUB_BYTES = 64 * 128;
constexpr int R2_CHUNK_BYTES = R2_SUB * R2_SUB_BYTES;        // 40960
constexpr int R2_RING = 3;
#ifndef R2_EPI_REGIONS
#define R2_EPI_REGIONS 8     // of the 10 two-k16 regions of a step, how many carry the previous pair's epilogue (A/B: 4)
#endif
constexpr int R2_MAX_WROWS = 2560;
constexpr int R2_CONST_N = R2_MAX_WROWS + 64;                // + one pair of slack: a partial last pair indexes past N
constexpr int R2_OFF_BIAS = R2_RING * R2_CHUNK_BYTES;        // fp32 [R2_CONST_N]
constexpr int R2_LDS = R2_OFF_BIAS + R2_CONST_N * 4;         // 133376 B: one workgroup per CU
static_assert(R2_LDS <= 160 * 1024, "LDS");
}  // namespace

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// ABL (timing ablations, wrong results; hallo_set_option("gemm_rs_dbg", bits)): 1 = no stores, 2 = no epilogue, 5 = no stores and
// no MFMAs (the epilogue's VALU work alone)
// KVS (hallo_gemm_desc.kv_out, round 6): the columns from p.kv_col0 on -- K then V of 8 heads x 40 -- are stored head-major,
// [image][head][row][40], where hallo_attention's LDS-DMA reads a 64-key tile as one contiguous 5 KB piece.  A lane's 16-byte store
// covers 8 columns of one head (40 = 5 x 8), so only the store ADDRESS changes: per store two constant divisions on the column.
template <typename T, bool GEGLU, bool LNF, int ABL, bool KVS = false>
__global__ __launch_bounds__(512, 2) void gemm_rs2_kernel(const GemmArgs p) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  constexpr int K16 = R2_K16, WG_ROWS = 256;
  constexpr int PAIR_COLS = GEGLU ? 32 : 64;       // output columns per step
  constexpr int NELEM = GEGLU ? 16 : 32;           // epilogue elements per lane per step
  constexpr int EPS = NELEM / 16;                  // elements per k16 step (k16 steps 0 .. 15 carry the epilogue)

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* const lds = (lds_u8*)smem;
  float* const sBias = reinterpret_cast<float*>(smem + R2_OFF_BIAS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, l31 = lane & 31;
  // Work decomposition (launch_gemm_rs2): the first p.tiles_m workgroups own one 256-row block each and ALL of N.  A last,
  // under-filled round of row blocks would leave most CUs idle for a whole pass over W (73728 rows = 288 blocks = 1.125
  // rounds), so the remaining blocks are each shared by p.tiles_n workgroups that take a slice of the pairs: the tail costs
  // ~1 / tiles_n of a round (A rows of the tail re-read tiles_n times: a few MB).
  int rowblk = blockIdx.x, pair0 = 0;
  int npairs = (p.N + PAIR_COLS - 1) / PAIR_COLS;
  if (rowblk >= p.tiles_m) {
    const int t = rowblk - p.tiles_m, sl = t % p.tiles_n;
    rowblk = p.tiles_m + t / p.tiles_n;
    pair0 = (sl * npairs) / p.tiles_n;
    npairs = ((sl + 1) * npairs) / p.tiles_n - pair0;
  }
  const int m0 = rowblk * WG_ROWS + wave * 32;

  const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(p.B);
  T* __restrict__ C = reinterpret_cast<T*>(p.C);
  const int wrows = GEGLU ? 2 * p.N : p.N;

  // ---- W stream.  Per sub-tile a wave issues ONE LDS-DMA instruction: 8 rows x 128 B (lane = row wave*8 + lane/8, 16-byte
  // piece lane%8, XOR-swizzled through the source address).  The pair advance is part of the per-lane offset so that rows
  // past the end of W fall outside the descriptor (zero fill); the K advance (64 elements per sub-tile) is the scalar offset.
  const long w_bytes = (((long)wrows - 1) * p.ldb + p.K) * 2;
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(W), 0, (int)(w_bytes > 0xFFFFFFFFL ? 0xFFFFFFFFL : w_bytes), 0x00020000);
  unsigned w_voff;
  {
    const int row = wave * 8 + (lane >> 3);                       // row of the 64-row sub-tile
    const int logical = (lane & 7) ^ ((row >> 1) & 7);
    const long wrow = GEGLU ? (long)(row & 31) + (row >= 32 ? p.N : 0) : row;
    w_voff = (unsigned)((wrow * p.ldb + logical * 8) * 2);
  }
  const unsigned pair_step = (unsigned)((long)PAIR_COLS * p.ldb * 2);
  int issued = 0;
  auto issue_next = [&]() {
    const int slot = issued % R2_RING;
    const int dst = slot * R2_CHUNK_BYTES + wave_u * 1024;
    const unsigned vo = w_voff + (unsigned)(pair0 + issued) * pair_step;
#pragma unroll
    for (int s = 0; s < R2_SUB; ++s)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(lds + dst + s * R2_SUB_BYTES), 16,
                                               (int)vo, s * 128, 0, 0);
    ++issued;
  };
  if (npairs > 0) issue_next();
  if (npairs > 1) issue_next();

  // ---- epilogue constants of all W rows -> LDS (fp32) ----
  {
    const T* bias = reinterpret_cast<const T*>(p.bias);
    const T* bias2 = reinterpret_cast<const T*>(p.bias2);
    const long b2row = bias2 ? (long)((rowblk * WG_ROWS) / p.bias2_rpg) * p.bias2_ld : 0;
    for (int n = tid; n < R2_CONST_N; n += 512) {
      float b = 0.0f;
      if (n < wrows) {
        if (bias) b = to_f32(bias[n]);
        if (bias2) b += to_f32(bias2[b2row + n]);
        if (!GEGLU) b *= p.alpha * (n < p.lead_cols ? p.lead_alpha : 1.0f);     // the column scale is folded into the constants
        else b *= (n < p.N) ? GELU_U_INV : GELU_U_SCALE;    // value rows carry 1 / s, gate rows s = sqrt(log2(e) / 2): see gelu_u()
      }
      sBias[n] = b;
    }
  }

  // ---- A fragments (B operand): lane holds A[m0 + l31][k16*16 + hi*8 .. +8] ----
  V8 af[K16];
  {
    const int m = min(m0 + l31, p.M - 1);
    const T* arow = A + (long)m * p.lda + hi * 8;
#pragma unroll
    for (int k = 0; k < K16; ++k) af[k] = ld8<T>(arow + k * 16);
  }
  // LayerNorm: statistics of the lane's row from the fragments (packed dot products), then the fragments themselves are
  // normalised in place, (x - mean) * rstd rounded to T -- the rounding nn.LayerNorm's output has in the reference.  gamma is
  // folded into W and beta . W^T into the bias by the caller (hallo_gemm's ln_colsum contract; the column sums themselves are
  // not needed here), so the epilogue is acc + bias: no per-element statistics arithmetic (the epilogue is what bounds
  // this kernel: every VALU instruction of it costs ~7 cycles of the SIMD).
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
    sm += __shfl_xor(sm, 32, 64);
    sq += __shfl_xor(sq, 32, 64);
    const float mean = sm / (float)p.K;
    const float rstd = rsqrtf(fmaxf(sq / (float)p.K - mean * mean, 0.0f) + p.ln_eps);
    const float shift = -mean * rstd;
#pragma unroll
    for (int k = 0; k < K16; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) af[k][e] = from_f32<T>(__builtin_fmaf(to_f32(af[k][e]), rstd, shift));
  }
  __syncthreads();      // constants visible (drains the two chunks in flight once; the counted waits below stay valid)

  // W fragment (A operand) of block nb, k16 step k: sub-tile k / 4, row nb * 32 + l31, 16-byte piece ((k % 4) * 2 + hi)
  // XOR-swizzled.  Four lane offsets (one per k % 4); slot, sub-tile and block are immediates on top of them.
  const int xsw = hi ^ ((l31 >> 1) & 7);
  int fo[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fo[kk] = l31 * 128 + ((kk * 2) ^ xsw) * 16;
  const int m = m0 + l31;
  T* const crow = C + (long)min(m, p.M - 1) * p.ldc;
  const bool row_live = m < p.M;
  // KVS: this lane's row inside the head-major tensors: image m / L, row m % L; head h and tensor t add h * L * 40 + t * stride
  T* kvrow = nullptr;
  if (KVS) {
    const int mm = min(m, p.M - 1), img = mm / p.kv_L, pos = mm - img * p.kv_L;
    kvrow = reinterpret_cast<T*>(p.kv_out) + ((long)img * 8 * p.kv_L + pos) * 40;
  }
  auto out_ptr = [&](const int nc) -> T* {
    if (KVS && nc >= p.kv_col0) {
      const unsigned c = (unsigned)(nc - p.kv_col0);          // < 640
      const unsigned t = c >= 320u ? 1u : 0u, cc = c - t * 320u, h = cc / 40u, d = cc - h * 40u;
      return kvrow + (long)t * p.kv_tstride + (long)h * p.kv_L * 40 + d;
    }
    return crow + nc;
  };
  const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  const float lead = p.lead_alpha, alpha = p.alpha;

  // ---- epilogue of one pair, one element (compile-time index E) at a time.  State between elements: the constants of the
  // current 4-column group, the packed halves of the current pair of groups. ----
  f32x4 cb0 = {0, 0, 0, 0}, cb1 = {0, 0, 0, 0};
  float e_cs = 1.0f;
  float e_f[4] = {0, 0, 0, 0};
  unsigned pk[2][2] = {{0, 0}, {0, 0}};
  typedef const __attribute__((address_space(3))) f32x4* ldsf4;
  // bv / bg: LDS byte addresses of this lane's value / gate constants of the pair being finished (set once per step)
  auto epi_elem = [&](auto e_c, const f32x16* acc, const int n0, const unsigned bv, const unsigned bg) {
    constexpr int E = decltype(e_c)::value;
    constexpr int ob = GEGLU ? 0 : E / 16, g = (E % 16) / 4, j = E % 4;
    if (j == 0) {
      cb0 = *(ldsf4)(lds + bv + (ob * 32 + 8 * g) * 4);
      if (GEGLU) cb1 = *(ldsf4)(lds + bg + 8 * g * 4);
      else e_cs = alpha * ((n0 + ob * 32 < p.lead_cols) ? lead : 1.0f);
    }
    if (GEGLU) {
      // gate argument u = s * (acc + b), value v' = (acc + b) / s, out = v' * gelu_u(u) (common.h); s rides in the LDS constants
      const float hv = __builtin_fmaf(GELU_U_INV, acc[0][g * 4 + j], cb0[j]);
      const float gu = __builtin_fmaf(GELU_U_SCALE, acc[1][g * 4 + j], cb1[j]);
      e_f[j] = hv * gelu_u(gu);
    } else {
      e_f[j] = __builtin_fmaf(e_cs, acc[ob][g * 4 + j], cb0[j]);      // cb0 = cs * bias
    }
    if (j == 3) {
      const V4 e_o = {from_f32<T>(e_f[0]), from_f32<T>(e_f[1]), from_f32<T>(e_f[2]), from_f32<T>(e_f[3])};   // two packed converts
      const uint2 u = __builtin_bit_cast(uint2, e_o);
      pk[g & 1][0] = u.x; pk[g & 1][1] = u.y;
      if (g & 1) {
        // groups g-1 (cols 8(g-1) + 4hi ..) and g: after the half swap lanes hi = 0 hold cols 8(g-1) .. +7, lanes hi = 1 the next 8
        const auto x = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
        const auto y = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
        const int nc = n0 + ob * 32 + 8 * (g - 1) + 8 * hi;
        if (ABL & 1) { R2_KEEP(x); R2_KEEP(y); }
        else if (row_live && nc < p.N) *reinterpret_cast<uint4*>(out_ptr(nc)) = make_uint4(x[0], y[0], x[1], y[1]);
      }
    }
  };

  // ---- one step: chunk `pr` -> acc_cur (2 blocks x 20 k16), epilogue elements of the previous pair in between ----
  auto step = [&](auto prev_c, const int pr, f32x16* acc_cur, const f32x16* acc_prev) {
    constexpr bool PREV = decltype(prev_c)::value;
    // chunk pr has landed for this wave when at most the 5 DMAs of chunk pr+1 are younger (loads retire in order; stores in
    // between only make the count conservative)
    if (issued > pr + 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // chunk pr visible to everybody; everybody is done with chunk pr-1 (slot of chunk pr+2)
    if (issued < npairs) issue_next();
    // (the mask makes the range of the base visible to the compiler: it folds a constant into the 16-bit offset field of a DS
    // instruction only when base + offset provably does not wrap)
    const int n0p = (pair0 + pr - 1) * PAIR_COLS;
    const unsigned slot_off = (unsigned)((pr % R2_RING) * R2_CHUNK_BYTES);
    const lds_u8* const sw0 = lds + ((slot_off + (unsigned)fo[0]) & 0x1FFF0u);
    const lds_u8* const sw1 = lds + ((slot_off + (unsigned)fo[1]) & 0x1FFF0u);
    const lds_u8* const sw2 = lds + ((slot_off + (unsigned)fo[2]) & 0x1FFF0u);
    const lds_u8* const sw3 = lds + ((slot_off + (unsigned)fo[3]) & 0x1FFF0u);
    const unsigned bv = ((unsigned)(R2_OFF_BIAS + (n0p + 4 * hi) * 4)) & 0x3FFF0u;
    const unsigned bg = ((unsigned)(R2_OFF_BIAS + (p.N + n0p + 4 * hi) * 4)) & 0x3FFF0u;
    typedef const __attribute__((address_space(3))) V8* ldsv8;
    V8 fr[3][2];
#define R2_LOADW(set, k) _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) \
      fr[set][nb] = *(ldsv8)((((k) & 3) == 0 ? sw0 : ((k) & 3) == 1 ? sw1 : ((k) & 3) == 2 ? sw2 : sw3) + ((k) >> 2) * R2_SUB_BYTES + nb * 32 * 128)
    R2_LOADW(0, 0);
    R2_LOADW(1, 1);
    // Regions of 2 k16 steps (4 MFMAs), pinned with sched_barrier.  A region carries TWO (GEGLU) / FOUR elements of the
    // previous pair's epilogue: independent chains, so that the in-order issue of a wave does not sit on the latency of one
    // dependent chain (fma -> rcp -> 4 fma -> ...: ~130 cycles per element when issued alone, measured), sliced behind the
    // MFMAs (MFMA | VALU slice | MFMA | ...: two MFMAs back to back would hold everything behind the second one's wait for
    // the matrix pipe).
    static_for<0, K16 / 2>([&](auto rc) {
      constexpr int R = decltype(rc)::value;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int k = 2 * R + kk;
        if (k + 2 < K16) { R2_LOADW((k + 2) % 3, k + 2); }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          if constexpr (ABL & 4) { if (k == 0) acc_cur[nb] = zero16; acc_cur[nb][k & 15] += to_f32(fr[k % 3][nb][0]); }   // no MFMAs
          else acc_cur[nb] = Vec<T>::mfma32(fr[k % 3][nb], af[k], k == 0 ? zero16 : acc_cur[nb]);
        }
      }
      if constexpr (PREV && R < R2_EPI_REGIONS && !(ABL & 2)) {
        constexpr int PER = NELEM / R2_EPI_REGIONS;
        static_for<0, PER>([&](auto jc) { epi_elem(std::integral_constant<int, R * PER + decltype(jc)::value>{}, acc_prev, n0p, bv, bg); });
        constexpr int SLICE = (GEGLU ? 56 : 24) / R2_EPI_REGIONS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, SLICE, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
#undef R2_LOADW
  };

  f32x16 acc_a[2], acc_b[2];
  if (npairs > 0) {
    using TT = std::true_type;
    using FF = std::false_type;
    step(FF{}, 0, acc_a, acc_b);
    int pr = 1;
    for (; pr + 1 < npairs; pr += 2) {        // two steps per trip: the accumulator sets alternate by name
      step(TT{}, pr, acc_b, acc_a);
      step(TT{}, pr + 1, acc_a, acc_b);
    }
    // the last pair's epilogue has nothing to hide behind.  (Two explicit branches: a run-time selected pointer to the
    // accumulator set would push both sets into scratch memory.)
    const int n0l = (pair0 + npairs - 1) * PAIR_COLS;
    const unsigned bvl = (unsigned)(R2_OFF_BIAS + (n0l + 4 * hi) * 4), bgl = (unsigned)(R2_OFF_BIAS + (p.N + n0l + 4 * hi) * 4);
    if (pr < npairs) {
      step(TT{}, pr, acc_b, acc_a);
      if (ABL & 2) { R2_KEEP(acc_b[0]); R2_KEEP(acc_b[1]); }
      else static_for<0, NELEM>([&](auto ec) { epi_elem(ec, acc_b, n0l, bvl, bgl); });
    } else {
      if (ABL & 2) { R2_KEEP(acc_a[0]); R2_KEEP(acc_a[1]); }
      else static_for<0, NELEM>([&](auto ec) { epi_elem(ec, acc_a, n0l, bvl, bgl); });
    }
  }
}

// Routing rule (launch_gemm): K = 320 problems of the shapes gemm_rs.hip takes; the N-slice tail makes any row count fill
// the chip, so only a minimum size is required.
bool gemm_rs2_eligible(const GemmArgs& a, bool conv, bool geglu, int batch) {
  if (conv || batch != 1 || a.splits > 1 || a.K != 320) return false;
  if (a.residual || a.rowscale || a.bias_per_row || a.out_f32 || a.act != ACT_NONE) return false;
  if (a.ln_colsum && a.ln_stats) return false;                 // the caller already paid for a statistics pass
  const int wrows = geglu ? 2 * a.N : a.N;
  if ((a.N & 7) || (geglu && (a.N & 31)) || wrows > R2_MAX_WROWS) return false;
  if (a.lead_cols % 32) return false;
  if (a.bias2 && (a.bias2_rpg % 256)) return false;            // the bias2 row must be constant per workgroup
  if (a.M < 8192) return false;                                // 32 row blocks: below that the tiled kernels' finer grid wins
  if ((a.lda & 7) || (a.ldb & 7) || (a.ldc & 7) || (reinterpret_cast<uintptr_t>(a.C) & 15) || (reinterpret_cast<uintptr_t>(a.A) & 15)) return false;
  return true;
}

// grid of a problem: `full` row blocks with all of N each, the remaining `tail` blocks shared by `slices` workgroups each
static void rs2_grid(int M, int npairs, int* full, int* slices, int* grid) {
  const int wgs = (M + 255) / 256, rem = wgs % 256;
  int f = wgs, sl = 1;
  if (rem != 0 && rem * 100 < 85 * 256) {                      // last round under-filled: slice it
    f = wgs - rem;
    sl = 256 / rem;
    if (sl > npairs) sl = npairs;
    if (sl < 1) sl = 1;
  }
  *full = f; *slices = sl; *grid = f + (wgs - f) * sl;
}

static int g_rs2_dbg = 0;
void set_gemm_rs2_dbg(int v) { g_rs2_dbg = v; }

template <typename T, bool G, bool L, int ABL, bool KVS = false>
static int launch_rs2_one(const GemmArgs& a, dim3 grid, hipStream_t st) {
  static bool attr_done[64] = {};    // per device: the opt-in to > 64 KB of dynamic LDS belongs to the device's function
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -19;
  if (!attr_done[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rs2_kernel<T, G, L, ABL, KVS>), hipFuncAttributeMaxDynamicSharedMemorySize, R2_LDS) != hipSuccess)
      return -12;
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL((gemm_rs2_kernel<T, G, L, ABL, KVS>), grid, dim3(512), R2_LDS, st, a);
  return 0;
}

template <typename T>
int launch_gemm_rs2(const GemmArgs& a0, bool geglu, hipStream_t st) {
  GemmArgs a = a0;
  const bool lnf = a.ln_colsum != nullptr;
  int full, slices, nwg;
  rs2_grid(a.M, (a.N + (geglu ? 31 : 63)) / (geglu ? 32 : 64), &full, &slices, &nwg);
  a.tiles_m = full; a.tiles_n = slices;
  dim3 grid(nwg);
  int rc = 0;
#ifdef HALLO_ABLATIONS
  if (g_rs2_dbg && lnf) {          // timing ablations (wrong results) exist for the LayerNorm-fused forms only
    const int abl = g_rs2_dbg & 7;
    if (geglu) {
      if (abl == 1) rc = launch_rs2_one<T, true, true, 1>(a, grid, st);
      else if (abl == 5) rc = launch_rs2_one<T, true, true, 5>(a, grid, st);
      else rc = launch_rs2_one<T, true, true, 2>(a, grid, st);
    }
    else { if (abl == 1) rc = launch_rs2_one<T, false, true, 1>(a, grid, st); else rc = launch_rs2_one<T, false, true, 2>(a, grid, st); }
  } else
#endif
  if (a.kv_out) { if (geglu || !lnf) return -22; rc = launch_rs2_one<T, false, true, 0, true>(a, grid, st); }
  else if (geglu) { if (lnf) rc = launch_rs2_one<T, true, true, 0>(a, grid, st); else rc = launch_rs2_one<T, true, false, 0>(a, grid, st); }
  else { if (lnf) rc = launch_rs2_one<T, false, true, 0>(a, grid, st); else rc = launch_rs2_one<T, false, false, 0>(a, grid, st); }
  if (rc) return rc;
  HALLO_CHECK_LAUNCH();
  return 0;
}

template int launch_gemm_rs2<_Float16>(const GemmArgs&, bool, hipStream_t);
template int launch_gemm_rs2<__bf16>(const GemmArgs&, bool, hipStream_t);

}  // namespace hallo
