// Fused feed-forward for C = 320: y = res + Linear(4C -> C)(GEGLU(Linear(C -> 8C)(LayerNorm(x)))) in ONE kernel.
//
// Replaces the pair of hallo_gemm launches behind diffusers FeedForward(activation_fn="geglu") at the 64 x 64-latent level
// (hallo/models/attention.py:601,905 -- `ff(norm3(x)) + x` of the spatial and audio transformer blocks;
// hallo/models/motion_module.py:420 -- `ff(ff_norm(x)) + x`): GEGLU into a [rows, 1280] intermediate (168 MB at 65536 rows),
// then net[2] + residual reading it back: 128 + 84 us per pair, 375 pairs per clip (profiles/r2_shape_breakdown.json).
//
// Row-stationary, back to back: a wave owns 32 rows for the whole kernel.
//   * Their 20 MFMA B-fragments of LayerNorm(x) live in 80 registers (as in gemm_rs2.hip: x read once, statistics by packed
//     dot products, (x - mean) * rstd rounded to the storage type like nn.LayerNorm's output; gamma / beta folded into W1 / b1
//     by the caller).
//   * A step = 16 columns of the 1280-wide intermediate.  ONE 32-row W1 block holds the 16 value rows and the 16 gate rows of
//     those columns: after its 20 MFMAs (K = 320) a lane holds value (accumulator registers 0..7) and gate (8..15) of the SAME
//     8 columns of its row -- GEGLU is lane-local, and the 8 products, packed to the storage type, ARE the B operand of the
//     second GEMM's MFMA for that 16-column K slice (the k-slot order of the accumulator layout is baked into the W2 image,
//     the flash-attention P -> PV trick).  10 MFMAs add W2[:, 16 columns] . H^T into the 32 x 320 fp32 output accumulator
//     (160 registers per lane).  The intermediate never exists in memory: HBM traffic is x in, y out, the residual.
//   * One wave per SIMD (4 waves = 128 rows per workgroup, 80 + 160 + 32 accumulator / fragment registers need the 512-entry
//     file).  The VALU work of a step (8 erf-GELUs per lane) is interleaved, element by element, with the NEXT step's GEGLU
//     MFMAs (sched_group_barrier regions as in gemm_rs2.hip); the second GEMM of step j-1 follows them.
//   * Weights: the caller packs [W1 | W2 | b1] per step into one 32 KB image (hallo_ff320_pack; lane-linear for LDS-DMA, bank
//     swizzles applied at packing time), streamed through a 4-slot LDS ring by `buffer_load ... lds`: 8 instructions per
//     wave per step, two steps in flight behind a counted vmcnt, one barrier per step.  Every workgroup streams all 2.6 MB
//     (L2-resident) past its rows.
#include "gemm_args.h"
#include <type_traits>

namespace hallo {

namespace {
constexpr int FF_C = 320, FF_K16 = 20, FF_INNER = 1280, FF_CH = 16;
constexpr int FF_NSTEP = FF_INNER / FF_CH;                    // 80
constexpr int FF_W1_BYTES = 32 * FF_C * 2;                    // 20480: 5 sub-tiles [32 rows][64 k], 128-byte rows, XOR-swizzled
constexpr int FF_W2_BYTES = FF_C * FF_CH * 2;                 // 10240: [320 n][16 k-slots], 32-byte rows, halves swizzled
constexpr int FF_B1_OFF = FF_W1_BYTES + FF_W2_BYTES;          // 30720: fp32 [2 halves][16] scaled value / gate biases
constexpr int FF_CHUNK = 32768;
constexpr int FF_RING = 4;
constexpr int FF_LDS = FF_RING * FF_CHUNK;                    // 131072
constexpr int FF_DMA = FF_CHUNK / 1024 / 4;                   // 8 LDS-DMA instructions per wave per step
}  // namespace

struct FfArgs {
  const void* x; long ldx;          // [M, 320] input of the (folded) LayerNorm
  const void* res; long ldr;        // [M, 320] residual (usually x)
  void* y; long ldy;                // [M, 320]
  const void* wpack;                // hallo_ff320_pack image: FF_NSTEP x 32 KB
  const void* b2;                   // [320] storage type
  int M;
  float ln_eps;
};

template <int B, int E, typename F>
__device__ __forceinline__ void ff_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    ff_static_for<B + 1, E>(f);
  }
}

// LNF: apply LayerNorm to the resident rows; OVL: interleave the GEGLU arithmetic of step j-1 with the MFMAs of step j
template <typename T, bool LNF, bool OVL>
__global__ __launch_bounds__(256) void ff320_kernel(const FfArgs p) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* const lds = (lds_u8*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.x * 128 + wave_u * 32;
  const int m = m0 + l31;
  const int mc = min(m, p.M - 1);
  const bool row_live = m < p.M;

  // ---- weight stream: step s = bytes [s * 32 KB, +32 KB) of the packed image, lane-linear ----
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wpack), 0, FF_NSTEP * FF_CHUNK, 0x00020000);
  const int w_voff = lane * 16;
  int issued = 0;
  auto issue_next = [&]() {
    const int dst = (issued % FF_RING) * FF_CHUNK + wave_u * 1024;
    const int src = issued * FF_CHUNK + wave_u * 1024;
#pragma unroll
    for (int i = 0; i < FF_DMA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(lds + dst + i * 4096), 16, w_voff,
                                               src + i * 4096, 0, 0);
    ++issued;
  };
  issue_next();
  issue_next();

  // ---- resident rows: B operand fragments of LayerNorm(x): lane holds x[m][k16 * 16 + hi * 8 .. +8] ----
  V8 af[FF_K16];
  {
    const T* xrow = reinterpret_cast<const T*>(p.x) + (long)mc * p.ldx + hi * 8;
#pragma unroll
    for (int k = 0; k < FF_K16; ++k) af[k] = ld8<T>(xrow + k * 16);
  }
  if (LNF) {
    typedef __attribute__((ext_vector_type(2))) T V2t;
    const V2t one2 = {from_f32<T>(1.0f), from_f32<T>(1.0f)};
    float sm = 0.0f, sq = 0.0f;
#pragma unroll
    for (int k = 0; k < FF_K16; ++k)
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const V2t x2 = {af[k][e], af[k][e + 1]};
        sm = dot2(x2, one2, sm);
        sq = dot2(x2, x2, sq);
      }
    sm += __shfl_xor(sm, 32, 64);
    sq += __shfl_xor(sq, 32, 64);
    const float mean = sm * (1.0f / FF_C);
    const float rstd = rsqrtf(fmaxf(sq * (1.0f / FF_C) - mean * mean, 0.0f) + p.ln_eps);
    const float shift = -mean * rstd;
#pragma unroll
    for (int k = 0; k < FF_K16; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) af[k][e] = from_f32<T>(__builtin_fmaf(to_f32(af[k][e]), rstd, shift));
  }

  // ---- fragment addresses (per lane, fixed): W1 sub-tile k / 4, row l31, 16-byte piece ((k % 4) * 2 + hi) ^ ((l31 >> 1) & 7);
  //      W2 row blk * 32 + l31, half hi ^ ((l31 >> 3) & 1) ----
  const int xsw = hi ^ ((l31 >> 1) & 7);
  int fo[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fo[kk] = l31 * 128 + ((kk * 2) ^ xsw) * 16;
  const int w2o = FF_W1_BYTES + l31 * 32 + ((hi ^ ((l31 >> 3) & 1)) * 16);
  const int b1o = FF_B1_OFF + hi * 64;

  const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  f32x16 yacc[10];
#pragma unroll
  for (int b = 0; b < 10; ++b) yacc[b] = zero16;

  typedef const __attribute__((address_space(3))) V8* ldsv8;
  typedef const __attribute__((address_space(3))) f32x4* ldsf4;

  // GEGLU of one accumulator element pair: value register E, gate register E + 8 (common.h gelu_u: the scales ride in b1)
  f32x4 cv0, cv1, cg0, cg1;
  float hf[8];
  auto geglu_elem = [&](auto e_c, const f32x16& acc) {
    constexpr int E = decltype(e_c)::value;
    const float bv = E < 4 ? cv0[E & 3] : cv1[E & 3];
    const float bg = E < 4 ? cg0[E & 3] : cg1[E & 3];
    const float hv = __builtin_fmaf(GELU_U_INV, acc[E], bv);
    const float gu = __builtin_fmaf(GELU_U_SCALE, acc[E + 8], bg);
    hf[E] = hv * gelu_u(gu);
  };
  auto pack_h = [&]() {
    V8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(hf[e]);
    return v;
  };

  // ---- one step: W1 block of step `s` -> acc_cur (20 MFMAs), with the GEGLU arithmetic of the previous step's accumulator in
  // between (OVL); then the previous step's 8 products against its W2 slice (10 MFMAs into yacc) ----
  auto step = [&](auto prev_c, auto cur_c, const int s, f32x16& acc_cur, const f32x16& acc_prev) {
    constexpr bool PREV = decltype(prev_c)::value;      // a step s - 1 exists: finish it
    constexpr bool CUR = decltype(cur_c)::value;        // a step s exists (false: drain only)
    if (CUR) {
      // chunk s has landed when at most the 8 DMAs of chunk s + 1 are younger
      if (issued > s + 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();      // chunk s visible to every wave; every wave is done with chunk s - 2 (slot of chunk s + 2)
      if (issued < FF_NSTEP) issue_next();
    }
    const unsigned slot = (unsigned)((s % FF_RING) * FF_CHUNK);
    const unsigned pslot = (unsigned)(((s + FF_RING - 1) % FF_RING) * FF_CHUNK);
    const lds_u8* const sw0 = lds + ((slot + (unsigned)fo[0]) & 0x3FFF0u);
    const lds_u8* const sw1 = lds + ((slot + (unsigned)fo[1]) & 0x3FFF0u);
    const lds_u8* const sw2 = lds + ((slot + (unsigned)fo[2]) & 0x3FFF0u);
    const lds_u8* const sw3 = lds + ((slot + (unsigned)fo[3]) & 0x3FFF0u);
    const lds_u8* const pw2 = lds + ((pslot + (unsigned)w2o) & 0x3FFF0u);
    if (PREV) {
      const lds_u8* const pb = lds + ((pslot + (unsigned)b1o) & 0x3FFF0u);
      cv0 = *(ldsf4)(pb); cv1 = *(ldsf4)(pb + 16); cg0 = *(ldsf4)(pb + 32); cg1 = *(ldsf4)(pb + 48);
    }
    if (CUR) {
      V8 fr[3];
#define FF_LOADW(set, k) fr[set] = *(ldsv8)((((k) & 3) == 0 ? sw0 : ((k) & 3) == 1 ? sw1 : ((k) & 3) == 2 ? sw2 : sw3) + ((k) >> 2) * 4096)
      FF_LOADW(0, 0);
      FF_LOADW(1, 1);
      ff_static_for<0, FF_K16 / 2>([&](auto rc) {
        constexpr int R = decltype(rc)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int k = 2 * R + kk;
          if (k + 2 < FF_K16) { FF_LOADW((k + 2) % 3, k + 2); }
          acc_cur = Vec<T>::mfma32(fr[k % 3], af[k], k == 0 ? zero16 : acc_cur);
        }
        if constexpr (PREV && OVL && R < 8) {
          geglu_elem(std::integral_constant<int, R>{}, acc_prev);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
#undef FF_LOADW
    }
    if (PREV) {
      if (!(OVL && CUR)) ff_static_for<0, 8>([&](auto ec) { geglu_elem(ec, acc_prev); });
      const V8 pf = pack_h();
      V8 w2f[2];
      w2f[0] = *(ldsv8)(pw2);
#pragma unroll
      for (int b = 0; b < 10; ++b) {
        if (b + 1 < 10) w2f[(b + 1) & 1] = *(ldsv8)(pw2 + (b + 1) * 1024);
        yacc[b] = Vec<T>::mfma32(w2f[b & 1], pf, yacc[b]);
      }
    }
  };

  f32x16 acc_a, acc_b;
  {
    using TT = std::true_type;
    using FF = std::false_type;
    step(FF{}, TT{}, 0, acc_a, acc_b);
    int s = 1;
    for (; s + 1 < FF_NSTEP; s += 2) {          // two steps per trip: the accumulators alternate by name
      step(TT{}, TT{}, s, acc_b, acc_a);
      step(TT{}, TT{}, s + 1, acc_a, acc_b);
    }
    step(TT{}, TT{}, FF_NSTEP - 1, acc_b, acc_a);      // FF_NSTEP is even: the last step lands in acc_b
    step(TT{}, FF{}, FF_NSTEP, acc_a, acc_b);          // drain: GEGLU + second GEMM of the last step
  }

  // ---- epilogue: y = yacc + b2 + res; lane owns row m, columns blk * 32 + 8 g + 4 hi + j; pairs of 4-column groups are merged
  // with a half swap so that every lane stores 16 contiguous bytes ----
  {
    const T* rrow = reinterpret_cast<const T*>(p.res) + (long)mc * p.ldr;
    const T* b2 = reinterpret_cast<const T*>(p.b2);
    T* yrow = reinterpret_cast<T*>(p.y) + (long)mc * p.ldy;
#pragma unroll
    for (int b = 0; b < 10; ++b) {
      unsigned pk[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = b * 32 + 8 * g + 4 * hi;
        const V4 r4 = *reinterpret_cast<const V4*>(rrow + n);
        const V4 c4 = *reinterpret_cast<const V4*>(b2 + n);
        V4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = from_f32<T>(yacc[b][g * 4 + j] + to_f32(c4[j]) + to_f32(r4[j]));
        const uint2 u = __builtin_bit_cast(uint2, o);
        pk[g][0] = u.x; pk[g][1] = u.y;
      }
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        const auto x = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
        const auto y = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
        if (row_live) *reinterpret_cast<uint4*>(yrow + b * 32 + 8 * g + 8 * hi) = make_uint4(x[0], y[0], x[1], y[1]);
      }
    }
  }
}

static int g_ff_variant = 1;          // hallo_set_option("ff_fused", 0 | 1 | 2): 0 = callers use the two-GEMM path, 1 = overlapped, 2 = serial (A/B)
int ff_fused_variant() { return g_ff_variant; }
void set_ff_fused_variant(int v) { g_ff_variant = v; }

template <typename T, bool LNF, bool OVL>
static int launch_ff_one(const FfArgs& a, hipStream_t st) {
  static bool attr_done[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -19;
  if (!attr_done[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ff320_kernel<T, LNF, OVL>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS) != hipSuccess)
      return -12;
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL((ff320_kernel<T, LNF, OVL>), dim3((a.M + 127) / 128), dim3(256), FF_LDS, st, a);
  return 0;
}

template <typename T>
static int launch_ff(const FfArgs& a, bool lnf, hipStream_t st) {
  int rc;
  if (g_ff_variant == 2) rc = lnf ? launch_ff_one<T, true, false>(a, st) : launch_ff_one<T, false, false>(a, st);
  else rc = lnf ? launch_ff_one<T, true, true>(a, st) : launch_ff_one<T, false, true>(a, st);
  if (rc) return rc;
  HALLO_CHECK_LAUNCH();
  return 0;
}

}  // namespace hallo

using namespace hallo;

extern "C" int64_t hallo_ff320_pack_bytes(void) { return (int64_t)FF_NSTEP * FF_CHUNK; }

extern "C" int hallo_ff320(const void* x, int64_t ldx, const void* res, int64_t ldr, void* y, int64_t ldy, const void* wpack,
                           const void* b2, int64_t M, int layernorm, float ln_eps, int dtype, void* stream) {
  if (!x || !res || !y || !wpack || !b2 || M <= 0 || M > 0x7FFFFFFF) return -22;
  if ((ldx & 7) || (ldr & 3) || (ldy & 7) || ldx < FF_C || ldr < FF_C || ldy < FF_C) return -22;
  const auto al = [](const void* ptr, int a) { return (reinterpret_cast<uintptr_t>(ptr) & (a - 1)) == 0; };
  if (!al(x, 16) || !al(y, 16) || !al(res, 8) || !al(wpack, 16) || !al(b2, 8)) return -22;
  FfArgs a;
  a.x = x; a.ldx = ldx; a.res = res; a.ldr = ldr; a.y = y; a.ldy = ldy; a.wpack = wpack; a.b2 = b2; a.M = (int)M; a.ln_eps = ln_eps;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DT_F16) return launch_ff<_Float16>(a, layernorm != 0, st);
  if (dtype == DT_BF16) return launch_ff<__bf16>(a, layernorm != 0, st);
  return -22;
}
