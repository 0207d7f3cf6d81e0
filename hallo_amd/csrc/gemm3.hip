// Large-tile MFMA GEMM / implicit-GEMM conv3x3 for gfx950: 256x320x64 (TM = 2) or 128x320x64 (TM = 1) per
// 512-thread workgroup.
//
// Why this tile.  The 128x128 kernel of gemm.hip stages 32 KB of operands per 64-deep K step for 4 x 16 MFMAs:
// at the CU's ~64 B/clk L1->LDS fill rate that is 512 cycles of fill against 512 cycles of matrix pipe per SIMD,
// so the kernel is fill-bound near 1/3 of the MFMA peak whatever the schedule.  Here one K step stages
// (256 + 320) x 128 B = 72 KB (1152 fill cycles) for 8 waves x 40 MFMAs = 2560 matrix cycles per SIMD: the
// fill runs at 45 % duty and can hide under the MFMAs.  Every projection / feed-forward / conv width of the
// Hallo UNets is a multiple of 320 (320, 640, 960, 1280, 1920, 2560, 3840, 5120, 10240), so a 320-wide N
// tile has no edge waste, and the token counts (65536 / 16384 / 4096, x18/16 for the motion modules) are
// multiples of 256.
//
//  * 8 waves as 4 (M) x 2 (N); a wave owns TM x 5 MFMA 32x32x16 blocks = (TM*32) x 160 outputs, fp32 accumulators
//    (160 VGPRs at TM = 2), operands swapped (D = W . A^T) so a lane owns one output row.
//  * Two LDS slots of one K tile each (A [BM][64] then W [320][64], 128-B rows, XOR-swizzled through the source
//    address exactly as in gemm2_kernel).  ONE barrier per K step:
//        wait own DMA of tile k; barrier; { issue the DMA of tile k+1 into the other slot, interleaved with }
//        { the 4 x (TM + 5) fragment reads and 4 x 5 TM MFMAs of tile k };
//    the other slot was last read in step k-1, which every wave finished before it arrived at this barrier.
//  * Loader: 8-row x 128-B slabs by `buffer_load_dwordx4 ... lds`; wave w owns slabs w, w+8, ... so the swizzle
//    term is one per-lane constant; K advances through the scalar offset; rows past M / N read zeros through
//    the descriptor's range check (voffset = 0xFFFFFFFF).
//  * Epilogue: per wave, 32 x 64 fp32 sub-tiles are transposed through an 8 KB LDS slice so that bias /
//    residual loads and C stores are 16 bytes per lane, 128 contiguous bytes per row (same scheme as gemm2).
//
// Replaces the same reference calls as hallo_gemm / hallo_conv3x3_nhwc (see gemm.hip); selected by the host
// launcher for shapes whose grid fills the chip with these tiles.
#include "common.h"
#include "gemm_args.h"
#include <type_traits>

namespace hallo {

constexpr int G3_BN = 320, G3_BK = 64;
constexpr unsigned G3_OOB = 0xFFFFFFFFu;

#define G3_BLOAD(rs, ldsptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ldsptr), 16, (int)(voff), (int)(soff), 0, 0)

// (A persistent form -- one workgroup per CU walking the tile list with the next tile's first K step prefetched under the
// epilogue -- existed in rounds 1-5 behind gemm_variant 7 / 8: next to 160 accumulator registers its loader state spilled 201
// VGPRs, it never won an A/B and no routing rule selected it.  Removed in round 6.)
template <typename T, int MODE /*0 gemm, 1 conv3x3 (fast gather), 2 geglu*/, int TM>
__global__ __launch_bounds__(512, 2) void gemm3_kernel(const GemmArgs p) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  constexpr bool CONV = MODE == 1, GEGLU = MODE == 2;
  constexpr int BM = 128 * TM;
  constexpr int NA = BM / 64;            // A slabs per wave per K step
  constexpr int NW = 5;                  // W slabs per wave per K step
  constexpr int NP = NA + NW;
#ifndef G3_FRONT
#define G3_FRONT 0
#endif
  // DMA pieces issued per 16-deep k slice: spread over the four slices of a K step, or (G3_FRONT, A/B build) front-loaded
  // into the first two so that the last piece has half a step to land before the next step's vmcnt(0)
  constexpr int PPK = G3_FRONT ? (NP + 1) / 2 : (NP + 3) / 4;
  constexpr int SLOT = (BM + G3_BN) * G3_BK;
  constexpr int SLOT_STRIDE = SLOT;
  __shared__ __attribute__((aligned(16))) T smem[SLOT_STRIDE + SLOT];
  // GemmArgs::ln_parts > 0 (round 5): ln_stats holds the producer's partial (sum, sum of squares) per 64-column block of A's rows;
  // (mean, rstd) of this tile's rows are reduced here in slot order.
  __shared__ float s_ln[CONV ? 2 : 2 * BM];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;

  const int nwg = p.tiles_m * p.tiles_n;
  const long zb = blockIdx.z;

  const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + zb * p.sA;
  const T* __restrict__ B = reinterpret_cast<const T*>(p.B) + zb * p.sB;

  // ---- loader state ----
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int lrow = lane >> 3, lp = lane & 7;
  const int lc = lp ^ (((lane >> 4) + 4 * (wave & 1)) & 7);   // logical 16-B chunk this lane fetches (slab parity = wave parity)
  auto clamp32 = [](long bytes) { return (int)(bytes > 0xFFFFFFFFL ? 0xFFFFFFFFL : bytes); };

  const int nk_all = p.K / G3_BK;
  const int kt_begin = (p.splits > 1) ? blockIdx.y * p.nk_per_split : 0;
  const int kt_end = (p.splits > 1) ? min(nk_all, kt_begin + p.nk_per_split) : nk_all;

  __amdgpu_buffer_rsrc_t rsA, rsB;
  unsigned w_voff[NW];
  unsigned a_voff[NA];
  unsigned a_mask[NA];
  int tap_u = 0, ch_u = 0;     // conv K position: one tap per K tile (Cin % 64 == 0), scalars, for the tile being STAGED
  // per-tile loader state: descriptors and per-lane byte offsets of this wave's W / A slabs
  auto setup_loader = [&](int tile_m, int tile_n) {
    const int m0 = tile_m * BM;
    const int n0 = GEGLU ? tile_n * 160 : tile_n * G3_BN;
    const T* Bbase = GEGLU ? B : B + (long)n0 * p.ldb;
    const long b_rows = GEGLU ? 2L * p.N : (long)(p.N - n0);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Bbase), 0, clamp32(((b_rows - 1) * p.ldb + p.K) * 2), 0x00020000);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int r = (wave + 8 * j) * 8 + lrow;     // W tile row 0..319
      long wrow;
      bool ok;
      if (GEGLU) {
        // wave-local row q of 160: [value 0..79 | gate 64..79 | gate 0..63]  (block 2 = value 64..79 + gate 64..79)
        const int wq = r / 160, q = r - wq * 160;
        const int t = q >= 80 ? 1 : 0;
        const int col = q < 80 ? q : (q < 96 ? q - 16 : q - 96);
        const int cg = n0 + wq * 80 + col;
        ok = cg < p.N;
        wrow = (long)t * p.N + cg;
      } else {
        ok = n0 + r < p.N;
        wrow = r;
      }
      w_voff[j] = ok ? (unsigned)((wrow * p.ldb + lc * 8) * 2) : G3_OOB;
    }
    const T* Abase;
    long a_bytes;
    if (CONV) {
      const int hw = p.OH * p.OW;
      const int img0 = m0 / hw, n_img = p.M / hw;
      const long img_elems = (long)p.H * p.W * p.Cin;
      const long shift = ((long)p.pad_t * p.W + p.pad_l) * p.Cin;   // taps are addressed from (-pad_t, -pad_l)
      Abase = A + img0 * img_elems - shift;
      a_bytes = ((long)(n_img - img0) * img_elems + shift) * 2;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int m = m0 + (wave + 8 * j) * 8 + lrow;
        const int img = m / hw, rem = m - img * hw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
        const int iy = oy * p.stride - p.pad_t, ix = ox * p.stride - p.pad_l;
        const unsigned img_off = (unsigned)((img - img0) * img_elems * 2);
        a_voff[j] = (m < p.M) ? img_off + (unsigned)((((long)oy * p.stride * p.W + ox * p.stride) * p.Cin + lc * 8) * 2) : G3_OOB;
        unsigned mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int vy = iy + t / 3, vx = ix + t % 3;
          if (m < p.M && vy >= 0 && vy < p.H && vx >= 0 && vx < p.W) mk |= 1u << t;
        }
        a_mask[j] = mk;
      }
      const int k0 = kt_begin * G3_BK;
      tap_u = k0 / p.Cin;
      ch_u = k0 - tap_u * p.Cin;
    } else {
      Abase = A + (long)m0 * p.lda;
      const int rows = min(p.M - m0, BM);
      a_bytes = ((long)(rows - 1) * p.lda + p.K) * 2;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int r = (wave + 8 * j) * 8 + lrow;
        a_voff[j] = (r < rows) ? (unsigned)(((long)r * p.lda + lc * 8) * 2) : G3_OOB;
        a_mask[j] = 0;
      }
    }
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Abase), 0, clamp32(a_bytes), 0x00020000);
  };
  auto slot_of = [&](int kt) { return kt & 1; };

  // DMA piece i of K tile kt into slot `buf`: i < NW -> W slab i, else A slab i - NW
  int soffW = 0, soffA = 0;
  unsigned tapbit = 0;
  auto stage_begin = [&](int kt) {
    soffW = kt * G3_BK * 2;
    if (CONV) {
      const int ky = tap_u / 3, kx = tap_u - ky * 3;
      soffA = ((ky * p.W + kx) * p.Cin + ch_u) * 2;
      tapbit = 1u << tap_u;
      ch_u += G3_BK;
      if (ch_u >= p.Cin) { ch_u = 0; ++tap_u; }
    } else {
      soffA = soffW;
    }
  };
  auto stage_piece = [&](int i, int buf) {
    T* dA = smem + buf * SLOT_STRIDE;
    T* dW = dA + BM * G3_BK;
    if (i < NW) {
      G3_BLOAD(rsB, dW + (wave_u + 8 * i) * 512, w_voff[i], soffW);
    } else {
      const int j = i - NW;
      if (CONV) G3_BLOAD(rsA, dA + (wave_u + 8 * j) * 512, (a_mask[j] & tapbit) ? a_voff[j] : G3_OOB, soffA);
      else G3_BLOAD(rsA, dA + (wave_u + 8 * j) * 512, a_voff[j], soffA);
    }
  };

  // fragment read offsets: row = base + l31, logical chunk ks*2 + hi -> physical (ks*2) ^ (hi ^ ((l31>>1)&7))
  const int xsw = hi ^ ((l31 >> 1) & 7);
  const int fa_row = (wm * 32 * TM + l31) * G3_BK;
  const int fw_row = (wn * 160 + l31) * G3_BK;

  const int vt = (int)blockIdx.x;
  if (vt >= nwg) return;
  const int bid = xcd_remap(vt, nwg);
  const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = GEGLU ? tile_n * 160 : tile_n * G3_BN;

  f32x16 acc[NW][TM];   // [tn][tm]
#pragma unroll
  for (int i = 0; i < NW; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  setup_loader(tile_m, tile_n);
  if (kt_begin < kt_end) {
    stage_begin(kt_begin);
#pragma unroll
    for (int i = 0; i < NP; ++i) stage_piece(i, slot_of(kt_begin));
  }
  // One K step.  MORE (compile-time) = a next tile exists and its DMA is issued from inside this step; the steady-state
  // loop body is branch-free so that the scheduler can run fragment reads of slice ks+1 under the MFMAs of slice ks.
  auto kstep = [&](auto more_c, int kt) {
    constexpr bool MORE = decltype(more_c)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of tile kt has landed
    __builtin_amdgcn_s_barrier();                      // ... everyone's has, and everyone is done reading the other slot
    const int buf = slot_of(kt);
    if (MORE) stage_begin(kt + 1);
    const T* sA = smem + buf * SLOT_STRIDE;
    const T* sW = sA + BM * G3_BK;
    // explicit software pipeline over the four 16-deep slices: fragments of slice ks+1 are read while the MFMAs of
    // slice ks run
    V8 fa[2][TM], fw[2][NW];
    auto frag = [&](int ks, V8* a, V8* w) {
      const int co = ((ks * 2) ^ xsw) * 8;
#pragma unroll
      for (int j = 0; j < TM; ++j) a[j] = ld8<T>(sA + fa_row + j * 32 * G3_BK + co);
#pragma unroll
      for (int i = 0; i < NW; ++i) w[i] = ld8<T>(sW + fw_row + i * 32 * G3_BK + co);
    };
    frag(0, fa[0], fw[0]);
#pragma unroll
    for (int ks = 0; ks < G3_BK / 16; ++ks) {
      if (ks + 1 < G3_BK / 16) frag(ks + 1, fa[(ks + 1) & 1], fw[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch reads ABOVE this slice's MFMAs (hipcc would sink them to their uses)
      constexpr int NMF = NW * TM;
      const int np = MORE ? (NP - ks * PPK < PPK ? (NP - ks * PPK > 0 ? NP - ks * PPK : 0) : PPK) : 0;
      if (MORE) {
#pragma unroll
        for (int i = ks * PPK; i < (ks + 1) * PPK && i < NP; ++i) stage_piece(i, buf ^ 1);
      }
#pragma unroll
      for (int i = 0; i < NW; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = Vec<T>::mfma32(fw[ks & 1][i], fa[ks & 1][j], acc[i][j]);
      // spread this slice's DMA issues between its MFMAs (an LDS-DMA issue costs ~60-100 cycles of this wave's
      // issue slot; behind an MFMA it is free)
      if (np >= 4) {
        // np pieces between NMF MFMAs: MFMA groups of NMF / (np + 1)
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          if (q < np) { __builtin_amdgcn_sched_group_barrier(0x008, NMF / (PPK + 1) > 0 ? NMF / (PPK + 1) : 1, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
      } else if (np == 3) {
        __builtin_amdgcn_sched_group_barrier(0x008, NMF / 4, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMF / 4, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMF / 4, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - 3 * (NMF / 4), 0);
      } else if (np == 2) {
        __builtin_amdgcn_sched_group_barrier(0x008, NMF / 3, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMF / 3, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - 2 * (NMF / 3), 0);
      } else if (np == 1) {
        __builtin_amdgcn_sched_group_barrier(0x008, NMF / 2, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - NMF / 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragment reads retired before the next barrier releases this slot
  };
  for (int kt = kt_begin; kt + 1 < kt_end; ++kt) kstep(std::true_type{}, kt);
  if (kt_begin < kt_end) kstep(std::false_type{}, kt_end - 1);
  __builtin_amdgcn_s_barrier();   // every wave is done with the staging LDS: the epilogue reuses it
  if (!CONV && p.ln_parts > 0 && p.ln_colsum != nullptr) {
    // the producer's partial (sum, sum of squares) per 64-column block of this tile's A rows -> (mean, rstd) per row: coalesced
    // 16-byte loads into slot 0 (idle), then thread r reduces row r in slot order
    const int P2 = p.ln_parts * 2;
    const int rows = min(BM, p.M - m0);
    const int nflt = rows * P2, nvec = nflt >> 2;           // whole 16-byte pieces, then the (<= 3 floats) tail: nothing is read past the rows' partials
    const f32x4* src = reinterpret_cast<const f32x4*>(p.ln_stats + (long)m0 * P2);
    f32x4* stg = reinterpret_cast<f32x4*>(smem);
    for (int i = tid; i < nvec; i += 512) stg[i] = src[i];
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
    __syncthreads();
  }
  // ---- epilogue ----
  const bool lnf = !CONV && p.ln_colsum != nullptr && p.ln_stats != nullptr && p.splits <= 1;
  const T* bias = reinterpret_cast<const T*>(p.bias);
  const T* bias2 = reinterpret_cast<const T*>(p.bias2);
  const T* res = p.residual ? reinterpret_cast<const T*>(p.residual) + zb * p.sR : nullptr;
  T* C = reinterpret_cast<T*>(p.C) + zb * p.sC;
  float* Cf = reinterpret_cast<float*>(p.C) + zb * p.sC;
  float* tile = reinterpret_cast<float*>(smem) + wave * (32 * 64);   // [32 rows][16 chunks of 4 fp32], 8 KB per wave

  auto ld8f = [&](const T* ptr, bool aligned, float* o) {
    if (aligned) {
      V8 v = ld8<T>(ptr);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = to_f32(v[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = to_f32(ptr[j]);
    }
  };
  const int rr0 = lane >> 3, rc = lane & 7;
  auto put = [&](const f32x16& v, int slot) {   // acc block -> tile[row = l31][cols slot*32 + 8g + 4hi + 0..3]
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int chunk = (slot * 8 + 2 * g + hi) ^ (l31 & 15);
      *reinterpret_cast<f32x4*>(tile + l31 * 64 + chunk * 4) = f32x4{v[g * 4], v[g * 4 + 1], v[g * 4 + 2], v[g * 4 + 3]};
    }
  };

  if (GEGLU) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int ps = 0; ps < 3; ++ps) {
        // pass 0 / 1: value block ps, gate block ps + 3 (32 columns); pass 2: block 2 holds value 64..79 | gate 64..79
        put(acc[ps][tm], 0);
        if (ps < 2) put(acc[ps + 3][tm], 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int gch = ps < 2 ? 8 + rc : 4 + rc;
        const bool lane_on = ps < 2 || rc < 4;
        const int n = n0 + wn * 80 + ps * 32 + rc * 4;
        // value / gate = [rstd * (acc - mean * G[n])] + bias; the gate argument is scaled by s and the value by 1 / s for
        // gelu_u (common.h): the factors ride in the constants
        float bh[4] = {0, 0, 0, 0}, bg[4] = {0, 0, 0, 0}, gh[4] = {0, 0, 0, 0}, gg[4] = {0, 0, 0, 0};
        if (lane_on && n < p.N) {
          if (bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { bh[j] = GELU_U_INV * to_f32(bias[n + j]); bg[j] = GELU_U_SCALE * to_f32(bias[p.N + n + j]); }
          }
          if (lnf) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.ln_colsum + n), b = *reinterpret_cast<const f32x4*>(p.ln_colsum + p.N + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) { gh[j] = a[j]; gg[j] = b[j]; }
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rr = rr0 + 8 * i;
          const int m = m0 + wm * 32 * TM + tm * 32 + rr;
          const f32x4 hv = *reinterpret_cast<const f32x4*>(tile + rr * 64 + ((rc ^ (rr & 15)) * 4));
          const f32x4 gv = *reinterpret_cast<const f32x4*>(tile + rr * 64 + ((gch ^ (rr & 15)) * 4));
          if (lane_on && m < p.M && n < p.N) {
            float vs = GELU_U_INV, gs = GELU_U_SCALE, vc = 0.0f, gc = 0.0f;      // out = scale * acc + shift * G[n] + bias'
            if (lnf) {
              const int lr = 2 * (wm * 32 * TM + tm * 32 + rr);
              const float mean = p.ln_parts > 0 ? s_ln[lr] : p.ln_stats[2 * (long)m];
              const float rstd = p.ln_parts > 0 ? s_ln[lr + 1] : p.ln_stats[2 * (long)m + 1];
              vs *= rstd; gs *= rstd; vc = -mean * vs; gc = -mean * gs;
            }
            V4 o;
#pragma unroll
            for (int j = 0; j < 4; j += 2) {      // two columns per issue on the packed fp32 pipe (common.h gelu_u2)
              const f32x2 hvv = pk_fma(f32x2{vs, vs}, f32x2{hv[j], hv[j + 1]}, pk_fma(f32x2{vc, vc}, f32x2{gh[j], gh[j + 1]}, f32x2{bh[j], bh[j + 1]}));
              const f32x2 gu = pk_fma(f32x2{gs, gs}, f32x2{gv[j], gv[j + 1]}, pk_fma(f32x2{gc, gc}, f32x2{gg[j], gg[j + 1]}, f32x2{bg[j], bg[j + 1]}));
              const f32x2 h = hvv * gelu_u2(gu);
              o[j] = from_f32<T>(h[0]);
              o[j + 1] = from_f32<T>(h[1]);
            }
            *reinterpret_cast<V4*>(C + (long)m * p.ldc + n) = o;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  } else {
  const bool use_res = res && p.res_vec_ok && p.splits <= 1;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
      // passes: n blocks (0,1), (2,3), (4)
      const bool lane_on = ps < 2 || rc < 4;
      const int n = n0 + wn * 160 + ps * 64 + rc * 8;
      const bool n_ok = lane_on && n < p.N;
      V8 rpre[4];
      if (use_res) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + wm * 32 * TM + tm * 32 + rr0 + 8 * i;
          rpre[i] = (m < p.M && n_ok) ? ld8<T>(res + (long)m * p.ldr + n) : zero8<T>();
        }
      }
      float bcol[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (bias && !p.bias_per_row && n_ok && p.splits <= 1) ld8f(bias + n, p.bias_vec_ok, bcol);
      float gcol[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (lnf && n_ok) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.ln_colsum + n), b = *reinterpret_cast<const f32x4*>(p.ln_colsum + n + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { gcol[j] = a[j]; gcol[4 + j] = b[j]; }
      }
      put(acc[2 * ps][tm], 0);
      if (ps < 2) put(acc[2 * ps + 1][tm], 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rr = rr0 + 8 * i;
        const int m = m0 + wm * 32 * TM + tm * 32 + rr;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(tile + rr * 64 + (((2 * rc) ^ (rr & 15)) * 4));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(tile + rr * 64 + (((2 * rc + 1) ^ (rr & 15)) * 4));
        if (!(n_ok && m < p.M)) continue;
        if (p.splits > 1) {
          float* sp = p.slab + ((long)blockIdx.y * p.M + m) * p.N + n;
          if (p.slab_nt) {
            __builtin_nontemporal_store(v0, reinterpret_cast<f32x4*>(sp));
            __builtin_nontemporal_store(v1, reinterpret_cast<f32x4*>(sp + 4));
          } else {
            *reinterpret_cast<f32x4*>(sp) = v0;
            *reinterpret_cast<f32x4*>(sp + 4) = v1;
          }
          continue;
        }
        float o[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        float t8[8];
        const float br = (bias && p.bias_per_row) ? to_f32(bias[m]) : 0.0f;
        if (lnf) {       // fused LayerNorm: rstd * (acc - mean * G[n]) with the statistics of hallo_row_stats (or the producer's partial sums)
          const int lr = 2 * (wm * 32 * TM + tm * 32 + rr);
          const float mean = p.ln_parts > 0 ? s_ln[lr] : p.ln_stats[2 * (long)m];
          const float rstd = p.ln_parts > 0 ? s_ln[lr + 1] : p.ln_stats[2 * (long)m + 1];
          const float c1 = -mean * rstd;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = __builtin_fmaf(rstd, o[j], c1 * gcol[j]);
        }
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
        }
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
      __builtin_amdgcn_wave_barrier();   // this wave's reads are issued (LDS is in-order per wave) before the next pass's writes
    }
  }
  }   // !GEGLU
}
#undef G3_BLOAD

template <typename T>
void launch_gemm3(const GemmArgs& a, int mode, int tm, int batch, hipStream_t st) {
  const int tiles = a.tiles_m * a.tiles_n;
  dim3 block(512);
  dim3 grid(tiles, a.splits, batch);
  if (tm == 2) {
    if (mode == 2) hipLaunchKernelGGL((gemm3_kernel<T, 2, 2>), grid, block, 0, st, a);
    else if (mode == 1) hipLaunchKernelGGL((gemm3_kernel<T, 1, 2>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gemm3_kernel<T, 0, 2>), grid, block, 0, st, a);
  } else {
    if (mode == 2) hipLaunchKernelGGL((gemm3_kernel<T, 2, 1>), grid, block, 0, st, a);
    else if (mode == 1) hipLaunchKernelGGL((gemm3_kernel<T, 1, 1>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((gemm3_kernel<T, 0, 1>), grid, block, 0, st, a);
  }
}
template void launch_gemm3<_Float16>(const GemmArgs&, int, int, int, hipStream_t);
template void launch_gemm3<__bf16>(const GemmArgs&, int, int, int, hipStream_t);

}  // namespace hallo
