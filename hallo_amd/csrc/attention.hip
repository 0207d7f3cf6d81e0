// Flash-style attention for gfx950 with two key/value segments.
//
// Replaces diffusers AttnProcessor2_0 (F.scaled_dot_product_attention) as called from
//   hallo/models/mutual_self_attention.py:253-284 (spatial self-attention whose K/V is the
//     concatenation [self ; ReferenceNet bank], CFG uncond rows attending to self only),
//   hallo/models/attention.py:828-831 (audio block self-attention),
//   hallo/models/mutual_self_attention.py:296-303 (4 face tokens) and
//   hallo/models/attention.py:846-884 (3 x 32 audio tokens).
//
// Work decomposition: one 256-thread workgroup = 128 query rows of one (batch, head);
// each of the 4 waves owns 32 query rows.  K/V tiles (KVB rows) are staged global -> VGPR -> LDS
// with the next tile in flight under the current tile's MFMAs.
//
// Both contractions use the "swapped" MFMA form so that every lane owns one query row:
//   S^T[kv, q] = K[kv, :] . Q[q, :]^T      (A = K tile from LDS, B = Q fragment in registers)
//   O^T[d,  q] = V^T[d, :] . P^T[:, q]     (A = V^T tile from LDS, B = P in registers)
// so the online-softmax statistics (row max / row sum) and the O rescale are lane-local, with a
// single cross-lane exchange (lane ^ 32) for the row max.  P never leaves registers: the k-slot
// order of the second MFMA is chosen to match the accumulator layout of the first
// (kv = 16*h2 + (j&3) + 8*(j>>2) + 4*hi for slot j of lane half hi), and the V^T fragment is read
// from LDS with the same permutation (two 8-byte reads per fragment).
#include "common.h"
#include "attn_args.h"
#include <stdlib.h>
#include "../../include/hallo_amd.h"
#include <type_traits>
#include <string.h>

namespace hallo {


// PRE = q already carries scale * log2(e) (hallo_gemm lead_alpha).  For head dim 40 the QK^T contraction is padded
// to 48: pad column 40 of every K row in LDS is 1.0 and pad element 40 of the lane's Q row holds -m_run (kept exactly
// representable in T), so the scores leave the MFMA as s * c - m_run and the per-score VALU work is exp2 + convert
// only -- no v_fma / v_sub per score.  On the head-dim-40 shapes the kernel is VALU-bound (PMC: VALU busy 68 %,
// MFMA 40 %; a plain VALU op costs 4 cycles and v_exp_f32 8 cycles per wave64), so this removes ~17 % of the VALU
// cycles per tile.  Other head dims (no pad column) subtract m_run on the VALU as before.
template <typename T, int HD, bool PRE>
__global__ __launch_bounds__(256, HD <= 40 ? 3 : (HD <= 80 ? 2 : 1)) void attn_kernel(const AttnArgs p) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  constexpr int KVB = (HD > 80) ? 32 : 64;       // kv rows per tile
  constexpr int HDP = ((HD + 15) / 16) * 16;     // head dim padded to the MFMA K step
  constexpr int NKS = HDP / 16;                  // QK^T k-steps
  constexpr int NDB = (HD + 31) / 32;            // 32-wide d blocks of O^T
  constexpr int NT = KVB / 32;                   // S^T row tiles per kv tile
  constexpr int NCH = HD / 8;                    // 16-byte chunks per K/V row
  constexpr int K_LD = HDP + 8;                  // LDS row pitch of K (elements)
  constexpr int VT_LD = KVB + 4;                 // LDS row pitch of V^T (elements)
  constexpr int KU = (KVB * NCH + 255) / 256;    // K loader units per thread
  constexpr int VU = ((KVB / 2) * NCH + 255) / 256;  // V loader units (kv pairs) per thread
  // Row sums for free: when the padded O^T block has a spare row (hd 40 -> 64 rows, 80 -> 96), V^T row HD is set to
  // all ones, so the PV MFMA accumulates sum_kv P into O^T[HD][q] -- rescaled together with O, no VALU adds.
  constexpr bool ONES_ROW = (NDB * 32 > HD);
  constexpr bool PADM = PRE && (HDP > HD);       // -m_run rides in pad column HD of the QK^T contraction
  constexpr int L_DB = HD / 32, L_R = ((HD % 32) / 8) * 4 + (HD % 4);   // accumulator slot of row HD (lanes hi = 0); HD % 8 == 0

  // two LDS stages for K and V^T: tile it+1 is written into the other stage while tile it is consumed, so the
  // K/V loop needs ONE barrier per tile (global loads of tile it+1 are in flight under the MFMAs of tile it)
  constexpr int SK_ELEMS = KVB * K_LD, SVT_ELEMS = NDB * 32 * VT_LD;
  __shared__ __attribute__((aligned(16))) T smem_kv[2 * (SK_ELEMS + SVT_ELEMS)];
  T* const sK0 = smem_kv;
  T* const sVT0 = smem_kv + 2 * SK_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;

  // block id -> (batch, head, q block), XCD-contiguous so a (batch, head)'s K/V stays in one L2
  const int nwg = p.batch * p.heads * p.nqb;
  int bid = xcd_remap(blockIdx.x, nwg);
  int qb, h;
  if (p.head_fastest) { h = bid % p.heads; bid /= p.heads; qb = bid % p.nqb; bid /= p.nqb; }
  else { qb = bid % p.nqb; bid /= p.nqb; h = bid % p.heads; bid /= p.heads; }
  const int b = bid;

  const T* __restrict__ Qg = reinterpret_cast<const T*>(p.q) + (long)b * p.q_bs + h * HD;
  const T* __restrict__ K1 = reinterpret_cast<const T*>(p.k1) + (long)b * p.k1_bs + h * HD;
  const T* __restrict__ V1 = reinterpret_cast<const T*>(p.v1) + (long)b * p.v1_bs + h * HD;
  const bool use2 = p.k2 != nullptr && p.Lkv2 > 0 && b >= p.kv2_first;
  const int b2 = use2 ? (p.kv2_mod > 0 ? (b / p.kv2_div) % p.kv2_mod : b / p.kv2_div) : 0;
  const T* __restrict__ K2 = use2 ? reinterpret_cast<const T*>(p.k2) + (long)b2 * p.k2_bs + h * HD : K1;
  const T* __restrict__ V2 = use2 ? reinterpret_cast<const T*>(p.v2) + (long)b2 * p.v2_bs + h * HD : V1;
  T* __restrict__ Og = reinterpret_cast<T*>(p.o) + (long)b * p.o_bs + h * HD;

  // zero the K pad columns (HD..HDP-1) once; the loaders never touch them
  if (HDP > HD) {
    V8 padv = zero8<T>();
    if (PADM) padv[0] = from_f32<T>(1.0f);
    for (int r = tid; r < 2 * KVB; r += 256) st8<T>(&sK0[r * K_LD + NCH * 8], padv);   // both stages (contiguous)
  }

  if (ONES_ROW) {
    for (int c = tid; c < 2 * VT_LD; c += 256)
      sVT0[(c / VT_LD) * SVT_ELEMS + HD * VT_LD + (c % VT_LD)] = from_f32<T>(1.0f);
  }

  // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q0 + l31][ks*16 + hi*8 .. +8] ----
  const int q0 = qb * 128 + wave * 32;
  const int qrow = min(q0 + l31, p.Lq - 1);
  V8 qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int d = ks * 16 + hi * 8;
    qf[ks] = (d < HD) ? ld8<T>(Qg + (long)qrow * p.q_rs + d) : zero8<T>();
  }

  const int nt1 = (p.Lkv1 + KVB - 1) / KVB;
  const int nt2 = use2 ? (p.Lkv2 + KVB - 1) / KVB : 0;
  const int nt = nt1 + nt2;

  // ---- K/V loaders: buffer loads with per-thread byte offsets computed once per segment and the tile advance in the
  // scalar offset (no per-tile address arithmetic); rows past the segment end (ragged last tile only) use an
  // out-of-range offset and read zeros ----
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  constexpr unsigned OOB = 0xFFFFFFFFu;
  auto clamp32 = [](long bytes) { return (int)(bytes > 0xFFFFFFFFL ? 0xFFFFFFFFL : (bytes < 0 ? 0 : bytes)); };
  auto mk_rsrc = [&](const T* base, long rs, int L) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, clamp32(((long)(L - 1) * rs + HD) * 2), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsK1 = mk_rsrc(K1, p.k1_rs, p.Lkv1), rsV1 = mk_rsrc(V1, p.v1_rs, p.Lkv1);
  const __amdgpu_buffer_rsrc_t rsK2 = mk_rsrc(K2, use2 ? p.k2_rs : p.k1_rs, use2 ? p.Lkv2 : p.Lkv1);
  const __amdgpu_buffer_rsrc_t rsV2 = mk_rsrc(V2, use2 ? p.v2_rs : p.v1_rs, use2 ? p.Lkv2 : p.Lkv1);
  // K and V run on separate tile counters (the pipelined loop below stages K two tiles ahead and V one tile ahead),
  // so each has its own segment state.
  unsigned offK[KU], offV[VU][2];
  auto set_segment_k = [&](long krs) {
#pragma unroll
    for (int i = 0; i < KU; ++i) {
      const int u = tid + 256 * i;
      const int row = u / NCH, ch = u - row * NCH;
      offK[i] = (u < KVB * NCH) ? (unsigned)((row * krs + ch * 8) * 2) : OOB;
    }
  };
  auto set_segment_v = [&](long vrs) {
#pragma unroll
    for (int i = 0; i < VU; ++i) {
      const int u = tid + 256 * i;
      const int pr = u % (KVB / 2), ch = u / (KVB / 2);
      const bool ok = u < (KVB / 2) * NCH;
      offV[i][0] = ok ? (unsigned)((2 * pr * vrs + ch * 8) * 2) : OOB;
      offV[i][1] = ok ? (unsigned)(((2 * pr + 1) * vrs + ch * 8) * 2) : OOB;
    }
  };
  set_segment_k(p.k1_rs);
  set_segment_v(p.v1_rs);

  V8 rk[KU];
  V8 rv[VU][2];
  auto ldb = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, int soff) {
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0);
    return __builtin_bit_cast(V8, r);
  };
  auto load_k = [&](int it) {
    const bool s2 = it >= nt1;
    if (it == nt1) set_segment_k(p.k2_rs);     // wave-uniform, once per kernel
    const long krs = s2 ? p.k2_rs : p.k1_rs;
    const int L = s2 ? p.Lkv2 : p.Lkv1;
    const int kv0 = (s2 ? it - nt1 : it) * KVB;
    const int soK = (int)(kv0 * krs * 2);
    if (kv0 + KVB <= L) {
#pragma unroll
      for (int i = 0; i < KU; ++i) rk[i] = s2 ? ldb(rsK2, offK[i], soK) : ldb(rsK1, offK[i], soK);
    } else {
#pragma unroll
      for (int i = 0; i < KU; ++i) {
        const unsigned o = (kv0 + (tid + 256 * i) / NCH < L) ? offK[i] : OOB;
        rk[i] = s2 ? ldb(rsK2, o, soK) : ldb(rsK1, o, soK);
      }
    }
  };
  auto load_v = [&](int it) {
    const bool s2 = it >= nt1;
    if (it == nt1) set_segment_v(p.v2_rs);
    const long vrs = s2 ? p.v2_rs : p.v1_rs;
    const int L = s2 ? p.Lkv2 : p.Lkv1;
    const int kv0 = (s2 ? it - nt1 : it) * KVB;
    const int soV = (int)(kv0 * vrs * 2);
    if (kv0 + KVB <= L) {
#pragma unroll
      for (int i = 0; i < VU; ++i) {
        rv[i][0] = s2 ? ldb(rsV2, offV[i][0], soV) : ldb(rsV1, offV[i][0], soV);
        rv[i][1] = s2 ? ldb(rsV2, offV[i][1], soV) : ldb(rsV1, offV[i][1], soV);
      }
    } else {
#pragma unroll
      for (int i = 0; i < VU; ++i) {
        const int r0 = 2 * ((tid + 256 * i) % (KVB / 2));
        const unsigned o0 = (kv0 + r0 < L) ? offV[i][0] : OOB, o1 = (kv0 + r0 + 1 < L) ? offV[i][1] : OOB;
        rv[i][0] = s2 ? ldb(rsV2, o0, soV) : ldb(rsV1, o0, soV);
        rv[i][1] = s2 ? ldb(rsV2, o1, soV) : ldb(rsV1, o1, soV);
      }
    }
  };
  auto store_k = [&](int stage) {
    T* sK = sK0 + stage * SK_ELEMS;
#pragma unroll
    for (int i = 0; i < KU; ++i) {
      const int u = tid + 256 * i;
      if (u < KVB * NCH) {
        const int row = u / NCH, ch = u - row * NCH;
        st8<T>(&sK[row * K_LD + ch * 8], rk[i]);
      }
    }
  };
  auto store_v = [&](int stage) {
    T* sVT = sVT0 + stage * SVT_ELEMS;
#pragma unroll
    for (int i = 0; i < VU; ++i) {
      const int u = tid + 256 * i;
      if (u < (KVB / 2) * NCH) {
        const int pr = u % (KVB / 2), ch = u / (KVB / 2);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // V^T[d = ch*8 + j][kv = 2*pr, 2*pr+1] as one 4-byte store
          typedef __attribute__((ext_vector_type(2))) T V2t;
          V2t w;
          w[0] = rv[i][0][j];
          w[1] = rv[i][1][j];
          *reinterpret_cast<V2t*>(&sVT[(ch * 8 + j) * VT_LD + 2 * pr]) = w;
        }
      }
    }
  };

  f32x16 oacc[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.0f;
  float m_run = PADM ? 0.0f : -1e30f, l_run = 0.0f;
  constexpr float RESCALE_THR = 6.0f;   // log2 units: P <= 64
  const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

  // S^T = K . Q^T of one K stage
  auto qk = [&](const T* sK, f32x16* sc) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        V8 kf = ld8<T>(&sK[(t * 32 + l31) * K_LD + ks * 16 + hi * 8]);
        sc[t] = Vec<T>::mfma32(kf, qf[ks], ks == 0 ? zero16 : sc[t]);   // C = inline 0 on the first k-step
      }
    }
  };

  // ---- software pipeline over the K/V tiles ----
  // Iteration `it` runs, in ONE basic block, the QK^T MFMAs of tile it+1, the exp / convert VALU work of tile it and
  // the PV MFMAs of tile it: the matrix pipe has independent work while the softmax of the current tile is on the
  // VALU (a wave that does QK -> softmax -> PV in sequence leaves each pipe idle while it uses the other; measured
  // on this kernel: MFMA busy 43 %, VALU busy 65 %, waves parked 43 % of their cycles).
  // LDS: K is staged two tiles ahead (K(it+1) must be complete when iteration it starts), V one tile ahead; both
  // double-buffered, ONE barrier per tile:
  //   K(it+2) -> K stage it&1       (last read by QK(it) in iteration it-1)
  //   V(it+1) -> V stage (it+1)&1   (last read by PV(it-1) in iteration it-1)
  f32x16 s_a[NT], s_b[NT];     // score tiles of the current / next K tile; the roles swap every iteration
  if (nt > 0) {
    load_k(0);
    load_v(0);
    store_k(0);
    store_v(0);
    if (nt > 1) {
      load_k(1);
      store_k(1);
    }
  }
  __syncthreads();
  if (nt > 0) qk(sK0, s_a);

  auto tile_step = [&](auto more_c, int it, f32x16* s_cur, f32x16* s_nxt) {
    constexpr bool MORE = decltype(more_c)::value;      // a tile it+1 exists
    if (MORE) {
      load_v(it + 1);
      if (it + 2 < nt) load_k(it + 2);
    }
    const T* sVT = sVT0 + (it & 1) * SVT_ELEMS;

    // ---- online softmax statistics on the raw scores of tile it (lane-local row; partner lane^32 holds the other
    // kv half).  p = exp2(s * c - m) with c = scale * log2(e) folded into one FMA per element; only the ragged last
    // tile of a segment is masked.
    const bool s2 = it >= nt1;
    const int L = s2 ? p.Lkv2 : p.Lkv1;
    const int kv0 = (s2 ? it - nt1 : it) * KVB;
    if (__builtin_expect(kv0 + KVB > L, 0)) {
      asm volatile("" ::: "memory");   // keep this a real (wave-uniform) branch: if-converted it costs ~100 VALU on EVERY tile
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = kv0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          s_cur[t][r] = (kv < L) ? s_cur[t][r] : -3.0e38f;
        }
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(s_cur[t][r], s_cur[t][r + 1]), mx);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // deferred rescale (keep the old reference max while the row max grew by < 2^RESCALE_THR): the O / l
    // rescale pass is skipped for most tiles; P stays <= 2^RESCALE_THR, exact in the fp32 accumulators.
    if (PADM) {
      // s_cur holds s*c - m_run (m_run = 0 before the first tile, which always takes this branch)
      if (__builtin_expect(it == 0 || !__all(mx <= RESCALE_THR), 0)) {
        asm volatile("" ::: "memory");   // a real branch: the rescale pass runs on a handful of tiles per row block
        const float want = m_run + (it == 0 ? mx : fmaxf(mx, 0.0f));
        const float m_new = to_f32(from_f32<T>(want));          // any reference value works; it must be exact in T
        const float delta = m_new - m_run;
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        m_run = m_new;
        if (hi) qf[NKS - 1][HD - (NKS - 1) * 16 - 8] = from_f32<T>(-m_new);   // Q'[row][HD]: lanes hi = 1 hold d = HD .. HD+7
        if (!ONES_ROW) l_run *= alpha;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) s_cur[t][r] -= delta;
      }
    } else {
      if (!PRE) mx *= p.scale_log2e;
      if (__builtin_expect(!__all(mx - m_run <= RESCALE_THR), 0)) {
        asm volatile("" ::: "memory");   // a real branch: the rescale pass runs on a handful of tiles per row block
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        if (!ONES_ROW) l_run *= alpha;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
      }
    }

    // ---- one block: QK^T of tile it+1 | exp + convert of tile it | O^T += V^T . P^T of tile it ----
    if (MORE) qk(sK0 + ((it + 1) & 1) * SK_ELEMS, s_nxt);
    float psum = 0.0f;
    V8 pf[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = PADM ? __builtin_amdgcn_exp2f(s_cur[t][r])
                       : PRE ? __builtin_amdgcn_exp2f(s_cur[t][r] - m_run)
                             : __builtin_amdgcn_exp2f(__builtin_fmaf(s_cur[t][r], p.scale_log2e, -m_run));
        if (!ONES_ROW) psum += pv;
        pf[t][r >> 3][r & 7] = from_f32<T>(pv);
      }
    }
    if (!ONES_ROW) l_run += psum;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int kvl = t * 32 + h2 * 16 + hi * 4;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
          const T* vp = &sVT[(db * 32 + l31) * VT_LD + kvl];
          V4 lo = *reinterpret_cast<const V4*>(vp);
          V4 hi4 = *reinterpret_cast<const V4*>(vp + 8);
          V8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi4[j]; }
          oacc[db] = Vec<T>::mfma32(vf, pf[t][h2], oacc[db]);
        }
      }
    }

    if (MORE) {
      store_v((it + 1) & 1);
      if (it + 2 < nt) store_k(it & 1);
    }
    __syncthreads();
  };
  {
    int it = 0;
    for (; it + 2 < nt; it += 2) {     // two tiles per trip so that the current / next score registers swap by name
      tile_step(std::true_type{}, it, s_a, s_b);
      tile_step(std::true_type{}, it + 1, s_b, s_a);
    }
    if (it + 2 == nt) {
      tile_step(std::true_type{}, it, s_a, s_b);
      tile_step(std::false_type{}, it + 1, s_b, s_a);
    } else if (it + 1 == nt) {
      tile_step(std::false_type{}, it, s_a, s_b);
    }
  }

  // ---- normalise and store: lane owns row q0+l31, d = db*32 + (r&3) + 8*(r>>2) + 4*hi ----
  float l_tot;
  if (ONES_ROW) l_tot = __shfl(oacc[L_DB][L_R], l31, 64);       // row HD of O^T lives in the hi = 0 lane of column q
  else l_tot = l_run + __shfl_xor(l_run, 32, 64);
  float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
  if (p.o_rowscale && q0 + l31 < p.Lq) inv *= p.o_rowscale[(long)(h / p.rs_hdiv) * p.rs_stride + (long)b * p.Lq + q0 + l31];
  if (q0 + l31 < p.Lq) {
    T* orow = Og + (long)(q0 + l31) * p.o_rs;
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = db * 32 + 8 * g + 4 * hi;
        if (d0 < HD) {
          V4 w;
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = from_f32<T>(oacc[db][g * 4 + j] * inv);
          *reinterpret_cast<V4*>(orow + d0) = w;
        }
      }
    }
  }
}

template <typename T>
static int launch_attn(const AttnArgs& a, int hd, bool pre, hipStream_t st) {
  dim3 grid(a.batch * a.heads * a.nqb), block(256);
  if (pre) {
    switch (hd) {
      case 40: hipLaunchKernelGGL((attn_kernel<T, 40, true>), grid, block, 0, st, a); break;
      case 80: hipLaunchKernelGGL((attn_kernel<T, 80, true>), grid, block, 0, st, a); break;
      case 160: hipLaunchKernelGGL((attn_kernel<T, 160, true>), grid, block, 0, st, a); break;
      default: return -22;
    }
  } else {
    switch (hd) {
      case 40: hipLaunchKernelGGL((attn_kernel<T, 40, false>), grid, block, 0, st, a); break;
      case 80: hipLaunchKernelGGL((attn_kernel<T, 80, false>), grid, block, 0, st, a); break;
      case 160: hipLaunchKernelGGL((attn_kernel<T, 160, false>), grid, block, 0, st, a); break;
      default: return -22;
    }
  }
  HALLO_CHECK_LAUNCH();
  return 0;
}

// Frame row of temporal position f (0 <= f < F) of batch entry b in the [frames, HW, C] tensors of hallo_temporal_attention_lead:
// the first `lead` positions of every batch entry (the motion frames put in front of a clip, unet_3d_blocks.py:696-748) are
// stored together at the FRONT -- rows [b * lead, (b + 1) * lead) -- and the other F - lead positions of entry b behind all of
// them, so that the clip rows of every batch entry form one contiguous [B * (F - lead), HW, C] block.  lead = 0: rows b * F + f.
__device__ __forceinline__ long temporal_frame_row(long b, int f, int F, int lead, int B) {
  return f < lead ? b * lead + f : (long)B * lead + b * (F - lead) + (f - lead);
}

// -------------------------------------------------------------------------------------------
// Temporal (per-pixel, over frames) attention.  One workgroup = one pixel of one batch entry and
// a group of HPB heads; the F' x 3 x (HPB*hd) slab [q|k|v] of that pixel is staged in LDS with
// coalesced row-segment loads.  Scores (F' x F', F' <= 32) and the PV product run on the VALU:
// the op is HBM/layout bound (0.1 % of the step's FLOPs); the point of the kernel is
// to remove the four "(b f) d c <-> (b d) f c" transposes around it.
// -------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                            int F, int HW, int C, int hd, int hpb,
                                                            float scale_log2e, int lead, int B) {
  using V8 = typename Vec<T>::v8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int Wd = hpb * hd;                    // channels handled by this block
  const int W3 = 3 * Wd + 8;                  // frame pitch: +16 B so that the per-frame rows fall on distinct bank groups
  T* sQKV = reinterpret_cast<T*>(smem_raw);   // [F][3*Wd + 8]
  float* sP = reinterpret_cast<float*>(smem_raw + (size_t)F * W3 * sizeof(T));  // [hpb][F][F+1]

  const int tid = threadIdx.x;
  const int pix = blockIdx.x % HW, b = blockIdx.x / HW;
  const int cbase = blockIdx.y * Wd;          // first channel of this head group
  const long C3 = 3L * C;
  const int vpp = Wd / 8;                     // 16-byte vectors per part

  // all of a thread's 16-byte loads are issued before the first LDS store (one memory latency per workgroup, not one
  // per loop trip: a `load; store` loop with a runtime trip count waits for every load in turn)
  {
    constexpr int MAXU = 8;                     // F * 3 * vpp <= 32 * 3 * 20 = 1920 units -> <= 8 per thread
    const int total = F * 3 * vpp;
    V8 tmp[MAXU];
    int dst[MAXU];
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
      const int u = tid + 256 * i;
      dst[i] = -1;
      if (u < total) {
        const int f = u / (3 * vpp), rem = u - f * 3 * vpp;
        const int part = rem / vpp, v = rem - part * vpp;
        const long row = (temporal_frame_row(b, f, F, lead, B) * HW + pix);
        tmp[i] = ld8<T>(qkv + row * C3 + (long)part * C + cbase + v * 8);
        dst[i] = f * W3 + part * Wd + v * 8;
      }
    }
#pragma unroll
    for (int i = 0; i < MAXU; ++i)
      if (dst[i] >= 0) st8<T>(&sQKV[dst[i]], tmp[i]);
    for (int u = tid + 256 * MAXU; u < total; u += 256) {     // not reached for F <= 32 with <= 160 channels per block
      const int f = u / (3 * vpp), rem = u - f * 3 * vpp;
      const int part = rem / vpp, v = rem - part * vpp;
      const long row = (temporal_frame_row(b, f, F, lead, B) * HW + pix);
      st8<T>(&sQKV[f * W3 + part * Wd + v * 8], ld8<T>(qkv + row * C3 + (long)part * C + cbase + v * 8));
    }
  }
  __syncthreads();

  const int FP = F + 1;
  const int nsc = hpb * F * F;
  for (int u = tid; u < nsc; u += 256) {
    const int hh = u / (F * F), rem = u - hh * F * F;
    const int i = rem / F, j = rem - i * F;
    const T* qp = &sQKV[i * W3 + hh * hd];
    const T* kp = &sQKV[j * W3 + Wd + hh * hd];
    float acc = 0.0f;
    for (int d = 0; d < hd; d += 8) acc = dot8<V8>(ld8<T>(qp + d), ld8<T>(kp + d), acc);
    sP[(hh * F + i) * FP + j] = acc * scale_log2e;
  }
  __syncthreads();
  for (int u = tid; u < hpb * F; u += 256) {
    float* row = &sP[u * FP];
    float mx = -1e30f;
    for (int j = 0; j < F; ++j) mx = fmaxf(mx, row[j]);
    float sum = 0.0f;
    for (int j = 0; j < F; ++j) { const float e = exp2f(row[j] - mx); row[j] = e; sum += e; }
    const float inv = 1.0f / sum;
    for (int j = 0; j < F; ++j) row[j] *= inv;
  }
  __syncthreads();
  for (int u = tid; u < F * vpp; u += 256) {
    const int i = u / vpp, v = u - i * vpp;
    const int c0 = v * 8, hh = c0 / hd;
    const float* prow = &sP[(hh * F + i) * FP];
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
    for (int j = 0; j < F; ++j) {
      const float pj = prow[j];
      V8 vv = ld8<T>(&sQKV[j * W3 + 2 * Wd + c0]);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += pj * to_f32(vv[e]);
    }
    V8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(acc[e]);
    const long row = (temporal_frame_row(b, i, F, lead, B) * HW + pix);
    st8<T>(out + row * C + cbase + c0, o);
  }
}

// -------------------------------------------------------------------------------------------
// Temporal attention on the matrix pipe: one WAVE per (batch entry, pixel, head), no LDS, no barrier.
// The F' <= 32 frames of a pixel are one 32-row MFMA tile:
//   S^T[j, i] = K[j, :] . Q[i, :]^T   -- K / Q fragments are 16-byte loads straight from the fused [q | k | v] rows
//                                        (lane = frame, the row stride is the frame stride H*W*3C),
//   softmax over j is lane-local (+ one lane^32 exchange), P stays in registers,
//   O^T[d, i] = V^T[d, :] . P^T[:, i] -- V^T fragments are gathered with 2-byte loads (lane = channel d: 64 contiguous
//                                        bytes per frame row and instruction) in the k-slot order that matches the
//                                        accumulator layout of S^T, so no transposition pass exists anywhere.
// The VALU kernel above spends ~18x the tile's bytes in LDS reads (every (i, j) dot re-reads both rows); this one
// touches each operand once.  A workgroup = 4 consecutive heads of one pixel (320 contiguous bytes per frame row at
// head dim 40).
// -------------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ __launch_bounds__(256) void temporal_attn_mfma_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                                 int F, int HW, int C, int heads, float scale_log2e, int lead, int B) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  constexpr int HDP = ((HD + 15) / 16) * 16;
  constexpr int NKS = HDP / 16;
  constexpr int NDB = (HD + 31) / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int hgroups = heads >> 2;
  // (b, pixel, head group), head group fastest; XCD-contiguous: the head groups of a pixel and its neighbours share the
  // 128-byte lines of the fused q|k|v rows (1920 bytes per pixel and frame at C = 320), so they should meet in ONE L2 instead
  // of being dealt round-robin to the 8 XCDs
  const long item = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int hg = (int)(item % hgroups);
  const long bp = item / hgroups;                     // b * HW + pixel
  const int pix = (int)(bp % HW);
  const long b = bp / HW;
  const int h = hg * 4 + wave;
  const long C3 = 3L * C;
  const int fr = min(l31, F - 1);                     // rows >= F re-read the last frame (valid memory), masked below
  const long frow = temporal_frame_row(b, fr, F, lead, B);   // this lane's frame row of the [frames, HW, .] tensors

  // ---- S^T = K . Q^T ----
  const T* qrow = qkv + (frow * HW + pix) * C3 + (long)h * HD;
  const T* krow = qrow + C;
  V8 qf[NKS], kf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int d = ks * 16 + hi * 8;
    qf[ks] = (d < HD) ? ld8<T>(qrow + d) : zero8<T>();
    kf[ks] = (d < HD) ? ld8<T>(krow + d) : zero8<T>();
  }
  // V^T (A operand of O^T = V^T . P^T): k-slot (ks2, hi, e) <-> frame j = 8*(2*ks2 + (e >> 2)) + 4*hi + (e & 3).
  // Round 3: the V rows are read like Q and K -- 16-byte loads, lane = frame -- parked row-major in a wave-private LDS tile
  // [32 frames][HDP] and read back TRANSPOSED with ds_read_b64_tr_b16 (each 16-lane group fetches a [4 frames][16 d] block, lane
  // i of the group receives column i).  The first form gathered V^T with 32 two-byte global loads per lane, which held the
  // kernel at 3.2 TB/s (40 % of the HBM peak; profiles/r2_bench.json attention_families).
  constexpr int VP = HDP * 2;                         // row pitch in bytes: 96 / 160 / 320 -> the 4 rows of a transposing read fall on disjoint banks
  __shared__ __attribute__((aligned(16))) unsigned char vt_smem[4 * 32 * VP];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* const vt = (lds_u8*)vt_smem + wave * (32 * VP);
  {
    const T* vrow = qrow + 2 * C;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int d = ks * 16 + hi * 8;
      const V8 v8 = (d < HD) ? ld8<T>(vrow + d) : zero8<T>();
      *(__attribute__((address_space(3))) V8*)(vt + l31 * VP + d * 2) = v8;
    }
  }
  f32x16 s;
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) s = Vec<T>::mfma32(kf[ks], qf[ks], s);

  // ---- softmax over the key frames j (register r = 4g + jj <-> j = 8g + 4hi + jj; the partner lane holds the rest) ----
  float mx = -3.0e38f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = 8 * (r >> 2) + 4 * hi + (r & 3);
    s[r] = (j < F) ? s[r] : -3.0e38f;
    mx = fmaxf(mx, s[r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float mneg = -mx * scale_log2e;
  float l = 0.0f;
  V8 pf[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale_log2e, mneg));
    l += pv;
    pf[r >> 3][r & 7] = from_f32<T>(pv);
  }
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;

  // ---- O^T = V^T . P^T, normalise, store rows i < F (lane = query frame, 8-byte stores) ----
  const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  T* orow = out + (frow * HW + pix) * C + (long)h * HD;
  // the tile is private to the wave: LDS operations of one wave complete in order, no barrier
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int gi = lane & 15, gdh = (lane >> 4) & 1;
  const lds_u8* const vb = vt + (4 * hi + (gi >> 2)) * VP + (gi & 3) * 8;
  auto vfrag = [&](int db, int ks2) {
    // columns d = db*32 + gdh*16 + i of frames 16*ks2 + 4*hi + 0..3 (slots 0..3) and + 8 (slots 4..7).  Groups whose columns lie
    // past the padded head dim re-read the last 16 columns: those output rows (d >= HD) are never stored
    const int cb = min(db * 32 + gdh * 16, HDP - 16) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + (16 * ks2) * VP + cb));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + (16 * ks2 + 8) * VP + cb));
    const s16x8 v = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(V8, v);
  };
#pragma unroll
  for (int db = 0; db < NDB; ++db) {
    f32x16 o = Vec<T>::mfma32(vfrag(db, 0), pf[0], zero16);
    o = Vec<T>::mfma32(vfrag(db, 1), pf[1], o);
    if (l31 < F) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = db * 32 + 8 * g + 4 * hi;
        if (d0 < HD) {
          V4 w;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) w[jj] = from_f32<T>(o[4 * g + jj] * inv);
          *reinterpret_cast<V4*>(orow + d0) = w;
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------
// Temporal attention, LDS-staged form (round 3, 8 heads x head dim 40 / 80 = the 64 x 64 and 32 x 32 levels): one workgroup per
// PIXEL.  The fused [q | k | v] row of a pixel and frame is 3C contiguous elements (1920 / 3840 bytes = 15 / 30 whole lines); the
// kernel above lets each wave pick its head's 80 / 160-byte pieces out of those rows with one lane per frame, and stores 8 bytes
// per lane.  Here the F rows go global -> LDS by whole-line 16-byte loads (pitch 6C + 16 bytes), the four waves take two heads
// each with Q / K fragments and the transposed V reads straight from that tile (rows >= F clamp to the last frame: their
// probabilities are zero), O^T is written over the head's own q columns and the C output elements of every frame leave as whole
// lines.  Same MFMA / softmax sequence per (pixel, head) as above: the two kernels agree bit for bit.
// -------------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ __launch_bounds__(256) void temporal_attn_tiled_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                                  int F, int HW, float scale_log2e, int lead, int B) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  constexpr int HEADS = 8, C = HEADS * HD, C3 = 3 * C;
  constexpr int PITCH = C3 * 2 + 16;                 // bytes
  constexpr int HDP = ((HD + 15) / 16) * 16;
  constexpr int NKS = HDP / 16, NDB = (HD + 31) / 32;
  constexpr int CPR = C3 / 8, OPR = C / 8;           // 16-byte pieces per input / output row
  constexpr int MAXU = (19 * CPR + 255) / 256;       // pieces per thread with every load in flight (F <= 19; a tail loop covers more)
  extern __shared__ __attribute__((aligned(16))) unsigned char tt_smem[];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  unsigned char* const tile = tt_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const long bp = xcd_remap((int)blockIdx.x, (int)gridDim.x);       // b * HW + pixel
  const int pix = (int)(bp % HW);
  const long b = bp / HW;
  // element offset of temporal position f's row of this pixel (two row segments: temporal_frame_row)
  auto frame_off = [&](int f) { return (temporal_frame_row(b, f, F, lead, B) * HW + pix) * (long)C3; };

  // ---- the F rows of this pixel -> LDS ----
  {
    const int total = F * CPR;
    V8 tmp[MAXU];
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
      const int u = tid + 256 * i;
      if (u < total) { const int f = u / CPR, cc = u - f * CPR; tmp[i] = ld8<T>(qkv + frame_off(f) + cc * 8); }
    }
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
      const int u = tid + 256 * i;
      if (u < total) { const int f = u / CPR, cc = u - f * CPR; *reinterpret_cast<V8*>(tile + f * PITCH + cc * 16) = tmp[i]; }
    }
    for (int u = tid + 256 * MAXU; u < total; u += 256) {
      const int f = u / CPR, cc = u - f * CPR;
      *reinterpret_cast<V8*>(tile + f * PITCH + cc * 16) = ld8<T>(qkv + frame_off(f) + cc * 8);
    }
  }
  __syncthreads();

  const int fr = min(l31, F - 1);                     // rows >= F re-read the last frame, masked below
  const int gi = lane & 15, gdh = (lane >> 4) & 1;
  const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int h = wave + 4 * j;
    unsigned char* const qrow = tile + fr * PITCH + h * HD * 2;
    f32x16 s = zero16;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int d = ks * 16 + hi * 8;
      const V8 qf = (d < HD) ? *reinterpret_cast<const V8*>(qrow + d * 2) : zero8<T>();
      const V8 kf = (d < HD) ? *reinterpret_cast<const V8*>(qrow + (C + d) * 2) : zero8<T>();
      s = Vec<T>::mfma32(kf, qf, s);
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jf = 8 * (r >> 2) + 4 * hi + (r & 3);
      s[r] = (jf < F) ? s[r] : -3.0e38f;
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mneg = -mx * scale_log2e;
    float l = 0.0f;
    V8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], scale_log2e, mneg));
      l += pv;
      pf[r >> 3][r & 7] = from_f32<T>(pv);
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    // V^T fragments: frames 16 ks2 + 8 half + 4 hi + (gi >> 2), columns cb + 4 (gi & 3) ..; frames >= F clamp (their P is zero)
    const lds_u8* const vcol = (const lds_u8*)tile + (2 * C + h * HD) * 2 + (gi & 3) * 8;
    auto vread = [&](int frame, int cb) {
      return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vcol + min(frame, F - 1) * PITCH + cb));
    };
    const int frow = 4 * hi + (gi >> 2);
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      const int cb = min(db * 32 + gdh * 16, HDP - 16) * 2;
      const s16x8 v0 = __builtin_shufflevector(vread(frow, cb), vread(frow + 8, cb), 0, 1, 2, 3, 4, 5, 6, 7);
      const s16x8 v1 = __builtin_shufflevector(vread(frow + 16, cb), vread(frow + 24, cb), 0, 1, 2, 3, 4, 5, 6, 7);
      f32x16 o = Vec<T>::mfma32(__builtin_bit_cast(V8, v0), pf[0], zero16);
      o = Vec<T>::mfma32(__builtin_bit_cast(V8, v1), pf[1], o);
      if (l31 < F) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = db * 32 + 8 * g + 4 * hi;
          if (d0 < HD) {
            V4 w;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) w[jj] = from_f32<T>(o[4 * g + jj] * inv);
            *reinterpret_cast<V4*>(qrow + d0 * 2) = w;          // over this head's own q columns
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- the C output elements of every frame leave as whole lines ----
  for (int u = tid; u < F * OPR; u += 256) {
    const int f = u / OPR, cc = u - f * OPR;
    st8<T>(out + (temporal_frame_row(b, f, F, lead, B) * HW + pix) * C + cc * 8, *reinterpret_cast<const V8*>(tile + f * PITCH + cc * 16));
  }
}

// -------------------------------------------------------------------------------------------
// Token cross-attention (round 3): K/V of <= 32 rows -- the 32 audio tokens of a frame (hallo/models/attention.py:853-903,
// the three masked branches x 8 heads as one launch) or the 4 face tokens on the unfused path -- against thousands of query
// rows.  The flash kernels above give such a launch one workgroup per (query block, HEAD): a lane owns a query row and reads
// 80 bytes of it, the output leaves the same way, and every workgroup pays the K/V staging and LDS set-up for 20 KB of
// useful traffic: 81-95 us at the 64 x 64 level for 252 MB (a device copy moves them in 54).
// Here the unit is a 32-row query tile x a GROUP of heads spanning 320 contiguous channels (8 heads at head dim 40, 4 at 80, 2 at 160):
//   * workgroups are persistent over the tiles of one (frame, head group); the K fragments of a wave's heads stay in registers,
//     their V rows in a wave-private LDS tile (row-major, read back transposed with ds_read_b64_tr_b16 as in the temporal kernel);
//   * the query tile goes global -> LDS by whole-line 16-byte loads (pitch 2 CW + 16 B), S^T = K . Q^T takes its Q fragments from
//     LDS, softmax over the <= 32 tokens is lane-local + one lane^32 exchange, O^T = V^T . P^T is scaled (1 / l, optional fp32 row
//     scale) and written over the head's own Q columns of the tile, and the tile leaves by whole-line 16-byte stores.
// q must be pre-scaled (head_dim^-1/2 * log2 e), as everything on the UNet path is.
// -------------------------------------------------------------------------------------------
// PF (round 5): the NEXT query tile's global loads are issued right after this tile went to LDS, so they fly under the MFMA /
// softmax work of the current tile instead of opening the next iteration (a workgroup walks 1..8 tiles; the chain load -> LDS ->
// compute -> store per tile was fully serial inside a workgroup, hidden only by the 3 workgroups a CU holds).
template <typename T, int HD, bool PF>
__global__ __launch_bounds__(HD == 160 ? 128 : 256) void tok_attn_kernel(const AttnArgs p, int groups, int parts, int ntiles) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  constexpr int HG = (HD == 40) ? 8 : (HD == 80 ? 4 : 2);   // heads per workgroup: always 320 contiguous channels
  constexpr int NT = (HD == 160) ? 128 : 256;        // threads: 4 waves x 2 heads, 4 x 1, 2 x 1
  constexpr int HPW = HG / (NT / 64);                // heads per wave
  constexpr int CW = HG * HD;                        // channels per workgroup: 320
  constexpr int PITCH = CW * 2 + 16;                 // bytes
  constexpr int HDP = ((HD + 15) / 16) * 16;
  constexpr int NKS = HDP / 16, NDB = (HD + 31) / 32;
  constexpr int VP = HDP * 2;                        // V tile row pitch in bytes (96 / 160 / 320)
  constexpr int CPR = CW / 8, NLD = 32 * CPR / NT;
  static_assert((32 * CPR) % NT == 0, "tile pieces per thread");
  __shared__ __attribute__((aligned(16))) unsigned char tile[32 * PITCH];
  __shared__ __attribute__((aligned(16))) unsigned char vt_smem[(NT / 64) * HPW * 32 * VP];
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int bid = blockIdx.x;
  const int part = bid % parts, hg = (bid / parts) % groups, b = bid / (parts * groups);
  const int T_kv = p.Lkv1;
  const T* __restrict__ Qg = reinterpret_cast<const T*>(p.q) + (long)b * p.q_bs + hg * CW;
  T* __restrict__ Og = reinterpret_cast<T*>(p.o) + (long)b * p.o_bs + hg * CW;
  const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

  // ---- once per workgroup: K fragments (lane = token) and the V tiles of this wave's heads ----
  V8 kf[HPW][NKS];
  lds_u8* const vt0 = (lds_u8*)vt_smem + wave * (HPW * 32 * VP);
  {
    const int tok = min(l31, T_kv - 1);
#pragma unroll
    for (int j = 0; j < HPW; ++j) {
      const int h = hg * HG + wave * HPW + j;
      const T* krow = reinterpret_cast<const T*>(p.k1) + (long)b * p.k1_bs + (long)tok * p.k1_rs + h * HD;
      const T* vrow = reinterpret_cast<const T*>(p.v1) + (long)b * p.v1_bs + (long)tok * p.v1_rs + h * HD;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int d = ks * 16 + hi * 8;
        kf[j][ks] = (d < HD) ? ld8<T>(krow + d) : zero8<T>();
        const V8 v8 = (d < HD && l31 < T_kv) ? ld8<T>(vrow + d) : zero8<T>();        // rows >= T: zeros (their P is 0; 0 x garbage must not be NaN)
        *(__attribute__((address_space(3))) V8*)(vt0 + j * (32 * VP) + l31 * VP + d * 2) = v8;
      }
    }
  }
  const int gi = lane & 15, gdh = (lane >> 4) & 1;

  V8 st[NLD];
  auto load_tile = [&](int t) {
    const int r0 = t * 32;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = tid + NT * i, r = q / CPR, cc = q - r * CPR;
      st[i] = ld8<T>(Qg + (long)min(r0 + r, p.Lq - 1) * p.q_rs + cc * 8);
    }
  };
  if (PF && part < ntiles) load_tile(part);
  for (int t = part; t < ntiles; t += parts) {
    const int row0 = t * 32;
    // ---- query tile -> LDS ----
    {
      if (!PF) load_tile(t);
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int q = tid + NT * i, r = q / CPR, cc = q - r * CPR;
        *reinterpret_cast<V8*>(tile + r * PITCH + cc * 16) = st[i];
      }
      if (PF && t + parts < ntiles) load_tile(t + parts);
    }
    __syncthreads();
    const int qrow = min(row0 + l31, p.Lq - 1);
    unsigned char* trow = tile + l31 * PITCH;
#pragma unroll
    for (int j = 0; j < HPW; ++j) {
      const int hc = wave * HPW + j, h = hg * HG + hc, col0 = hc * HD;
      // S^T = K . Q^T (lane: query row l31, tokens 8g + 4hi + jj)
      f32x16 s = zero16;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int d = ks * 16 + hi * 8;
        const V8 xf = (d < HD) ? *reinterpret_cast<const V8*>(trow + (col0 + d) * 2) : zero8<T>();
        s = Vec<T>::mfma32(kf[j][ks], xf, s);
      }
      float mx = -3.0e38f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tk = 8 * (r >> 2) + 4 * hi + (r & 3);
        s[r] = (tk < T_kv) ? s[r] : -3.0e38f;
        mx = fmaxf(mx, s[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float l = 0.0f;
      V8 pf[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(s[r] - mx);
        l += pv;
        pf[r >> 3][r & 7] = from_f32<T>(pv);
      }
      l += __shfl_xor(l, 32, 64);
      float inv = 1.0f / l;
      if (p.o_rowscale) inv *= p.o_rowscale[(long)(h / p.rs_hdiv) * p.rs_stride + (long)b * p.Lq + qrow];
      // O^T = V^T . P^T over this head's own columns of the tile
      const lds_u8* const vb = vt0 + j * (32 * VP) + (4 * hi + (gi >> 2)) * VP + (gi & 3) * 8;
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        const int cb = min(db * 32 + gdh * 16, HDP - 16) * 2;
        f32x16 o;
        {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + cb));
          const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 8 * VP + cb));
          o = Vec<T>::mfma32(__builtin_bit_cast(V8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7)), pf[0], zero16);
        }
        {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 16 * VP + cb));
          const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 24 * VP + cb));
          o = Vec<T>::mfma32(__builtin_bit_cast(V8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7)), pf[1], o);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = db * 32 + 8 * g + 4 * hi;
          if (d0 < HD) {
            V4 w;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) w[jj] = from_f32<T>(o[4 * g + jj] * inv);
            *reinterpret_cast<V4*>(trow + (col0 + d0) * 2) = w;
          }
        }
      }
    }
    __syncthreads();
    // ---- tile -> out ----
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = tid + NT * i, r = q / CPR, cc = q - r * CPR;
      const V8 v = *reinterpret_cast<const V8*>(tile + r * PITCH + cc * 16);
      if (row0 + r < p.Lq) st8<T>(Og + (long)(row0 + r) * p.o_rs + cc * 8, v);
    }
    __syncthreads();
  }
}

static int g_last_attn = 0;       // hallo_get_option("last_attn_kernel"): 1 flash kernel of this file, 2 attention40.hip, 3 tok_attn_kernel

// hallo_set_option("tok_attn", 0 | 1 | 2): 0 flash kernels; 1 token cross-attention kernel for K/V of <= 32 rows (round 3); 2 (default, round 5) the
// same with the next query tile prefetched where that paid on MI355X (profiles/r5_prefetch_ab.json: workgroups that walk >= 4 tiles -- 64 x 64
// level, 57.4 -> 55.8 us -- and head dim 160 -- 29.8 -> 28.6 us; with 2 tiles per workgroup the extra 20 registers cost more than the overlap gives)
static int g_tok_attn = 2;

template <typename T, int HD>
static int launch_tok_attn_hd(const AttnArgs& a, hipStream_t st) {
  constexpr int HG = (HD == 40) ? 8 : (HD == 80 ? 4 : 2);
  constexpr int NT = (HD == 160) ? 128 : 256;
  const int groups = a.heads / HG;
  const int ntiles = (a.Lq + 31) / 32;
  // about the workgroups the LDS footprint keeps resident (3 per CU at 45 KB); each walks >= 1 tile
  const int target = (HD == 160 ? 1024 : 768);
  int parts = (target + a.batch * groups - 1) / (a.batch * groups);
  if (parts > ntiles) parts = ntiles;
  if (parts < 1) parts = 1;
  if (g_tok_attn == 2 && (HD == 160 || ntiles >= 4 * parts)) hipLaunchKernelGGL((tok_attn_kernel<T, HD, true>), dim3((unsigned)(a.batch * groups * parts)), dim3(NT), 0, st, a, groups, parts, ntiles);
  else hipLaunchKernelGGL((tok_attn_kernel<T, HD, false>), dim3((unsigned)(a.batch * groups * parts)), dim3(NT), 0, st, a, groups, parts, ntiles);
  HALLO_CHECK_LAUNCH();
  return 0;
}

template <typename T>
static int launch_tok_attn(const AttnArgs& a, int hd, hipStream_t st) {
  if (hd == 40) return launch_tok_attn_hd<T, 40>(a, st);
  if (hd == 80) return launch_tok_attn_hd<T, 80>(a, st);
  return launch_tok_attn_hd<T, 160>(a, st);
}

static int g_temporal_mfma = 2;   // hallo_set_option("temporal_mfma", 0 | 1 | 2): 0 VALU kernel, 1 one wave per (pixel, head), 2 (default) + the LDS-staged per-pixel kernel at 8 heads x head dim 40 / 80
static int g_attn_order = 2;      // hallo_set_option("attn_order", 0 query-block-fastest | 1 head-fastest | 2 auto: head-fastest for K/V of <= 128 rows)
static int g_attn40 = 1;          // hallo_set_option("attn40", 0 | 1): head-dim-40 pre-scaled-q launches on attention40.hip

}  // namespace hallo

using namespace hallo;

extern "C" int hallo_attention(const hallo_attn_desc* d, void* stream) {
  if (!d || !d->q || !d->k1 || !d->v1 || !d->o) return -22;
  if (d->batch <= 0 || d->heads <= 0 || d->Lq <= 0 || d->Lkv1 <= 0) return -22;
  if (d->head_dim != 40 && d->head_dim != 80 && d->head_dim != 160) return -22;
  if ((d->q_rs & 7) || (d->k1_rs & 7) || (d->v1_rs & 7) || (d->o_rs & 3)) return -22;
  if (d->k2 && (!d->v2 || (d->k2_rs & 7) || (d->v2_rs & 7) || d->kv2_batch_div < 1)) return -22;
  AttnArgs a;
  a.q = d->q; a.k1 = d->k1; a.v1 = d->v1; a.k2 = d->k2; a.v2 = d->v2; a.o = d->o;
  a.batch = d->batch; a.heads = d->heads; a.Lq = d->Lq; a.Lkv1 = d->Lkv1; a.Lkv2 = d->k2 ? d->Lkv2 : 0;
  a.q_bs = d->q_bs; a.q_rs = d->q_rs; a.k1_bs = d->k1_bs; a.k1_rs = d->k1_rs;
  a.v1_bs = d->v1_bs; a.v1_rs = d->v1_rs; a.k2_bs = d->k2_bs; a.k2_rs = d->k2_rs;
  a.v2_bs = d->v2_bs; a.v2_rs = d->v2_rs; a.o_bs = d->o_bs; a.o_rs = d->o_rs;
  a.kv2_div = d->kv2_batch_div > 0 ? d->kv2_batch_div : 1;
  a.kv2_first = d->kv2_first_batch;
  a.kv2_mod = d->kv2_batch_mod;
  a.scale_log2e = d->scale * 1.4426950408889634f;
  a.nqb = (d->Lq + 127) / 128;
  a.o_rowscale = d->o_rowscale;
  a.rs_hdiv = d->o_rowscale_head_div > 0 ? d->o_rowscale_head_div : d->heads;
  a.rs_stride = d->o_rowscale_stride;
  a.head_fastest = (g_attn_order == 1 || (g_attn_order == 2 && a.Lkv1 + a.Lkv2 <= 128)) ? 1 : 0;
  a.hs1 = d->kv1_hs > 0 ? d->kv1_hs : d->head_dim;
  a.hs2 = d->kv2_hs > 0 ? d->kv2_hs : d->head_dim;
  const bool head_strides = a.hs1 != d->head_dim || a.hs2 != d->head_dim;
  if (head_strides && ((a.hs1 | a.hs2) & 7)) return -22;
#ifdef HALLO_ABLATIONS
  if (getenv("HALLO_ATTN_KV_HEAD_MAJOR") && d->head_dim == 40) {      // timing only: read the same buffers AS IF K / V were stored [batch][head][row][40]
    a.k1_rs = a.v1_rs = 40; a.hs1 = (long)a.Lkv1 * 40; a.k1_bs = a.v1_bs = (long)a.heads * a.Lkv1 * 40;
    a.k2_rs = a.v2_rs = 40; a.hs2 = (long)a.Lkv2 * 40; a.k2_bs = a.v2_bs = (long)a.heads * a.Lkv2 * 40;
  }
#endif
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // attention40.hip stages K / V with 16-byte LDS-DMA (buffer addressing drops misaligned low address bits) and stores 16
  // bytes per lane: every K / V base, row and batch stride and the output must be 16-byte aligned, else the generic kernel
  const auto al16 = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
  const bool kv_aligned = al16(d->k1) && al16(d->v1) && !((d->k1_bs | d->v1_bs | d->k1_rs | d->v1_rs) & 7) &&
                          (!d->k2 || (al16(d->k2) && al16(d->v2) && !((d->k2_bs | d->v2_bs | d->k2_rs | d->v2_rs) & 7)));
  // token cross-attention: K/V of <= 32 rows, one segment, pre-scaled q, whole head groups, everything 16-byte aligned
  {
    const int hgp = d->head_dim == 40 ? 8 : (d->head_dim == 80 ? 4 : 2);
    if (g_tok_attn && !head_strides && !d->k2 && d->Lkv1 <= 32 && d->q_prescaled != 0 && (d->dtype == DT_F16 || d->dtype == DT_BF16) && kv_aligned &&
        d->heads % hgp == 0 && al16(d->q) && al16(d->o) && !((d->q_bs | d->o_bs | d->o_rs) & 7)) {
      g_last_attn = 3;
      if (d->dtype == DT_F16) return launch_tok_attn<_Float16>(a, d->head_dim, st);
      return launch_tok_attn<__bf16>(a, d->head_dim, st);
    }
  }
  if (g_attn40 && d->head_dim == 40 && d->q_prescaled != 0 && (d->dtype == DT_F16 || d->dtype == DT_BF16) && kv_aligned &&
      !(d->o_rs & 7) && !(d->o_bs & 7) && al16(d->o)) {
    g_last_attn = 2;
    return launch_attn40(a, d->dtype, st);          // attention40.hip: LDS-DMA staging + transposing V reads
  }
  if (head_strides) return -22;       // only attention40.hip takes head strides
  g_last_attn = 1;
  if (d->dtype == DT_F16) return launch_attn<_Float16>(a, d->head_dim, d->q_prescaled != 0, st);
  if (d->dtype == DT_BF16) return launch_attn<__bf16>(a, d->head_dim, d->q_prescaled != 0, st);
  return -22;
}

extern "C" int hallo_get_option_norm(const char* name);

extern "C" int hallo_get_option_attn(const char* name) {
  if (name && !strcmp(name, "attn40")) return g_attn40;
  if (name && !strcmp(name, "temporal_mfma")) return g_temporal_mfma;
  if (name && !strcmp(name, "attn_order")) return g_attn_order;
  if (name && !strcmp(name, "tok_attn")) return g_tok_attn;
  if (name && !strcmp(name, "last_attn_kernel")) return g_last_attn;
  return hallo_get_option_norm(name);      // norm_elementwise.hip: gn_fused (-> fused_xattn.hip: xattn_tiled)
}

extern "C" int hallo_set_option_xattn(const char* name, int value);
extern "C" int hallo_set_option_fp8(const char* name, int value);

extern "C" int hallo_set_option_attn(const char* name, int value) {
  if (name && !strcmp(name, "temporal_mfma")) { if (value < 0 || value > 2) return -22; g_temporal_mfma = value; return 0; }
  if (name && !strcmp(name, "attn_order")) { if (value < 0 || value > 2) return -22; g_attn_order = value; return 0; }
  if (name && !strcmp(name, "attn40")) { if (value < 0 || value > 40) return -22; g_attn40 = value; set_attn40_variant(value); return 0; }
  if (name && !strcmp(name, "tok_attn")) { if (value < 0 || value > 2) return -22; g_tok_attn = value; return 0; }
  if (name && (!strcmp(name, "xattn_tiled") || !strcmp(name, "xattn_cap"))) return hallo_set_option_xattn(name, value);     // fused_xattn.hip
  return hallo_set_option_fp8(name, value);                                                                                  // fp8.hip
}

extern "C" int hallo_temporal_attention_lead(const void* qkv, void* out, int B, int F, int lead, int HW, int C, int heads,
                                             float scale, int dtype, void* stream);

extern "C" int hallo_temporal_attention(const void* qkv, void* out, int B, int F, int HW, int C, int heads,
                                        float scale, int dtype, void* stream) {
  return hallo_temporal_attention_lead(qkv, out, B, F, 0, HW, C, heads, scale, dtype, stream);
}

extern "C" int hallo_temporal_attention_lead(const void* qkv, void* out, int B, int F, int lead, int HW, int C, int heads,
                                             float scale, int dtype, void* stream) {
  if (!qkv || !out || B <= 0 || F <= 0 || F > 32 || HW <= 0 || C <= 0 || heads <= 0 || lead < 0 || lead >= F) return -22;
  if (C % heads || (C / heads) % 8) return -22;
  const int hd = C / heads;
  hipStream_t st0 = reinterpret_cast<hipStream_t>(stream);
  if (g_temporal_mfma >= 2 && heads == 8 && (hd == 40 || hd == 80) && (dtype == DT_F16 || dtype == DT_BF16)) {
    // LDS-staged form: one workgroup per pixel, F rows of 6C + 16 bytes (34.8 / 69.4 KB at 18 frames)
    const float sl0 = scale * 1.4426950408889634f;
    const size_t lds = (size_t)F * (3 * C * 2 + 16);
    static bool attr_done[64][4] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -19;
    const int slot = (dtype == DT_F16 ? 0 : 2) + (hd == 40 ? 0 : 1);
#define HALLO_TTILED(TT, HDv)                                                                                                                  \
    do {                                                                                                                                       \
      if (!attr_done[dev][slot]) {                                                                                                             \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_tiled_kernel<TT, HDv>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                32 * (3 * 8 * HDv * 2 + 16)) != hipSuccess) return -19;                                                        \
        attr_done[dev][slot] = true;                                                                                                           \
      }                                                                                                                                        \
      hipLaunchKernelGGL((temporal_attn_tiled_kernel<TT, HDv>), dim3((unsigned)((long)B * HW)), dim3(256), lds, st0,                            \
                         reinterpret_cast<const TT*>(qkv), reinterpret_cast<TT*>(out), F, HW, sl0, lead, B);                                            \
    } while (0)
    if (dtype == DT_F16) { if (hd == 40) HALLO_TTILED(_Float16, 40); else HALLO_TTILED(_Float16, 80); }
    else { if (hd == 40) HALLO_TTILED(__bf16, 40); else HALLO_TTILED(__bf16, 80); }
#undef HALLO_TTILED
    HALLO_CHECK_LAUNCH();
    return 0;
  }
  if (g_temporal_mfma && (heads & 3) == 0 && (hd == 40 || hd == 80 || hd == 160) && (dtype == DT_F16 || dtype == DT_BF16)) {
    const float sl0 = scale * 1.4426950408889634f;
    dim3 grid((unsigned)((long)B * HW * (heads / 4))), block(256);
#define HALLO_TMFMA(TT, HDv) hipLaunchKernelGGL((temporal_attn_mfma_kernel<TT, HDv>), grid, block, 0, st0, \
    reinterpret_cast<const TT*>(qkv), reinterpret_cast<TT*>(out), F, HW, C, heads, sl0, lead, B)
    if (dtype == DT_F16) { if (hd == 40) HALLO_TMFMA(_Float16, 40); else if (hd == 80) HALLO_TMFMA(_Float16, 80); else HALLO_TMFMA(_Float16, 160); }
    else { if (hd == 40) HALLO_TMFMA(__bf16, 40); else if (hd == 80) HALLO_TMFMA(__bf16, 80); else HALLO_TMFMA(__bf16, 160); }
#undef HALLO_TMFMA
    HALLO_CHECK_LAUNCH();
    return 0;
  }
  // heads per block: largest divisor of `heads` whose slab fits ~40 KB (>= 4 workgroups per CU)
  int hpb = heads;
  auto lds_for = [&](int g) { return (size_t)F * (3 * g * hd + 8) * 2 + (size_t)g * F * (F + 1) * sizeof(float); };
  while (hpb > 1 && (lds_for(hpb) > 40 * 1024 || heads % hpb)) --hpb;
  const size_t lds = lds_for(hpb);
  if (lds > 64 * 1024) return -22;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(B * HW, heads / hpb), block(256);
  const float sl = scale * 1.4426950408889634f;
  if (dtype == DT_F16) {
    hipLaunchKernelGGL((temporal_attn_kernel<_Float16>), grid, block, lds, st,
                       reinterpret_cast<const _Float16*>(qkv), reinterpret_cast<_Float16*>(out), F, HW, C, hd, hpb, sl, lead, B);
  } else if (dtype == DT_BF16) {
    hipLaunchKernelGGL((temporal_attn_kernel<__bf16>), grid, block, lds, st,
                       reinterpret_cast<const __bf16*>(qkv), reinterpret_cast<__bf16*>(out), F, HW, C, hd, hpb, sl, lead, B);
  } else {
    return -22;
  }
  HALLO_CHECK_LAUNCH();
  return 0;
}
