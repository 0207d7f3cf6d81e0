// Flash-style attention for head dim 40 on gfx950: K/V tiles staged by LDS-DMA, V consumed with the transposing LDS read.
//
// Same operator as attention.hip's attn_kernel<T, 40, true> (diffusers AttnProcessor2_0 / F.scaled_dot_product_attention as
// called from hallo/models/mutual_self_attention.py:253-284 -- K/V = [self ; reference bank], CFG uncond rows self only --
// and hallo/models/attention.py:828-884), for q pre-scaled by head_dim^-1/2 * log2(e).  What changed, and why:
//
//  * On the hd-40 shapes the kernel is VALU-bound (r1 PMC: VALU busy 65-68 %, matrix pipe 40-45 %).  Per 64-key tile a wave
//    needs 32 exp + 16 convert + 16 max3 for the softmax -- and spent ~25 more VALU / 10 LDS-write instructions on STAGING:
//    global -> VGPR -> LDS for K, and a VGPR transpose (8 two-element packs + 8 ds_write_b32 per thread) to build V^T.
//  * Here K and V go global -> LDS by DMA (`buffer_load_dwordx4 ... lds`, 16 B per lane, lane-linear destination): no staging
//    registers, no address VALU in the loop (per-lane offsets are computed once per segment, the tile advance is the scalar
//    offset), no LDS write instructions.  V stays ROW-major in LDS and the A operand of O^T += V^T . P^T is read with
//    `ds_read_b64_tr_b16` (each 16-lane group fetches a [4 kv][16 d] block transposed): the V^T build disappears.
//  * LDS images are dense and conflict-free without padding: K [64][40] at an 80-byte pitch (5 x 16 B, odd -> the 16 rows of a
//    ds_read_b128 lane group hit 16 distinct 16-byte slots); V as two planes, d 0..31 [64][32] (64-byte pitch: the 4 rows x
//    64 B of one transposing read are 256 contiguous bytes) and d 32..39 [64][8].
//  * The pad operands are LDS constants addressed per lane instead of per-tile data: the K fragment lanes of columns 40..47
//    read a constant [1, 0, ..., 0] (column 40 = 1.0 carries -m_run of the Q fragment into the scores, as in attention.hip),
//    the V fragment lanes of d = 40..43 read the same constant (row 40 of O^T = sum_kv P: row sums on the matrix pipe),
//    d >= 44 read zeros.
//
// Work decomposition, swapped-operand MFMAs (lane owns a query row), software pipeline (QK^T of tile t+1 | exp / convert of
// tile t | PV of tile t in one basic block), deferred rescale and the two-segment K/V walk are attention.hip's.
#include "common.h"
#include "attn_args.h"
#include <type_traits>

namespace hallo {

namespace {

constexpr int A40_HD = 40, A40_KVB = 64;
constexpr int A40_K_TILE = A40_KVB * 80;          // bytes: [64][40] elements, 80-byte pitch
constexpr int A40_VA_TILE = A40_KVB * 64;         // d 0..31
constexpr int A40_VB_TILE = A40_KVB * 16;         // d 32..39
constexpr int A40_OFF_K = 0;
constexpr int A40_OFF_VA = A40_OFF_K + 2 * A40_K_TILE;       // 10240
constexpr int A40_OFF_VB = A40_OFF_VA + 2 * A40_VA_TILE;     // 18432
constexpr int A40_OFF_ONE = A40_OFF_VB + 2 * A40_VB_TILE;    // 20480: 16-byte pattern [1, 0, 0, 0, 0, 0, 0, 0]
constexpr int A40_ONE_BYTES = 8192;                          // >= largest immediate of a K pad read (stage + t) + 16
constexpr int A40_OFF_ZERO = A40_OFF_ONE + A40_ONE_BYTES;    // 28672
constexpr int A40_ZERO_BYTES = 2304;                         // >= largest immediate of a V pad read + 8 + A40_PAD_ZERO_SKEW
// Bank skew of the constant operands of the plane-B transposing read.  Plane B, the ones pattern and the zeros all start at
// bank 0 and are addressed with the same immediates, so the three addresses of one ds_read_b64_tr_b16 (8 lanes of V data,
// the lanes of row 40, the zero lanes) met on the same banks: SQ_LDS_BANK_CONFLICT was 30 % of SQ_LDS_IDX_ACTIVE
// (profiles/r2_attn_pmc.json).  The patterns repeat every 16 bytes, so the constants are simply read 64 / 128 bytes further.
constexpr int A40_PAD_ONE_SKEW = 64, A40_PAD_ZERO_SKEW = 128;
constexpr int A40_LDS = A40_OFF_ZERO + A40_ZERO_BYTES;       // 30720

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

}  // namespace

// EXA = exponentials (of 32 per tile) issued before the barrier; PRIO = raise the wave priority around the MFMA clusters.
// ABL (-DHALLO_ABLATIONS builds, tools/cbench attn-time 10..15; WRONG results, timing only): 1 = no K fragment LDS reads (the Q
// fragment stands in), 2 = no V fragment LDS reads (P stands in), 4 = the exponential is a multiplication (a full-rate VALU
// instruction instead of a quarter-rate one), 8 = no exponentials and no converts at all, 16 = no running maximum, 32 = no K / V DMA (the LDS tiles keep their initial garbage), 64 = no
// per-tile barrier, 128 = the per-tile barrier does not wait for the DMA queue (with 2: nothing in the loop does)
//
// PV48 (round 6): the second 32-row block of O^T -- d = 32..39, the row-sum row 40 and 23 rows of zeros -- as 16-row blocks on
// v_mfma_f32_16x16x32: per 32 keys two MFMAs of 16 matrix cycles (one per 16 queries) instead of two of 32, i.e. 12.3 instead of
// 14 32x32x16-equivalents per 64-key tile.  The timing ablations (profiles/r6_attn40_ablations.txt) say that is what counts: with
// every LDS fragment read, exponential, convert and maximum REMOVED the kernel keeps 78 % of its time -- it is bound by the
// matrix pipe at the clock the power limit allows, and 40 % of the MFMA work it executes is padding.  The 16x16x32 B operand wants
// P^T as [8 keys of group lane>>4][query lane&15]: v_permlane16_swap of the dwords of pf[t][0] / pf[t][1] (the two 16-key slices
// of a 32-key block, issued after the 32-row block's MFMAs have consumed them) leaves in the first register the operand of
// queries 0..15 and in the second that of queries 16..31, key groups (slice, half) = (0,0) (1,0) (0,1) (1,1); the A operand is the
// same transposing read of V plane B with the 16-lane group choosing (slice, half) instead of (d half, half).
template <typename T, int EXA, int PRIO, int ABL = 0, bool PV48 = false, int AUX = 0, bool KPRE = false>
__global__ __launch_bounds__(256, 3) void attn40_kernel(const AttnArgs p) {
  using V8 = typename Vec<T>::v8;
  using V4 = typename Vec<T>::v4;
  constexpr int HD = A40_HD, KVB = A40_KVB, NT = 2, NKS = 3, NDB = 2;
  constexpr int L_DB = 1, L_R = 4;               // accumulator slot of O^T row 40 (lanes hi = 0): block 1, row 8 -> r = 4

  // ONE shared object (a second one makes hipcc drain the DMA queue before every LDS read)
  __shared__ __attribute__((aligned(16))) unsigned char smem[A40_LDS];
  lds_u8* const lds = (lds_u8*)smem;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int hi = lane >> 5, l31 = lane & 31;

  const int nwg = p.batch * p.heads * p.nqb;
  int bid = xcd_remap(blockIdx.x, nwg);
  // block order: query block fastest, then head, then frame -- the ~96 workgroups an XCD runs at a time cover three
  // neighbouring HEADS of one frame, whose 80-byte K / V row slices share 128-byte lines.  (Frame-fastest order, which lets
  // three frames share the reference bank's K/V of a head, was measured and is worse: the lines are then used by one head
  // only -- FETCH_SIZE 2.4x / 3.1x the algorithmic reads instead of 1.15x / 2.6x, and the kernel 7 % slower.)
  int qb, h;
  if (p.head_fastest) { h = bid % p.heads; bid /= p.heads; qb = bid % p.nqb; bid /= p.nqb; }
  else { qb = bid % p.nqb; bid /= p.nqb; h = bid % p.heads; bid /= p.heads; }
  const int b = bid;

  const T* __restrict__ Qg = reinterpret_cast<const T*>(p.q) + (long)b * p.q_bs + h * HD;
  const T* __restrict__ K1 = reinterpret_cast<const T*>(p.k1) + (long)b * p.k1_bs + h * p.hs1;
  const T* __restrict__ V1 = reinterpret_cast<const T*>(p.v1) + (long)b * p.v1_bs + h * p.hs1;
  const bool use2 = p.k2 != nullptr && p.Lkv2 > 0 && b >= p.kv2_first;
  const int b2 = use2 ? (p.kv2_mod > 0 ? (b / p.kv2_div) % p.kv2_mod : b / p.kv2_div) : 0;
  const T* __restrict__ K2 = use2 ? reinterpret_cast<const T*>(p.k2) + (long)b2 * p.k2_bs + h * p.hs2 : K1;
  const T* __restrict__ V2 = use2 ? reinterpret_cast<const T*>(p.v2) + (long)b2 * p.v2_bs + h * p.hs2 : V1;
  T* __restrict__ Og = reinterpret_cast<T*>(p.o) + (long)b * p.o_bs + h * HD;

  // ---- LDS constants ----
  {
    V8 one = zero8<T>();
    one[0] = from_f32<T>(1.0f);
    for (int i = tid; i < A40_ONE_BYTES / 16; i += 256) *reinterpret_cast<V8*>(smem + A40_OFF_ONE + i * 16) = one;
    for (int i = tid; i < A40_ZERO_BYTES / 16; i += 256) *reinterpret_cast<V8*>(smem + A40_OFF_ZERO + i * 16) = zero8<T>();
  }

  // ---- Q fragments (B operand of S^T = K . Q^T): lane holds Q[q0 + l31][ks*16 + hi*8 .. +8]; element 40 (ks = 2, hi = 1,
  // slot 0) carries -m_run ----
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

  // ---- loaders: LDS-DMA through buffer descriptors.  A K tile is 5 wave-instructions (320 chunks of 16 B, chunk c = row
  // c / 5, column chunk c % 5), a V tile 4 (plane A: chunk c = row c / 4, column chunk c % 4) + 1 (plane B: one chunk per
  // row).  Wave w issues K instructions {w} (+ {4} for w = 0) and V instructions w0: A2, w1: A0 A3, w2: A1, w3: B.  Rows past
  // the end of a segment (ragged last tile) get an out-of-range offset: the DMA writes zeros.
  constexpr unsigned OOB = 0xFFFFFFFFu;
  auto clamp32 = [](long bytes) { return (int)(bytes > 0xFFFFFFFFL ? 0xFFFFFFFFL : (bytes < 0 ? 0 : bytes)); };
  auto mk_rsrc = [&](const T* base, long rs, int L) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, clamp32(((long)(L - 1) * rs + HD) * 2), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsK1 = mk_rsrc(K1, p.k1_rs, p.Lkv1), rsV1 = mk_rsrc(V1, p.v1_rs, p.Lkv1);
  const __amdgpu_buffer_rsrc_t rsK2 = mk_rsrc(K2, use2 ? p.k2_rs : p.k1_rs, use2 ? p.Lkv2 : p.Lkv1);
  const __amdgpu_buffer_rsrc_t rsV2 = mk_rsrc(V2, use2 ? p.v2_rs : p.v1_rs, use2 ? p.Lkv2 : p.Lkv1);

  // tile row / column chunk of this lane in each of its DMA instructions (fixed for the kernel)
  const int kc0 = wave_u * 64 + lane, kc1 = 256 + lane;                 // K chunks
  const int krow0 = kc0 / 5, kcol0 = kc0 - 5 * krow0, krow1 = kc1 / 5, kcol1 = kc1 - 5 * krow1;
  const int va_i0 = (wave_u == 0) ? 2 : (wave_u == 1 ? 0 : 1);          // plane-A instruction of V slot 0 (waves 0..2)
  const int vc0 = va_i0 * 64 + lane, vc1 = 192 + lane;                  // plane-A chunks (slot 1: instruction 3, wave 1 only)
  const int vrow0 = (wave_u == 3) ? lane : (vc0 >> 2), vcol0 = (wave_u == 3) ? 4 : (vc0 & 3);
  const int vrow1 = vc1 >> 2, vcol1 = vc1 & 3;
  const int k_dst0 = A40_OFF_K + wave_u * 1024, k_dst1 = A40_OFF_K + 4096;
  const int v_dst0 = (wave_u == 3) ? A40_OFF_VB : A40_OFF_VA + va_i0 * 1024, v_dst1 = A40_OFF_VA + 3072;
  const int v_stage0 = (wave_u == 3) ? A40_VB_TILE : A40_VA_TILE;       // stage stride of V slot 0

  unsigned offK0, offK1, offV0, offV1;
  auto set_segment_k = [&](long krs) {
    offK0 = (unsigned)((krow0 * krs + kcol0 * 8) * 2);
    offK1 = (unsigned)((krow1 * krs + kcol1 * 8) * 2);
  };
  auto set_segment_v = [&](long vrs) {
    offV0 = (unsigned)((vrow0 * vrs + vcol0 * 8) * 2);
    offV1 = (unsigned)((vrow1 * vrs + vcol1 * 8) * 2);
  };
  set_segment_k(p.k1_rs);
  set_segment_v(p.v1_rs);

#define A40_DMA(rs, dst, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (dst)), 16, (int)(voff), (int)(soff), 0, AUX)

  // Stream cursors (round 3).  The K stream runs three tiles ahead of the tile being consumed, the V stream one: each keeps
  // its OWN position -- descriptor of the current segment, scalar byte offset of the next tile, rows left in the segment --
  // and moves to the second segment in a rarely taken wave-uniform branch.  The per-tile issue is then: [ragged check],
  // DMA, offset += step, tile counter; the round-2 form re-derived segment, descriptor and offset from the tile index for
  // every issue (selects and branches over two descriptors: ~50 scalar instructions and 8 branches per tile in the ISA).
  __amdgpu_buffer_rsrc_t cK_rs = rsK1, cV_rs = rsV1;
  int cK_so = 0, cK_step = (int)(KVB * p.k1_rs * 2), cK_left = p.Lkv1, cK_tiles = nt1;
  int cV_so = 0, cV_step = (int)(KVB * p.v1_rs * 2), cV_left = p.Lkv1, cV_tiles = nt1;
  auto dma_k = [&](auto stg_c) {
    constexpr int st = decltype(stg_c)::value * A40_K_TILE;
    unsigned o0 = offK0, o1 = offK1;
    if (__builtin_expect(cK_left < KVB, 0)) {
      o0 = (krow0 < cK_left) ? o0 : OOB;
      o1 = (krow1 < cK_left) ? o1 : OOB;
    }
    if (!(ABL & 32)) {
    A40_DMA(cK_rs, k_dst0 + st, o0, cK_so);
    if (wave_u == 0) A40_DMA(cK_rs, k_dst1 + st, o1, cK_so);
    }
    cK_so += cK_step; cK_left -= KVB;
    if (__builtin_expect(--cK_tiles == 0, 0)) {        // the next tile opens the second segment (if any)
      cK_rs = rsK2; cK_so = 0; cK_step = (int)(KVB * p.k2_rs * 2); cK_left = p.Lkv2; cK_tiles = nt2;
      set_segment_k(p.k2_rs);
    }
  };
  auto dma_v = [&](auto stg_c) {
    constexpr int stg = decltype(stg_c)::value;
    const int st0 = stg * v_stage0, st1 = stg * A40_VA_TILE;
    unsigned o0 = offV0, o1 = offV1;
    if (__builtin_expect(cV_left < KVB, 0)) {
      o0 = (vrow0 < cV_left) ? o0 : OOB;
      o1 = (vrow1 < cV_left) ? o1 : OOB;
    }
    if (!(ABL & 32)) {
    A40_DMA(cV_rs, v_dst0 + st0, o0, cV_so);
    if (wave_u == 1) A40_DMA(cV_rs, v_dst1 + st1, o1, cV_so);
    }
    cV_so += cV_step; cV_left -= KVB;
    if (__builtin_expect(--cV_tiles == 0, 0)) {
      cV_rs = rsV2; cV_so = 0; cV_step = (int)(KVB * p.v2_rs * 2); cV_left = p.Lkv2; cV_tiles = nt2;
      set_segment_v(p.v2_rs);
    }
  };
#undef A40_DMA

  // ---- fragment read addresses (per lane, fixed): everything else is an immediate offset ----
  // K fragment (A operand of S^T): K[t*32 + l31][ks*16 + hi*8 .. +8]; ks = 2 / hi = 1 (columns 40..47) -> the constant
  const lds_u8* const kb01 = lds + A40_OFF_K + l31 * 80 + hi * 16;
  const lds_u8* const kb2 = hi ? lds + A40_OFF_ONE : lds + A40_OFF_K + l31 * 80 + 64;
  // V fragment (A operand of O^T): transposing read, 16-lane group g = (d half, kv half hi); lane i of the group addresses
  // row 4*hi + (i >> 2), 4 columns (i & 3)*4 of the group's 16
  const int gi = lane & 15, gdh = (lane >> 4) & 1;
  const lds_u8* const vbA = lds + A40_OFF_VA + (4 * hi + (gi >> 2)) * 64 + gdh * 32 + (gi & 3) * 8;
  const lds_u8* const vbB = gdh ? lds + A40_OFF_ZERO + A40_PAD_ZERO_SKEW
                                : ((gi & 3) < 2 ? lds + A40_OFF_VB + (4 * hi + (gi >> 2)) * 16 + (gi & 3) * 8
                                                : ((gi & 3) == 2 ? lds + A40_OFF_ONE + A40_PAD_ONE_SKEW : lds + A40_OFF_ZERO + A40_PAD_ZERO_SKEW));

  // PV48: V fragment of the 16-row block d = 32..47 for v_mfma_f32_16x16x32: 16-lane group g = lane >> 4 = key group (slice g & 1,
  // half g >> 1); lane i of the group addresses row 16 * slice + 4 * half + (i >> 2), 4 columns (i & 3) * 4
  const int g16 = lane >> 4;
  const lds_u8* const vbB16 = (gi & 3) < 2 ? lds + A40_OFF_VB + (16 * (g16 & 1) + 4 * (g16 >> 1) + (gi >> 2)) * 16 + (gi & 3) * 8
                                           : ((gi & 3) == 2 ? lds + A40_OFF_ONE + A40_PAD_ONE_SKEW : lds + A40_OFF_ZERO + A40_PAD_ZERO_SKEW);

  f32x16 oacc[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.0f;
  // PV48: O^T rows 32..47 of queries 0..15 / 16..31: lane holds rows 32 + 4 * (lane >> 4) + r of query (lane & 15) (+ 16)
  f32x4 o1a = {0.0f, 0.0f, 0.0f, 0.0f}, o1b = {0.0f, 0.0f, 0.0f, 0.0f};
  float m_run = 0.0f;
  constexpr float RESCALE_THR = 6.0f;   // log2 units: P <= 64
  const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

  // S^T = K . Q'^T of K stage `stg` (compile-time)
  auto qk = [&](auto stg_c, f32x16* sc) {
    constexpr int ST = decltype(stg_c)::value * A40_K_TILE;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      typedef const __attribute__((address_space(3))) V8* ldsv8;
      const V8 k0 = (ABL & 1) ? qf[0] : *(ldsv8)(kb01 + ST + t * 32 * 80);
      const V8 k1 = (ABL & 1) ? qf[1] : *(ldsv8)(kb01 + ST + t * 32 * 80 + 32);
      const V8 k2 = (ABL & 1) ? qf[2] : *(ldsv8)(kb2 + ST + t * 32 * 80);
      sc[t] = Vec<T>::mfma32(k0, qf[0], zero16);
      sc[t] = Vec<T>::mfma32(k1, qf[1], sc[t]);
      sc[t] = Vec<T>::mfma32(k2, qf[2], sc[t]);
    }
  };
  // the six K fragments of a stage, without their MFMAs (KPRE)
  auto kload = [&](auto stg_c, auto on_c, auto* kf) {
    if constexpr (KPRE && decltype(on_c)::value) {
      constexpr int ST = decltype(stg_c)::value * A40_K_TILE;
      typedef const __attribute__((address_space(3))) V8* ldsv8;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        kf[t][0] = *(ldsv8)(kb01 + ST + t * 32 * 80);
        kf[t][1] = *(ldsv8)(kb01 + ST + t * 32 * 80 + 32);
        kf[t][2] = *(ldsv8)(kb2 + ST + t * 32 * 80);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto tr4 = [&](const lds_u8* ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
  };

  // ---- pipeline.  hipcc drains the DMA queue (vmcnt(0)) in front of the first transposing read that follows a DMA issue in
  // program order, so the DMA is issued AFTER the PV reads of an iteration and its wait sits where it is needed anyway: at the
  // one barrier per tile, which stands between the softmax / QK^T half and the PV half.  Iteration `it`:
  //   A  softmax statistics of tile it, QK^T of tile it+1 (K stage (it+1)&1: landed before barrier B of iteration it-1),
  //      the first EXA exponentials
  //   B  vmcnt(0) + barrier: V(it) and K(it+2), issued at D of iteration it-1, have landed for every wave
  //   C  remaining exponentials | O^T += V^T . P^T of tile it (V stage it&1)
  //   D  DMA V(it+1) -> V stage (it+1)&1 (last read in C of it-1) and K(it+3) -> K stage (it+1)&1 (last read in A of it)
  f32x16 s_a[NT], s_b[NT];
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  int cons_left = p.Lkv1, cons_tiles = nt1;      // the tile being consumed: rows left / tiles left in its segment
  if (nt > 0) {
    dma_k(S0{});
    dma_v(S0{});
    if (nt > 1) dma_k(S1{});
  }
  __syncthreads();                      // vmcnt(0) + barrier: constants and the first tiles are visible
  if (nt > 0) qk(std::integral_constant<int, 0>{}, s_a);
  __syncthreads();                      // every wave has read K(0) before K(2) lands in its stage
  if (nt > 2) dma_k(S0{});

  auto tile_step = [&](auto more_c, auto par_c, int it, f32x16* s_cur, f32x16* s_nxt) {
    constexpr bool MORE = decltype(more_c)::value;      // a tile it+1 exists
    constexpr int PAR = decltype(par_c)::value;         // it & 1
    // KPRE (round 6, default with PV48): the six K fragments of tile it+1 are requested before the softmax statistics of tile it instead of
    // next to their MFMAs (hipcc's own placement waits for the second block's three reads at the barrier): -0.9 % per launch, same results
    typename std::conditional<KPRE, V8, char>::type kpre[NT][3];      // (no V8 object at all in the forms without KPRE: a dead V8 array there cost them 4 spilled VGPRs)
    kload(std::integral_constant<int, 1 - PAR>{}, more_c, kpre);

    if (__builtin_expect(cons_left < KVB, 0)) {
      asm volatile("" ::: "memory");   // keep this a real (wave-uniform) branch
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kv = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          s_cur[t][r] = (kv < cons_left) ? s_cur[t][r] : -3.0e38f;
        }
    }
    cons_left -= KVB;
    if (__builtin_expect(--cons_tiles == 0, 0)) { cons_left = p.Lkv2; cons_tiles = nt2; }
    float mx = -3.0e38f;
    if (ABL & 16) mx = s_cur[0][0];
    else
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(s_cur[t][r], s_cur[t][r + 1]), mx);
    {   // the partner lane (lane ^ 32) holds the other half of the row: one VALU half-swap instead of an LDS round trip.
        // Written as inline asm: through __builtin_amdgcn_permlane32_swap hipcc (ROCm 7.2) folds fmaxf(sw[0], sw[1]) to
        // sw[0] (ISA: v_permlane32_swap v32, v33 ; v_max_f32 v32, v32, v32 -- with identical or distinct operands), so the
        // maximum covered the hi = 0 half of the keys only.  That is still a valid softmax reference value (results were
        // right), but P was then unbounded: fp16 rows whose other half was > 2^10 larger overflowed to inf / NaN (found by
        // the run-to-run identity test, whose q is unscaled).  s_nop 1: VALU write -> permlane read wait states.
      float mx_hi = mx;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(mx), "+v"(mx_hi));
    }
    // s_cur holds s - m_run (m_run = 0 before the first tile, which always takes this branch); deferred rescale
    if (__builtin_expect(it == 0 || !__all(mx <= RESCALE_THR), 0)) {
      asm volatile("" ::: "memory");
      const float want = m_run + (it == 0 ? mx : fmaxf(mx, 0.0f));
      const float m_new = to_f32(from_f32<T>(want));          // any reference value works; it must be exact in T
      const float delta = m_new - m_run;
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      m_run = m_new;
      if (hi) qf[2][0] = from_f32<T>(-m_new);                  // Q'[row][40]
#pragma unroll
      for (int db = 0; db < (PV48 ? 1 : NDB); ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
      if (PV48) {      // the 16-row blocks hold query (lane & 15) (+ 16): its factor lives in that lane (this branch is wave-uniform)
        const float aa = __shfl(alpha, lane & 15, 64), ab = __shfl(alpha, 16 + (lane & 15), 64);
#pragma unroll
        for (int r = 0; r < 4; ++r) { o1a[r] *= aa; o1b[r] *= ab; }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_cur[t][r] -= delta;
    }

    // ---- A: QK^T of tile it+1 | first exponentials of tile it ----
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    if constexpr (KPRE && MORE) {
      // (the two blocks' chains interleaved -- t inner, k16 step outer -- is 1 % SLOWER: profiles/r6_attn40_pv48.txt)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        s_nxt[t] = Vec<T>::mfma32(kpre[t][0], qf[0], zero16);
        s_nxt[t] = Vec<T>::mfma32(kpre[t][1], qf[1], s_nxt[t]);
        s_nxt[t] = Vec<T>::mfma32(kpre[t][2], qf[2], s_nxt[t]);
      }
    } else if (MORE) qk(std::integral_constant<int, 1 - PAR>{}, s_nxt);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    V8 pf[NT][2];
    auto pexp = [&](float x) { return (ABL & 4) ? x * 0.001f : __builtin_amdgcn_exp2f(x); };
    if (ABL & 8) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 lo = {s_cur[t][0], s_cur[t][1], s_cur[t][2], s_cur[t][3]}, hi4 = {s_cur[t][4], s_cur[t][5], s_cur[t][6], s_cur[t][7]};
        pf[t][0] = __builtin_bit_cast(V8, lo);
        pf[t][1] = __builtin_bit_cast(V8, hi4);
      }
    } else
#pragma unroll
    for (int e = 0; e < EXA; ++e)
      pf[e >> 4][(e >> 3) & 1][e & 7] = from_f32<T>(pexp(s_cur[e >> 4][e & 15]));
    if (ABL & 128) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    else if (!(ABL & 64)) __syncthreads();        // B
    // ---- C: remaining exponentials | O^T += V^T . P^T of tile it ----
    if (!(ABL & 8))
#pragma unroll
    for (int e = EXA; e < 32; ++e)
      pf[e >> 4][(e >> 3) & 1][e & 7] = from_f32<T>(pexp(s_cur[e >> 4][e & 15]));
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int row0 = t * 32 + h2 * 16;       // slots 0..3: kv = row0 + 4*hi + j; slots 4..7: kv = row0 + 8 + 4*hi + j
        {
          if (ABL & 2) oacc[0] = Vec<T>::mfma32(pf[t][h2 ^ 1], pf[t][h2], oacc[0]);
          else {
          const s16x4 lo = tr4(vbA + PAR * A40_VA_TILE + row0 * 64);
          const s16x4 hi4 = tr4(vbA + PAR * A40_VA_TILE + (row0 + 8) * 64);
          const s16x8 v = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
          oacc[0] = Vec<T>::mfma32(__builtin_bit_cast(V8, v), pf[t][h2], oacc[0]);
          }
        }
        if (!PV48) {
          if (ABL & 2) oacc[1] = Vec<T>::mfma32(pf[t ^ 1][h2], pf[t][h2], oacc[1]);
          else {
          const s16x4 lo = tr4(vbB + PAR * A40_VB_TILE + row0 * 16);
          const s16x4 hi4 = tr4(vbB + PAR * A40_VB_TILE + (row0 + 8) * 16);
          const s16x8 v = __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
          oacc[1] = Vec<T>::mfma32(__builtin_bit_cast(V8, v), pf[t][h2], oacc[1]);
          }
        }
      }
      if (PV48) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const u32x4 p0 = __builtin_bit_cast(u32x4, pf[t][0]), p1 = __builtin_bit_cast(u32x4, pf[t][1]);
        u32x4 bq0, bq1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const auto sw = __builtin_amdgcn_permlane16_swap(p0[d], p1[d], false, false);
          bq0[d] = sw[0]; bq1[d] = sw[1];
        }
        const s16x4 lo = tr4(vbB16 + PAR * A40_VB_TILE + t * 32 * 16);
        const s16x4 hi4 = tr4(vbB16 + PAR * A40_VB_TILE + (t * 32 + 8) * 16);
        const V8 v = __builtin_bit_cast(V8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
        o1a = Vec<T>::mfma16(v, __builtin_bit_cast(V8, bq0), o1a);
        o1b = Vec<T>::mfma16(v, __builtin_bit_cast(V8, bq1), o1b);
      }
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    // ---- D ----
    if (MORE) dma_v(std::integral_constant<int, 1 - PAR>{});
    if (it + 3 < nt) dma_k(std::integral_constant<int, 1 - PAR>{});
  };
  {
    using TT = std::true_type;
    using FF = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int it = 0;
    for (; it + 2 < nt; it += 2) {     // two tiles per trip: score registers and LDS stages alternate by name
      tile_step(TT{}, P0{}, it, s_a, s_b);
      tile_step(TT{}, P1{}, it + 1, s_b, s_a);
    }
    if (it + 2 == nt) {
      tile_step(TT{}, P0{}, it, s_a, s_b);
      tile_step(FF{}, P1{}, it + 1, s_b, s_a);
    } else if (it + 1 == nt) {
      tile_step(FF{}, P0{}, it, s_a, s_b);
    }
  }

  // ---- normalise and store: lane owns row q0+l31, d = db*32 + (r&3) + 8*(r>>2) + 4*hi ----
  float l_tot;
  if (PV48) {      // row 40 = row 8 of the 16-row block: register 0 of lane 32 + (query & 15), first / second query half
    const float lt_a = __shfl(o1a[0], 32 + (l31 & 15), 64), lt_b = __shfl(o1b[0], 32 + (l31 & 15), 64);
    l_tot = (l31 & 16) ? lt_b : lt_a;
  } else {
    l_tot = __shfl(oacc[L_DB][L_R], l31, 64);       // row 40 of O^T lives in the hi = 0 lane of column q
  }
  float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
  if (p.o_rowscale && q0 + l31 < p.Lq) inv *= p.o_rowscale[(long)(h / p.rs_hdiv) * p.rs_stride + (long)b * p.Lq + q0 + l31];
  {
    // pack to T; pairs of 4-column groups are merged with a half swap (v_permlane32_swap) so that every lane stores 16
    // contiguous bytes: the 8-byte stores of the first version cost 1.4x the output bytes in WRITE_SIZE (partial sectors)
    T* orow = Og + (long)min(q0 + l31, p.Lq - 1) * p.o_rs;
    const bool live = q0 + l31 < p.Lq;
    unsigned pk[5][2];
#pragma unroll
    for (int g = 0; g < (PV48 ? 4 : 5); ++g) {
      V4 w;
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = from_f32<T>(oacc[g >> 2][(g & 3) * 4 + j] * inv);     // d = 8g + 4hi + j
      const uint2 u = __builtin_bit_cast(uint2, w);
      pk[g][0] = u.x; pk[g][1] = u.y;
    }
#pragma unroll
    for (int g = 0; g < 4; g += 2) {     // lanes hi = 0: d 8g .. 8g+7, lanes hi = 1: d 8g+8 .. 8g+15
      const auto x = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
      const auto y = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
      if (live) *reinterpret_cast<uint4*>(orow + 8 * g + 8 * hi) = make_uint4(x[0], y[0], x[1], y[1]);
    }
    if (PV48) {
      // d 32 .. 39 of query qa = lane & 15 (o1a) and qa + 16 (o1b): lanes 0..15 hold d 32..35, lanes 16..31 d 36..39; lanes 0..15
      // fetch the partner's half and store 16 contiguous bytes per query
      const int qa = lane & 15;
      const float inv_a = __shfl(inv, qa, 64), inv_b = __shfl(inv, 16 + qa, 64);
      V4 wa, wb;
#pragma unroll
      for (int j = 0; j < 4; ++j) { wa[j] = from_f32<T>(o1a[j] * inv_a); wb[j] = from_f32<T>(o1b[j] * inv_b); }
      const uint2 ua = __builtin_bit_cast(uint2, wa), ub = __builtin_bit_cast(uint2, wb);
      const unsigned ax = __shfl(ua.x, 16 + qa, 64), ay = __shfl(ua.y, 16 + qa, 64);
      const unsigned bx = __shfl(ub.x, 16 + qa, 64), by = __shfl(ub.y, 16 + qa, 64);
      if (lane < 16) {
        if (q0 + qa < p.Lq) *reinterpret_cast<uint4*>(Og + (long)(q0 + qa) * p.o_rs + 32) = make_uint4(ua.x, ua.y, ax, ay);
        if (q0 + 16 + qa < p.Lq) *reinterpret_cast<uint4*>(Og + (long)(q0 + 16 + qa) * p.o_rs + 32) = make_uint4(ub.x, ub.y, bx, by);
      }
    } else {                             // d 32 .. 39: lanes hi = 0 collect the partner's half and store alone
      unsigned c0 = pk[4][0], c1 = pk[4][1];           // opaque copies: see the note at the row-maximum exchange
      asm volatile("" : "+v"(c0), "+v"(c1));
      const auto x = __builtin_amdgcn_permlane32_swap(pk[4][0], c0, false, false);
      const auto y = __builtin_amdgcn_permlane32_swap(pk[4][1], c1, false, false);
      if (live && !hi) *reinterpret_cast<uint4*>(orow + 32) = make_uint4(x[0], y[0], x[1], y[1]);
    }
  }
}

// hallo_set_option("attn40", v): 0 = attention.hip; 1 (default, and any other value in a product build) = this kernel in its round-6 form:
// 48-row PV (PV48), all 32 exponentials before the barrier, K fragments of the next tile requested ahead of the softmax statistics (KPRE).
// -DHALLO_ABLATIONS builds keep the earlier forms for A/B (tools/cbench attn-time / attn-det): 2 / 3 / 5 / 9 = 64-row PV with 0 / 32 / 8 / 16
// exponentials before the barrier (9 = the kernel of rounds 2-5), 4 = raised wave priority around the MFMA clusters, 6 / 7 / 8 = PV48 without
// KPRE (16 / 0 / 32), 10..21 timing ablations, 30..35 cache-policy bits of the K / V DMA, 36 / 37 = PV48 + KPRE (32 / 16).  (In the shared source
// the 64-row forms cost 4 spilled VGPRs since KPRE exists -- one more reason they are not in the product build; at 1024 queries the
// default form is now the faster one too: 34.7 vs 35.5 us.)
static int g_attn40_variant = 1;

void set_attn40_variant(int v) { g_attn40_variant = v; }

template <typename T>
static void launch_variant(const AttnArgs& a, dim3 grid, hipStream_t st) {
  switch (g_attn40_variant) {
#ifdef HALLO_ABLATIONS
    case 2: hipLaunchKernelGGL((attn40_kernel<T, 0, 0>), grid, dim3(256), 0, st, a); break;
    case 3: hipLaunchKernelGGL((attn40_kernel<T, 32, 0>), grid, dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((attn40_kernel<T, 16, 1>), grid, dim3(256), 0, st, a); break;
    case 5: hipLaunchKernelGGL((attn40_kernel<T, 8, 0>), grid, dim3(256), 0, st, a); break;
    case 8: hipLaunchKernelGGL((attn40_kernel<T, 32, 0, 0, true>), grid, dim3(256), 0, st, a); break;
    case 9: hipLaunchKernelGGL((attn40_kernel<T, 16, 0>), grid, dim3(256), 0, st, a); break;
    case 6: hipLaunchKernelGGL((attn40_kernel<T, 16, 0, 0, true>), grid, dim3(256), 0, st, a); break;
    case 7: hipLaunchKernelGGL((attn40_kernel<T, 0, 0, 0, true>), grid, dim3(256), 0, st, a); break;
    case 10: hipLaunchKernelGGL((attn40_kernel<T, 16, 0, 1>), grid, dim3(256), 0, st, a); break;
    case 11: hipLaunchKernelGGL((attn40_kernel<T, 16, 0, 2>), grid, dim3(256), 0, st, a); break;
    case 12: hipLaunchKernelGGL((attn40_kernel<T, 16, 0, 3>), grid, dim3(256), 0, st, a); break;
    case 13: hipLaunchKernelGGL((attn40_kernel<T, 16, 0, 4>), grid, dim3(256), 0, st, a); break;
    case 14: hipLaunchKernelGGL((attn40_kernel<T, 16, 0, 8>), grid, dim3(256), 0, st, a); break;
    case 15: hipLaunchKernelGGL((attn40_kernel<T, 16, 0, 16>), grid, dim3(256), 0, st, a); break;
    case 16: hipLaunchKernelGGL((attn40_kernel<T, 16, 0, 8 + 16>), grid, dim3(256), 0, st, a); break;
    case 17: hipLaunchKernelGGL((attn40_kernel<T, 16, 0, 3 + 8 + 16>), grid, dim3(256), 0, st, a); break;
    case 18: hipLaunchKernelGGL((attn40_kernel<T, 16, 0, 32>), grid, dim3(256), 0, st, a); break;
    case 19: hipLaunchKernelGGL((attn40_kernel<T, 16, 0, 32 + 3 + 8 + 16>), grid, dim3(256), 0, st, a); break;
    case 20: hipLaunchKernelGGL((attn40_kernel<T, 16, 0, 32 + 64>), grid, dim3(256), 0, st, a); break;
    case 21: hipLaunchKernelGGL((attn40_kernel<T, 16, 0, 2 + 128>), grid, dim3(256), 0, st, a); break;
    // cache-policy bits of the K / V DMA (aux operand of buffer_load ... lds): 1 = sc0, 2 = nt, 16 = sc1
    case 30: hipLaunchKernelGGL((attn40_kernel<T, 32, 0, 0, true, 1>), grid, dim3(256), 0, st, a); break;
    case 31: hipLaunchKernelGGL((attn40_kernel<T, 32, 0, 0, true, 2>), grid, dim3(256), 0, st, a); break;
    case 32: hipLaunchKernelGGL((attn40_kernel<T, 32, 0, 0, true, 3>), grid, dim3(256), 0, st, a); break;
    case 33: hipLaunchKernelGGL((attn40_kernel<T, 32, 0, 0, true, 16>), grid, dim3(256), 0, st, a); break;
    case 34: hipLaunchKernelGGL((attn40_kernel<T, 32, 0, 0, true, 17>), grid, dim3(256), 0, st, a); break;
    case 35: hipLaunchKernelGGL((attn40_kernel<T, 32, 0, 0, true, 18>), grid, dim3(256), 0, st, a); break;
    case 36: hipLaunchKernelGGL((attn40_kernel<T, 32, 0, 0, true, 0, true>), grid, dim3(256), 0, st, a); break;
    case 37: hipLaunchKernelGGL((attn40_kernel<T, 16, 0, 0, true, 0, true>), grid, dim3(256), 0, st, a); break;
#endif
    default: hipLaunchKernelGGL((attn40_kernel<T, 32, 0, 0, true, 0, true>), grid, dim3(256), 0, st, a); break;
  }
}

int launch_attn40(const AttnArgs& a, int dtype, hipStream_t st) {
  dim3 grid(a.batch * a.heads * a.nqb);
  if (dtype == DT_F16) launch_variant<_Float16>(a, grid, st);
  else if (dtype == DT_BF16) launch_variant<__bf16>(a, grid, st);
  else return -22;
  HALLO_CHECK_LAUNCH();
  return 0;
}

}  // namespace hallo
