// Kernel argument block shared by the GEMM / implicit-GEMM conv kernels (gemm.hip, gemm3.hip).
#pragma once
#include "common.h"

namespace hallo {

struct GemmArgs {
  const void* A; const void* B; void* C;
  int M, N, K;
  long lda, ldb, ldc;
  long sA, sB, sC;          // batch strides (elements)
  const void* bias;         // [N] (or [M] if bias_per_row) or null; for GEGLU: [2N]
  int bias_per_row;
  const void* bias2;        // [M/bias2_rpg, N] or null
  int bias2_rpg;
  long bias2_ld;
  const float* rowscale;    // [M] fp32 or null
  const void* residual;     // [M,N] (ld = ldr) or null
  long ldr, sR;
  float alpha;
  int lead_cols;            // output columns n < lead_cols get an extra factor lead_alpha (the q part of a fused q|k|v
  float lead_alpha;         // projection carries softmax scale * log2(e) so that the attention kernel can exp2 raw scores)
  const float* ln_colsum;   // fused LayerNorm (gemm2 only): A is the UN-normalised input, W = gamma * W, bias includes beta . W^T;
  const float* ln_stats;    // optional [M][2] fp32 (mean, rstd) from hallo_row_stats: then the K loop does no statistics work
  float ln_eps;             // the kernel accumulates per-row sum / sum of squares over K next to the MFMAs and applies
                            // out = rstd_m * (acc - mean_m * ln_colsum[n]) + bias.  ln_colsum[n] = sum_k W[n,k] (fp32, [N] / GEGLU [2N])
  int ln_parts;             // > 0: ln_stats is [M][ln_parts][2] partial (sum, sum of squares) -- reduced to mean / rstd per row in the prologue
  int ws_zeroed;            // the workspace's counter tail is known to be zero (hallo_gemm_desc.workspace_zeroed): gemm4.hip may split a tile's K loop
  float* row_parts;         // optional out: [M][ceil(N / 64)][2] (sum, sum of squares) of the rounded output rows per 64-column block
  int act;
  int out_f32;
  int tiles_n, tiles_m;
  int splits, nk_per_split;   // split-K: blockIdx.y = split, each split owns nk_per_split K tiles
  float* slab;                // fp32 partial sums [splits][M][N] (splits > 1)
  int slab_nt;                // hallo_set_option("splitk_nt"): 1 = non-temporal slab stores, 2 = + non-temporal loads in the reduce pass (A/B)
  int vec_ok, res_vec_ok, bias_vec_ok, bias2_vec_ok;   // 8-byte (fp32: 16-byte) row accesses are aligned
  // conv gather
  int H, W, Cin, OH, OW, stride, pad_t, pad_l, upsample;
  int conv_fast;   // Cin % 64 == 0 and no upsample: one tap per K tile, scalar tap offsets
  // hallo_gemm_desc.kv_out (gemm_rs2.hip only): columns >= kv_col0 leave as head-major [image][8 heads][kv_L rows][40] K / V tensors
  void* kv_out;
  int kv_col0, kv_L;
  long kv_tstride;
};

// gemm3.hip: 256x320 / 128x320 tile kernel (8 waves, LDS-DMA, one barrier per K step).  MODE 0 gemm, 1 conv3x3
// (fast gather only: Cin % 64 == 0, no upsample), 2 geglu.  TM = 32-row blocks per wave along M (2 or 1).
template <typename T> void launch_gemm3(const GemmArgs& a, int mode, int tm, int batch, hipStream_t st);

// gemm_rs.hip: row-stationary kernel for K = 320 / 640 (A rows of a workgroup in registers, LayerNorm statistics computed
// from them, W streamed through an LDS ring).  gemm_rs_eligible() is the routing rule of launch_gemm().
bool gemm_rs_eligible(const GemmArgs& a, bool conv, bool geglu, int batch);
template <typename T> int launch_gemm_rs(const GemmArgs& a, bool geglu, hipStream_t st);
void set_gemm_rs_dbg(int v);

// gemm_rs2.hip: the K = 320 row-stationary kernel with the epilogue of a W block pair interleaved into the next pair's MFMAs.
bool gemm_rs2_eligible(const GemmArgs& a, bool conv, bool geglu, int batch);
template <typename T> int launch_gemm_rs2(const GemmArgs& a, bool geglu, hipStream_t st);
void set_gemm_rs2_dbg(int v);

// gemm4.hip: exact-fit kernel for the mid-size problems (128 x 160 tiles, one persistent workgroup per CU = 4 multiplying + 4
// loader waves, 4-slot LDS-DMA ring; the tiles of a partly filled last round may split their K loop, partial tiles reduced in K
// order by the last arriver).  gemm4_plan() fills the schedule and says whether the kernel covers the problem (K % 64 == 0,
// bias2 row groups >= 128 rows, vector-aligned residual); the routing rule is in launch_gemm().
struct G4Sched {
  int tiles_m, tiles_n, nk;   // 128 x 160 tiles, K steps of 64
  int G;                      // persistent workgroups
  int dp;                     // data-parallel rounds: workgroup w owns tiles xcd_remap(j * G + w), j < dp
  int R, parts, per;          // tail: the last R tiles, each tile's K loop dealt over `parts` workgroups (`per` K steps each; parts = 1: whole tiles)
  float* part;                // [R * parts] partial tiles (fp32 register images, 80 KB each; parts > 1)
  int* cnt;                   // [R] arrival counters, zero between launches
  long long* dbg;             // optional: s_memtime stamps of workgroup 0 (tools/cbench)
};
void set_gemm4_debug_buffer(long long* p);
bool gemm4_plan(const GemmArgs& a, int64_t ws_bytes, G4Sched* out);
template <typename T> int launch_gemm4(const GemmArgs& a, G4Sched s, void* ws, int64_t ws_bytes, hipStream_t st);

// gemm_ff.hip: fused LayerNorm -> GEGLU -> net[2] -> + residual for C = 320 (hallo_ff320)
int ff_fused_variant();
void set_ff_fused_variant(int v);

}  // namespace hallo
