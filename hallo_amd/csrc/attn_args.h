// Kernel argument block shared by the attention kernels (attention.hip: generic head dims; attention40.hip: the
// LDS-DMA / transposing-read kernel for head dim 40).
#pragma once
#include <hip/hip_runtime.h>

namespace hallo {

struct AttnArgs {
  const void* q; const void* k1; const void* v1; const void* k2; const void* v2; void* o;
  int batch, heads, Lq, Lkv1, Lkv2;
  long q_bs, q_rs, k1_bs, k1_rs, v1_bs, v1_rs, k2_bs, k2_rs, v2_bs, v2_rs, o_bs, o_rs;
  int kv2_div, kv2_mod, kv2_first;
  float scale_log2e;
  int nqb;  // query blocks per (batch, head)
  const float* o_rowscale;   // optional fp32 output row scale: o[b, q, head h] *= o_rowscale[(h / rs_hdiv) * rs_stride + b * Lq + q]
  int rs_hdiv;
  long rs_stride;
  // workgroup order: 0 = query block fastest, then head, then batch (a head's K/V stays in one XCD's L2 while its query blocks
  // run); 1 = HEAD fastest (short K/V, e.g. the 32 audio / 4 face tokens: nothing to keep, but the heads of one query block
  // share the 128-byte lines of Q and O -- 80-byte head slices at head dim 40 -- so they should run together)
  int head_fastest;
  // head strides of K / V in elements (attention40.hip): head_dim = the heads of a row are adjacent (every caller today);
  // Lkv * head_dim with row stride head_dim = head-major K / V (a tile is one contiguous 5 KB: timing experiment, HALLO_ABLATIONS)
  long hs1, hs2;
};

// attention40.hip: head dim 40, pre-scaled q.  Returns 0 or a negative status like the other launchers.
int launch_attn40(const AttnArgs& a, int dtype, hipStream_t st);
void set_attn40_variant(int v);

}  // namespace hallo
