// cbench: torch-free kernel bench / determinism / reference harness over the C ABI of libhallo_amd.so.
//
// A fresh GPU box pays 1-2 minutes for the first `import torch`; this binary starts in a second, so a kernel A/B costs
// tens of seconds of box time instead of minutes.  It links the product library exactly as an external host would
// (include/hallo_amd.h, plain device pointers) and carries its own naive fp32 device references -- it is test tooling,
// nothing on the product path uses it.
//
//   tools/cbench/build.sh
//   tools/cbench/cbench attn-det                 hd-40 attention: run-to-run determinism per kernel variant + error vs fp32
//   tools/cbench/cbench attn-time [variants]     L0 spatial attention timings per variant (e.g. 0,1,2)
//   tools/cbench/cbench gemm M N K [geglu] [ln] [res] [rs=0|1] [variant=v]
//   tools/cbench/cbench gemm-suite               the step's dominant GEMM shapes
//   tools/cbench/cbench ff M [f16] [noln]        fused feed-forward (hallo_ff320) vs fp32 reference and vs the two-GEMM path
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include "../../include/hallo_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define HK(x) do { int s_ = (x); if (s_ != 0) { fprintf(stderr, "hallo status %d at %s:%d\n", s_, __FILE__, __LINE__); exit(3); } } while (0)

static const int DT_F16 = 0, DT_BF16 = 1;     // include/hallo_amd.h dtype codes

// ---------------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------------
__device__ inline uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ inline float to_f(uint16_t v, int dt) {
  if (dt == 1) return __uint_as_float((uint32_t)v << 16);
  _Float16 h = __builtin_bit_cast(_Float16, v);
  return (float)h;
}
__device__ inline uint16_t from_f(float f, int dt) {
  if (dt == 1) { __bf16 b = (__bf16)f; return __builtin_bit_cast(uint16_t, b); }
  _Float16 h = (_Float16)f;
  return __builtin_bit_cast(uint16_t, h);
}

__global__ void fill_normal(uint16_t* p, long n, uint32_t seed, float scale, float shift, int dt) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t a = hash32((uint32_t)i * 2u + seed), b = hash32((uint32_t)i * 2u + 1u + seed * 7919u);
  float u1 = ((a >> 8) + 1.0f) * (1.0f / 16777217.0f), u2 = (b >> 8) * (1.0f / 16777216.0f);
  float z = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
  p[i] = from_f(z * scale + shift, dt);
}
__global__ void fill_normal_f32(float* p, long n, uint32_t seed, float scale, float shift) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t a = hash32((uint32_t)i * 2u + seed), b = hash32((uint32_t)i * 2u + 1u + seed * 7919u);
  float u1 = ((a >> 8) + 1.0f) * (1.0f / 16777217.0f), u2 = (b >> 8) * (1.0f / 16777216.0f);
  p[i] = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2) * scale + shift;
}

// naive attention reference: one thread per (b, h, q); q is pre-scaled by hd^-1/2 * log2(e) (exp2 domain)
__global__ void attn_ref(const uint16_t* q, const uint16_t* k1, const uint16_t* v1, const uint16_t* k2, const uint16_t* v2,
                         float* o, int B, int H, int HD, int Lq, int L1, int L2, long q_bs, long q_rs, long kv_bs, long kv_rs,
                         long k2_rs, int dt) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)B * H * Lq) return;
  int qi = i % Lq; int h = (i / Lq) % H; int b = i / ((long)Lq * H);
  float qv[160];
  for (int d = 0; d < HD; ++d) qv[d] = to_f(q[b * q_bs + qi * q_rs + h * HD + d], dt);
  float m = -3e38f, l = 0.0f, acc[160];
  for (int d = 0; d < HD; ++d) acc[d] = 0.0f;
  for (int seg = 0; seg < 2; ++seg) {
    const uint16_t* K = seg ? k2 : k1 + b * kv_bs; const uint16_t* V = seg ? v2 : v1 + b * kv_bs;
    const int L = seg ? L2 : L1; const long rs = seg ? k2_rs : kv_rs;
    if (!K) continue;
    for (int j = 0; j < L; ++j) {
      float s = 0.0f;
      for (int d = 0; d < HD; ++d) s += qv[d] * to_f(K[j * rs + h * HD + d], dt);
      if (s > m) { float a = exp2f(m - s); l *= a; for (int d = 0; d < HD; ++d) acc[d] *= a; m = s; }
      float p = exp2f(s - m);
      l += p;
      for (int d = 0; d < HD; ++d) acc[d] += p * to_f(V[j * rs + h * HD + d], dt);
    }
  }
  for (int d = 0; d < HD; ++d) o[((long)b * Lq + qi) * (H * HD) + h * HD + d] = acc[d] / l;
}

// naive GEMM reference (fp32): C = A . W^T (+ bias), optional LayerNorm on A rows (fp32 statistics), optional GEGLU
__global__ void gemm_ref(const uint16_t* A, const uint16_t* W, const uint16_t* bias, const uint16_t* gamma, const uint16_t* beta,
                         const uint16_t* res, float* C, int M, int N, int K, int geglu, int ln, int dt, int rows, int row0) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)rows * N) return;
  int n = i % N; int m = row0 + i / N;
  float mean = 0.0f, rstd = 1.0f;
  if (ln) {
    float s = 0.0f, s2 = 0.0f;
    for (int k = 0; k < K; ++k) { float x = to_f(A[(long)m * K + k], dt); s += x; s2 += x * x; }
    mean = s / K; rstd = rsqrtf(fmaxf(s2 / K - mean * mean, 0.0f) + 1e-5f);
  }
  auto dot = [&](int wr) {
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) {
      float x = to_f(A[(long)m * K + k], dt);
      if (ln) x = (x - mean) * rstd * to_f(gamma[k], dt) + to_f(beta[k], dt);
      acc += x * to_f(W[(long)wr * K + k], dt);
    }
    return acc + (bias ? to_f(bias[wr], dt) : 0.0f);
  };
  float v = dot(n);
  if (geglu) { float g = dot(N + n); v = v * 0.5f * g * (1.0f + erff(g * 0.70710678f)); }
  if (res) v += to_f(res[(long)m * N + n], dt);
  C[i] = v;
}

// W' = W * gamma (rounded), colsum[n] = sum_k W'[n,k] (fp32), bias'[n] = bias[n] + beta . W[n,:]   (hallo_gemm's LN contract)
__global__ void fold_ln(const uint16_t* W, const uint16_t* bias, const uint16_t* gamma, const uint16_t* beta, uint16_t* Wf,
                        uint16_t* bf, float* cs, int rows, int K, int dt) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= rows) return;
  float s = 0.0f, bb = bias ? to_f(bias[n], dt) : 0.0f;
  for (int k = 0; k < K; ++k) {
    float w = to_f(W[(long)n * K + k], dt);
    uint16_t wf = from_f(w * to_f(gamma[k], dt), dt);
    Wf[(long)n * K + k] = wf;
    s += to_f(wf, dt);
    bb += to_f(beta[k], dt) * w;
  }
  cs[n] = s; bf[n] = from_f(bb, dt);
}

// ---------------------------------------------------------------------------------------------------------------------
template <typename T> static T* dalloc(long n) { void* p; CK(hipMalloc(&p, n * sizeof(T))); return (T*)p; }
static void fill(uint16_t* p, long n, uint32_t seed, float scale, float shift, int dt) {
  hipLaunchKernelGGL(fill_normal, dim3((n + 255) / 256), dim3(256), 0, 0, p, n, seed, scale, shift, dt);
}
static float host_to_f(uint16_t v, int dt) {
  if (dt == 1) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
  // fp16
  uint32_t s = (v >> 15) & 1, e = (v >> 10) & 31, m = v & 1023; float f;
  if (e == 0) f = ldexpf((float)m, -24); else if (e == 31) f = m ? NAN : INFINITY; else f = ldexpf((float)(m + 1024), (int)e - 25);
  return s ? -f : f;
}
struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  template <typename F> float run(F f, int iters) {
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.0f / iters;      // us
  }
};
static int g_rounds = 5, g_iters = 10;      // timing loops; "quick" (PMC passes) -> 1 x 3
static float median(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

// ---------------------------------------------------------------------------------------------------------------------
// attention
// ---------------------------------------------------------------------------------------------------------------------
struct AttnCase { const char* name; int B, L, bank; };

static hallo_attn_desc attn_desc(const uint16_t* qkv, const uint16_t* bkv, uint16_t* o, int B, int L, int H, int HD, int bank, int dt) {
  const int C = H * HD;
  hallo_attn_desc d; memset(&d, 0, sizeof d);
  d.q = qkv; d.k1 = qkv + C; d.v1 = qkv + 2 * C; d.o = o;
  d.batch = B; d.heads = H; d.head_dim = HD; d.Lq = L; d.Lkv1 = L;
  d.q_bs = d.k1_bs = d.v1_bs = (long)L * 3 * C; d.q_rs = d.k1_rs = d.v1_rs = 3 * C;
  d.o_bs = (long)L * C; d.o_rs = C;
  // bank = number of reference banks (clips in the batch): frame row b reads bank b / (B / bank) -- one clip: every row bank 0
  if (bank) { d.k2 = bkv; d.v2 = bkv + C; d.Lkv2 = L; d.k2_bs = d.v2_bs = (long)L * 2 * C; d.k2_rs = d.v2_rs = 2 * C; d.kv2_batch_div = B / bank; }
  else d.kv2_batch_div = 1;
  d.kv2_batch_mod = 0; d.kv2_first_batch = 0;
  d.scale = 1.0f / sqrtf((float)HD); d.dtype = dt; d.q_prescaled = 1;
  return d;
}

static int cmd_attn_det(int argc, char** argv) {
  const int H = 8, HD = 40, C = H * HD;
  const AttnCase cases[] = {{"4x1024 self (the pytest case, unscaled q)", 4, 1024, 0}, {"4x1024 self+bank", 4, 1024, 1},
                            {"2x4096 self+bank", 2, 4096, 1}, {"3x1000 ragged self+bank", 3, 1000, 1}};
  std::vector<int> variants = {0, 1, 2, 3, 4, 5, 6};
  if (argc > 0) { variants.clear(); for (char* t = strtok(argv[0], ","); t; t = strtok(nullptr, ",")) variants.push_back(atoi(t)); }
  int bad = 0;
  for (int dt = 0; dt < 2; ++dt)
    for (const AttnCase& c : cases) {
      const long nq = (long)c.B * c.L * 3 * C, nb = (long)(c.bank ? c.bank : 1) * c.L * 2 * C, no = (long)c.B * c.L * C;
      uint16_t* qkv = dalloc<uint16_t>(nq); uint16_t* bkv = dalloc<uint16_t>(nb); uint16_t* o = dalloc<uint16_t>(no);
      float* ref = dalloc<float>(no);
      const bool first = (&c == &cases[0]);
      fill(qkv, nq, 11 + dt, 1.0f, 0.0f, dt); fill(bkv, nb, 23 + dt, 1.0f, 0.0f, dt);
      if (!first) {   // realistic score range: scale the q columns by hd^-1/2 * log2 e (done on the host for simplicity)
        std::vector<uint16_t> h(nq); CK(hipMemcpy(h.data(), qkv, nq * 2, hipMemcpyDeviceToHost));
        // re-fill q columns only
        uint16_t* tmp = dalloc<uint16_t>(nq); fill(tmp, nq, 77 + dt, 0.2281f, 0.0f, dt);
        CK(hipMemcpy2D(qkv, 3 * C * 2, tmp, 3 * C * 2, C * 2, (size_t)c.B * c.L, hipMemcpyDeviceToDevice));
        CK(hipFree(tmp));
      }
      hipLaunchKernelGGL(attn_ref, dim3(((long)c.B * H * c.L + 63) / 64), dim3(64), 0, 0, qkv, qkv + C, qkv + 2 * C,
                         c.bank ? bkv : nullptr, c.bank ? bkv + C : nullptr, ref, c.B, H, HD, c.L, c.L, c.bank ? c.L : 0,
                         (long)c.L * 3 * C, (long)3 * C, (long)c.L * 3 * C, (long)3 * C, (long)2 * C, dt);
      std::vector<float> href(no); CK(hipMemcpy(href.data(), ref, no * 4, hipMemcpyDeviceToHost));
      double refn = 0; for (long i = 0; i < no; ++i) refn += (double)href[i] * href[i];
      for (int v : variants) {
        HK(hallo_set_option("attn40", v));
        hallo_attn_desc d = attn_desc(qkv, bkv, o, c.B, c.L, H, HD, c.bank, dt);
        std::vector<uint16_t> first_out(no), out(no);
        long mism_total = 0, nan = 0; int runs_bad = 0;
        long hist_d[40] = {0}, hist_r[128] = {0}; long hist_b[16] = {0};
        for (int r = 0; r < 8; ++r) {
          CK(hipMemset(o, 0xFF, no * 2));
          HK(hallo_attention(&d, nullptr));
          CK(hipDeviceSynchronize());
          CK(hipMemcpy(out.data(), o, no * 2, hipMemcpyDeviceToHost));
          if (r == 0) { first_out = out; continue; }
          long mm = 0;
          for (long i = 0; i < no; ++i) if (out[i] != first_out[i]) {
            ++mm; hist_d[i % 40]++; hist_r[(i / C) % c.L % 128]++; hist_b[std::min<long>(15, i / ((long)c.L * C))]++;
          }
          if (mm) ++runs_bad;
          mism_total += mm;
        }
        double en = 0;
        for (long i = 0; i < no; ++i) { float x = host_to_f(first_out[i], dt); if (x != x) ++nan; double e = x - href[i]; en += e * e; }
        printf("attn-det dt=%s case='%s' attn40=%d: rel_l2_vs_fp32=%.3e nan=%ld nondeterministic_runs=%d/7 mismatched_elems=%ld\n",
               dt ? "bf16" : "f16", c.name, v, sqrt(en / refn), nan, runs_bad, mism_total);
        if (mism_total) {
          ++bad;
          printf("   by d:"); for (int i = 0; i < 40; ++i) printf(" %ld", hist_d[i]); printf("\n   by row%%128:");
          for (int i = 0; i < 128; ++i) printf(" %ld", hist_r[i]); printf("\n   by batch:"); for (int i = 0; i < 16; ++i) printf(" %ld", hist_b[i]);
          printf("\n");
        }
        fflush(stdout);
      }
      CK(hipFree(qkv)); CK(hipFree(bkv)); CK(hipFree(o)); CK(hipFree(ref));
    }
  HK(hallo_set_option("attn40", 1));
  return bad ? 1 : 0;
}

static int cmd_attn_time(int argc, char** argv) {
  std::vector<int> variants = {0, 1};
  if (argc > 0) { variants.clear(); for (char* t = strtok(argv[0], ","); t; t = strtok(nullptr, ",")) variants.push_back(atoi(t)); }
  const int H = 8, HD = 40, C = H * HD;
  const AttnCase cases[] = {{"L0 self+bank 16x4096x(4096+4096)", 16, 4096, 1}, {"L0 audio-block self 16x4096x4096", 16, 4096, 0},
                            {"256^2 L0 self+bank 8x1024x(1024+1024)", 8, 1024, 1},
                            // round 6: a batch of four clips through one evaluation (FaceAnimatePipeline.call_batch): 64 frame rows, 4 banks
                            {"L0 self+bank, 4 clips: 64x4096x(4096+4096)", 64, 4096, 4}, {"L0 audio-block self, 4 clips: 64x4096x4096", 64, 4096, 0}};
  Timer tm;
  const char* only_case = getenv("CBENCH_CASE");      // PMC passes: one shape (0..4), bf16 only
  for (int dt = 1; dt >= 0; --dt)
    for (const AttnCase& c : cases) {
      if (only_case && (dt != 1 || (int)(&c - cases) != atoi(only_case))) continue;
      const long nq = (long)c.B * c.L * 3 * C, nb = (long)(c.bank ? c.bank : 1) * c.L * 2 * C, no = (long)c.B * c.L * C;
      uint16_t* qkv = dalloc<uint16_t>(nq); uint16_t* bkv = dalloc<uint16_t>(nb); uint16_t* o = dalloc<uint16_t>(no);
      fill(qkv, nq, 11 + dt, 1.0f, 0.0f, dt); fill(bkv, nb, 23 + dt, 1.0f, 0.0f, dt);
      uint16_t* tmp = dalloc<uint16_t>(nq); fill(tmp, nq, 77 + dt, 0.2281f, 0.0f, dt);
      CK(hipMemcpy2D(qkv, 3 * C * 2, tmp, 3 * C * 2, C * 2, (size_t)c.B * c.L, hipMemcpyDeviceToDevice));
      CK(hipFree(tmp));
      hallo_attn_desc d = attn_desc(qkv, bkv, o, c.B, c.L, H, HD, c.bank, dt);
      uint16_t *kh = nullptr, *vh = nullptr, *k2h = nullptr, *v2h = nullptr;
      if (getenv("CBENCH_KV_HM")) {     // head-major K / V ([batch][head][row][40], ABI v9 kv1_hs / kv2_hs): what the pipeline launches since round 6
        const long n1 = (long)c.B * c.L * C, n2 = (long)(c.bank ? c.bank : 1) * c.L * C;
        kh = dalloc<uint16_t>(n1); vh = dalloc<uint16_t>(n1); fill(kh, n1, 31 + dt, 1.0f, 0.0f, dt); fill(vh, n1, 37 + dt, 1.0f, 0.0f, dt);
        d.k1 = kh; d.v1 = vh; d.k1_rs = d.v1_rs = HD; d.k1_bs = d.v1_bs = (long)H * c.L * HD; d.kv1_hs = (long)c.L * HD;
        if (c.bank) {
          k2h = dalloc<uint16_t>(n2); v2h = dalloc<uint16_t>(n2); fill(k2h, n2, 41 + dt, 1.0f, 0.0f, dt); fill(v2h, n2, 43 + dt, 1.0f, 0.0f, dt);
          d.k2 = k2h; d.v2 = v2h; d.k2_rs = d.v2_rs = HD; d.k2_bs = d.v2_bs = (long)H * c.L * HD; d.kv2_hs = (long)c.L * HD;
        }
      }
      const double flop = 4.0 * C * c.L * c.B * ((double)c.L * (c.bank ? 2 : 1));
      std::vector<std::vector<float>> t(variants.size());
      for (size_t i = 0; i < variants.size(); ++i) { HK(hallo_set_option("attn40", variants[i])); tm.run([&] { HK(hallo_attention(&d, nullptr)); }, 3); }
      for (int r = 0; r < g_rounds; ++r)
        for (size_t i = 0; i < variants.size(); ++i) {
          HK(hallo_set_option("attn40", variants[i]));
          t[i].push_back(tm.run([&] { HK(hallo_attention(&d, nullptr)); }, g_iters));
        }
      for (size_t i = 0; i < variants.size(); ++i) {
        const float us = median(t[i]);
        printf("attn-time dt=%s case='%s'%s attn40=%d: %.1f us  %.1f TFLOP/s\n", dt ? "bf16" : "f16", c.name, kh ? " [head-major K/V]" : "", variants[i], us, flop / us / 1e6);
      }
      fflush(stdout);
      CK(hipFree(qkv)); CK(hipFree(bkv)); CK(hipFree(o));
      if (kh) { CK(hipFree(kh)); CK(hipFree(vh)); }
      if (k2h) { CK(hipFree(k2h)); CK(hipFree(v2h)); }
    }
  HK(hallo_set_option("attn40", 1));
  return 0;
}


// attn-nan: locate non-finite output rows of the hd-40 kernel on the unscaled-q case and print the row's score profile
static int cmd_attn_nan(int argc, char** argv) {
  const int H = 8, HD = 40, C = H * HD, B = 4, L = 1024, dt = argc > 0 ? atoi(argv[0]) : 0;
  const long nq = (long)B * L * 3 * C, no = (long)B * L * C;
  uint16_t* qkv = dalloc<uint16_t>(nq); uint16_t* o = dalloc<uint16_t>(no);
  fill(qkv, nq, 11 + dt, 1.0f, 0.0f, dt);
  hallo_attn_desc d = attn_desc(qkv, nullptr, o, B, L, H, HD, 0, dt);
  HK(hallo_set_option("attn40", 1));
  HK(hallo_attention(&d, nullptr)); CK(hipDeviceSynchronize());
  std::vector<uint16_t> hq(nq), ho(no);
  CK(hipMemcpy(hq.data(), qkv, nq * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ho.data(), o, no * 2, hipMemcpyDeviceToHost));
  int shown = 0; long rows_bad = 0;
  for (int b = 0; b < B; ++b) for (int qi = 0; qi < L; ++qi) for (int h = 0; h < H; ++h) {
    int nbad = 0;
    for (int dd = 0; dd < HD; ++dd) { float x = host_to_f(ho[((long)b * L + qi) * C + h * HD + dd], dt); if (!(x == x) || fabsf(x) > 1e30f) ++nbad; }
    if (!nbad) continue;
    ++rows_bad;
    if (shown++ >= 12) continue;
    printf("bad row b=%d q=%d (q%%128=%d, lane=%d wave=%d) h=%d bad_d=%d | tile maxima (log2 domain):", b, qi, qi % 128, qi % 32, (qi % 128) / 32, h, nbad);
    float run = -1e30f;
    for (int t = 0; t < L / 64; ++t) {
      float mx = -1e30f;
      for (int j = 0; j < 64; ++j) {
        float sc = 0;
        for (int dd = 0; dd < HD; ++dd) sc += host_to_f(hq[((long)b * L + qi) * 3 * C + h * HD + dd], dt) * host_to_f(hq[((long)b * L + t * 64 + j) * 3 * C + C + h * HD + dd], dt);
        mx = fmaxf(mx, sc);
      }
      printf(" %.1f", mx);
      run = fmaxf(run, mx);
    }
    printf(" | row max %.1f\n", run);
  }
  printf("attn-nan dt=%d: %ld bad rows\n", dt, rows_bad);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// GEMM
// ---------------------------------------------------------------------------------------------------------------------
extern "C" void hallo_gemm4_debug_buffer(long long* p) __attribute__((weak));
struct GemmOpts { bool stamps = false; int M, N, K; bool geglu = false, ln = false, res = false; int rs = -1, variant = -1, rsdbg = 0, g4 = -1; float ascale = 1.0f, wscale = 1.0f; int dt = DT_BF16; bool check = true; };

static void run_gemm_case(const GemmOpts& g, Timer& tm) {
  const int wrows = g.geglu ? 2 * g.N : g.N, dt = g.dt;
  uint16_t* A = dalloc<uint16_t>((long)g.M * g.K); uint16_t* W = dalloc<uint16_t>((long)wrows * g.K);
  uint16_t* Wf = dalloc<uint16_t>((long)wrows * g.K); uint16_t* bias = dalloc<uint16_t>(wrows); uint16_t* bf = dalloc<uint16_t>(wrows);
  uint16_t* gamma = dalloc<uint16_t>(g.K); uint16_t* beta = dalloc<uint16_t>(g.K); float* cs = dalloc<float>(wrows);
  uint16_t* Cc = dalloc<uint16_t>((long)g.M * g.N); uint16_t* R = g.res ? dalloc<uint16_t>((long)g.M * g.N) : nullptr;
  float* stats = dalloc<float>((long)g.M * 2);
  fill(A, (long)g.M * g.K, 1, g.ascale, 0.3f * g.ascale, dt); fill(W, (long)wrows * g.K, 2, g.wscale / sqrtf((float)g.K), 0.0f, dt);
  fill(bias, wrows, 3, 1.0f, 0.0f, dt); fill(gamma, g.K, 4, 0.1f, 1.0f, dt); fill(beta, g.K, 5, 0.1f, 0.0f, dt);
  if (R) fill(R, (long)g.M * g.N, 6, 1.0f, 0.0f, dt);
  if (g.ln) hipLaunchKernelGGL(fold_ln, dim3((wrows + 63) / 64), dim3(64), 0, 0, W, bias, gamma, beta, Wf, bf, cs, wrows, g.K, dt);
  if (g.rs >= 0) HK(hallo_set_option("gemm_rs", g.rs));
  if (g.variant >= 0) HK(hallo_set_option("gemm_variant", g.variant));
  if (g.g4 >= 0) HK(hallo_set_option("gemm4", g.g4));
  HK(hallo_set_option("gemm_rs_dbg", g.rsdbg));
  hallo_gemm_desc d; memset(&d, 0, sizeof d);
  d.A = A; d.B = g.ln ? Wf : W; d.C = Cc; d.M = g.M; d.N = g.N; d.K = g.K; d.lda = g.K; d.ldb = g.K; d.ldc = g.N; d.batch = 1;
  d.bias = g.ln ? bf : bias; d.residual = R; d.ldr = g.N; d.alpha = 1.0f; d.geglu = g.geglu; d.dtype = dt; d.lead_alpha = 1.0f;
  if (g.ln) { d.ln_colsum = cs; d.ln_eps = 1e-5f; }
  static void* ws = nullptr;                       // split-K scratch, as hallo_amd/ops.py hands one to every GEMM
  const long ws_bytes = 256L << 20;
  if (!ws) { CK(hipMalloc(&ws, ws_bytes)); CK(hipMemset(ws, 0, ws_bytes)); }      // zero-initialised: the stream-K kernel's arrival counters
  d.workspace = ws; d.workspace_bytes = ws_bytes; d.workspace_zeroed = 1;
  const bool fused_stats = g.ln && hallo_gemm_fuses_row_stats(g.M, g.N, g.K, g.geglu, 0, 0);
  if (g.ln && !fused_stats) d.ln_stats = stats;
  auto launch = [&] {
    if (g.ln && !fused_stats) HK(hallo_row_stats(A, stats, g.M, g.K, 1e-5f, dt, nullptr));
    HK(hallo_gemm(&d, nullptr));
  };
  // g4 > 0: the same problem on the kernels it replaces first (gemm4 = 0), for a FULL-matrix comparison of the stream-K kernel
  std::vector<uint16_t> base_out;
  int base_kern = 0;
  if (g.g4 > 0) {
    HK(hallo_set_option("gemm4", 0));
    launch(); CK(hipDeviceSynchronize());
    base_kern = hallo_get_option("last_gemm_kernel");
    base_out.resize((long)g.M * g.N);
    CK(hipMemcpy(base_out.data(), Cc, base_out.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemset(Cc, 0xFF, (long)g.M * g.N * 2));
    HK(hallo_set_option("gemm4", g.g4));
  }
  launch(); CK(hipDeviceSynchronize());
  const int kern = hallo_get_option("last_gemm_kernel");
  double rel = -1; long nan = 0; int nondet = 0;
  if (!base_out.empty()) {
    std::vector<uint16_t> o4((long)g.M * g.N);
    CK(hipMemcpy(o4.data(), Cc, o4.size() * 2, hipMemcpyDeviceToHost));
    double en = 0, rn = 0, mx = 0; long differ = 0, bad = 0;
    for (long i = 0; i < (long)g.M * g.N; ++i) {
      const float x = host_to_f(o4[i], dt), y = host_to_f(base_out[i], dt);
      if (x != x) ++bad;
      const double e = (double)x - y; en += e * e; rn += (double)y * y; if (fabs(e) > mx) mx = fabs(e); if (o4[i] != base_out[i]) ++differ;
    }
    printf("  full matrix vs kernel %d: rel_l2=%.3e max_abs=%.3e elements differing=%ld of %ld nan=%ld\n", base_kern, sqrt(en / rn), mx, differ, (long)g.M * g.N, bad);
  }
  if (g.check) {
    const int rows = std::min(g.M, 512);
    float* ref = dalloc<float>((long)rows * g.N);
    std::vector<float> href((long)rows * g.N); std::vector<uint16_t> out((long)g.M * g.N), out2((long)g.M * g.N);
    CK(hipMemcpy(out.data(), Cc, out.size() * 2, hipMemcpyDeviceToHost));
    double en = 0, rn = 0;
    for (int part = 0; part < 2; ++part) {          // the first and the last `rows` rows
      const int row0 = part ? g.M - rows : 0;
      hipLaunchKernelGGL(gemm_ref, dim3(((long)rows * g.N + 255) / 256), dim3(256), 0, 0, A, W, bias, gamma, beta, R, ref, g.M, g.N, g.K,
                         (int)g.geglu, (int)g.ln, dt, rows, row0);
      CK(hipMemcpy(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost));
      for (long i = 0; i < (long)rows * g.N; ++i) { float x = host_to_f(out[(long)row0 * g.N + i], dt); if (x != x) ++nan; double e = x - href[i]; en += e * e; rn += (double)href[i] * href[i]; }
    }
    // also the last 256 rows region sanity: NaN scan over everything
    for (long i = 0; i < (long)g.M * g.N; ++i) { float x = host_to_f(out[i], dt); if (x != x) ++nan; }
    rel = sqrt(en / rn);
    for (int r = 0; r < 3; ++r) { launch(); CK(hipDeviceSynchronize()); CK(hipMemcpy(out2.data(), Cc, out2.size() * 2, hipMemcpyDeviceToHost)); if (memcmp(out.data(), out2.data(), out.size() * 2)) ++nondet; }
    CK(hipFree(ref));
  }
  tm.run(launch, 3);
  std::vector<float> t; for (int r = 0; r < g_rounds; ++r) t.push_back(tm.run(launch, g_iters));
  const float us = median(t);
  const double flop = 2.0 * g.M * (double)wrows * g.K, bytes = 2.0 * ((double)g.M * g.K + (double)wrows * g.K + (double)g.M * g.N * (g.res ? 2 : 1));
  printf("gemm M=%d N=%d K=%d%s%s%s dt=%s rs=%d variant=%d g4=%d rsdbg=%d kernel=%d splits=%d fused_stats=%d: %.1f us  %.1f TFLOP/s  %.0f GB/s  rel_l2(first rows)=%.2e nan=%ld nondet=%d\n",
         g.M, g.N, g.K, g.geglu ? " geglu" : "", g.ln ? " ln" : "", g.res ? " res" : "", dt ? "bf16" : "f16", g.rs, g.variant, g.g4, g.rsdbg, kern, hallo_get_option("last_gemm_splits"), (int)fused_stats,
         us, flop / us / 1e6, bytes / us / 1e3, rel, nan, nondet);
  if (g.stamps && hallo_gemm4_debug_buffer) {
    // s_memtime stamps of workgroup 0 / wave 0 of gemm4.hip (100 MHz constant clock on gfx950: 1 tick = 10 ns): [0] start, [1] first
    // K step visible, [2 + k] end of K step k (k < 24), [30] K loop done, [31] epilogue done
    long long* dbg = dalloc<long long>(64); CK(hipMemset(dbg, 0, 64 * 8));
    hallo_gemm4_debug_buffer(dbg);
    launch(); CK(hipDeviceSynchronize());
    hallo_gemm4_debug_buffer(nullptr);
    long long h[64]; CK(hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost));
    printf("  stamps (ticks from start): first step visible %lld |", h[1] - h[0]);
    for (int k = 0; k < 24 && h[2 + k]; ++k) printf(" %lld", h[2 + k] - h[0]);
    printf(" | loop done %lld | epilogue: constants + residual loads issued %lld, 16-row blocks done %lld %lld %lld %lld | epilogue done %lld\n", h[30] - h[0],
           h[32] - h[0], h[33] - h[0], h[34] - h[0], h[35] - h[0], h[36] - h[0], h[31] - h[0]);
    CK(hipFree(dbg));
  }
  fflush(stdout);
  CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(Wf)); CK(hipFree(bias)); CK(hipFree(bf)); CK(hipFree(gamma)); CK(hipFree(beta)); CK(hipFree(cs));
  CK(hipFree(Cc)); if (R) CK(hipFree(R)); CK(hipFree(stats));
  if (g.rs >= 0) HK(hallo_set_option("gemm_rs", 1));
  if (g.variant >= 0) HK(hallo_set_option("gemm_variant", 6));
  if (g.g4 >= 0) HK(hallo_set_option("gemm4", 1));
  HK(hallo_set_option("gemm_rs_dbg", 0));
}

static GemmOpts parse_gemm(int argc, char** argv) {
  GemmOpts g; g.M = atoi(argv[0]); g.N = atoi(argv[1]); g.K = atoi(argv[2]);
  for (int i = 3; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "geglu") g.geglu = true; else if (a == "ln") g.ln = true; else if (a == "res") g.res = true; else if (a == "nocheck") g.check = false;
    else if (a == "f16") g.dt = DT_F16; else if (a.rfind("rs=", 0) == 0) g.rs = atoi(a.c_str() + 3); else if (a.rfind("variant=", 0) == 0) g.variant = atoi(a.c_str() + 8); else if (a.rfind("g4=", 0) == 0) g.g4 = atoi(a.c_str() + 3); else if (a == "stamps") g.stamps = true;
    else if (a == "zeroA") { g.ascale = 0.0f; g.check = false; } else if (a == "zeroW") { g.wscale = 0.0f; g.check = false; }
    else if (a.rfind("rsdbg=", 0) == 0) { g.rsdbg = atoi(a.c_str() + 6); if (g.rsdbg) g.check = false; }
  }
  return g;
}

// ---------------------------------------------------------------------------------------------------------------------
// fused feed-forward (hallo_ff320): y = x + net2(GEGLU(net0(LN(x)))) for C = 320, against a two-stage fp32 reference and the
// two-GEMM path it replaces
// ---------------------------------------------------------------------------------------------------------------------
__global__ void ff2_ref(const float* H, const uint16_t* W2, const uint16_t* b2, const uint16_t* X, float* Y, int rows, int row0, int C, int I, int dt) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)rows * C) return;
  int n = i % C; int r = i / C;
  float acc = 0.0f;
  for (int c = 0; c < I; ++c) acc += H[(long)r * I + c] * to_f(W2[(long)n * I + c], dt);
  Y[i] = acc + to_f(b2[n], dt) + to_f(X[(long)(row0 + r) * C + n], dt);
}

// host packer: mirrors hallo_amd/ops.py ff320_pack (include/hallo_amd.h documents the image)
static uint16_t host_from_f(float f, int dt) {
  if (dt == 1) { uint32_t u; memcpy(&u, &f, 4); uint32_t r = u + 0x7FFFu + ((u >> 16) & 1u); return (uint16_t)(r >> 16); }
  _Float16 h = (_Float16)f; uint16_t v; memcpy(&v, &h, 2); return v;
}
static std::vector<uint8_t> ff_pack(const std::vector<uint16_t>& w1f, const std::vector<uint16_t>& b1f, const std::vector<uint16_t>& w2, int dt) {
  const int C = 320, I = 1280, NS = 80, CH = 32768;
  const float INV = 1.17741002251547469101f, SC = 0.84932180028801904272f;
  std::vector<uint8_t> img((size_t)NS * CH, 0);
  for (int s = 0; s < NS; ++s) {
    uint8_t* base = img.data() + (size_t)s * CH;
    const int c0 = 16 * s;
    for (int t = 0; t < 5; ++t) for (int r = 0; r < 32; ++r) for (int pc = 0; pc < 8; ++pc) {
      const int src_row = r < 16 ? c0 + r : I + c0 + (r - 16);
      const uint16_t* src = &w1f[(size_t)src_row * C + t * 64 + pc * 8];
      memcpy(base + t * 4096 + r * 128 + ((pc ^ ((r >> 1) & 7)) * 16), src, 16);
    }
    for (int n = 0; n < C; ++n) for (int h = 0; h < 2; ++h) for (int e = 0; e < 8; ++e) {
      const int c = c0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      memcpy(base + 20480 + n * 32 + ((h ^ ((n >> 3) & 1)) * 16) + e * 2, &w2[(size_t)n * I + c], 2);
    }
    float* bp = reinterpret_cast<float*>(base + 30720);
    for (int h = 0; h < 2; ++h) for (int e = 0; e < 8; ++e) {
      const int c = c0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      bp[h * 16 + e] = host_to_f(b1f[c], dt) * INV;
      bp[h * 16 + 8 + e] = host_to_f(b1f[I + c], dt) * SC;
    }
  }
  return img;
}

extern "C" void hallo_ff320_debug_buffer(long long* p) __attribute__((weak));

static int cmd_ff(int argc, char** argv) {
  const int C = 320, I = 1280;
  int M = argc > 0 ? atoi(argv[0]) : 65536, dt = DT_BF16; bool ln = true;
  std::vector<int> variants = {1};
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "f16") dt = DT_F16; else if (a == "noln") ln = false;
    else if (a.rfind("v=", 0) == 0) { variants.clear(); for (char* tok = strtok(argv[i] + 2, ","); tok; tok = strtok(nullptr, ",")) variants.push_back(atoi(tok)); }
  }
  Timer tm;
  uint16_t* X = dalloc<uint16_t>((long)M * C); uint16_t* Y = dalloc<uint16_t>((long)M * C); uint16_t* Y2 = dalloc<uint16_t>((long)M * C);
  uint16_t* Hb = dalloc<uint16_t>((long)M * I);
  uint16_t* W1 = dalloc<uint16_t>((long)2 * I * C); uint16_t* W1f = dalloc<uint16_t>((long)2 * I * C); uint16_t* b1 = dalloc<uint16_t>(2 * I);
  uint16_t* b1f = dalloc<uint16_t>(2 * I); float* cs = dalloc<float>(2 * I);
  uint16_t* gamma = dalloc<uint16_t>(C); uint16_t* beta = dalloc<uint16_t>(C); uint16_t* W2 = dalloc<uint16_t>((long)C * I); uint16_t* b2 = dalloc<uint16_t>(C);
  fill(X, (long)M * C, 1, 1.2f, -0.3f, dt); fill(W1, (long)2 * I * C, 2, 1.0f / sqrtf((float)C), 0.0f, dt); fill(b1, 2 * I, 3, 0.1f, 0.0f, dt);
  fill(gamma, C, 4, 0.1f, 1.0f, dt); fill(beta, C, 5, 0.1f, 0.0f, dt); fill(W2, (long)C * I, 6, 1.0f / sqrtf((float)I), 0.0f, dt); fill(b2, C, 7, 0.1f, 0.0f, dt);
  if (ln) hipLaunchKernelGGL(fold_ln, dim3((2 * I + 63) / 64), dim3(64), 0, 0, W1, b1, gamma, beta, W1f, b1f, cs, 2 * I, C, dt);
  else { CK(hipMemcpy(W1f, W1, (size_t)2 * I * C * 2, hipMemcpyDeviceToDevice)); CK(hipMemcpy(b1f, b1, 2 * I * 2, hipMemcpyDeviceToDevice)); }
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> hw1((size_t)2 * I * C), hb1(2 * I), hw2((size_t)C * I);
  CK(hipMemcpy(hw1.data(), W1f, hw1.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb1.data(), b1f, hb1.size() * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hw2.data(), W2, hw2.size() * 2, hipMemcpyDeviceToHost));
  std::vector<uint8_t> img = ff_pack(hw1, hb1, hw2, dt);
  if ((int64_t)img.size() != hallo_ff320_pack_bytes()) { fprintf(stderr, "pack size mismatch\n"); return 4; }
  uint8_t* dimg = dalloc<uint8_t>(img.size()); CK(hipMemcpy(dimg, img.data(), img.size(), hipMemcpyHostToDevice));
  static void* ws = nullptr; const long ws_bytes = 256L << 20; if (!ws) { CK(hipMalloc(&ws, ws_bytes)); CK(hipMemset(ws, 0, ws_bytes)); }
  // the two-GEMM path
  hallo_gemm_desc d1; memset(&d1, 0, sizeof d1);
  d1.A = X; d1.B = W1f; d1.C = Hb; d1.M = M; d1.N = I; d1.K = C; d1.lda = C; d1.ldb = C; d1.ldc = I; d1.batch = 1; d1.bias = b1f; d1.alpha = 1.0f;
  d1.geglu = 1; d1.dtype = dt; d1.lead_alpha = 1.0f; d1.workspace = ws; d1.workspace_bytes = ws_bytes; d1.workspace_zeroed = 1;
  float* stats = dalloc<float>((long)M * 2);
  const bool fused_stats = ln && hallo_gemm_fuses_row_stats(M, I, C, 1, 0, 0);
  if (ln) { d1.ln_colsum = cs; d1.ln_eps = 1e-5f; if (!fused_stats) d1.ln_stats = stats; }
  hallo_gemm_desc d2; memset(&d2, 0, sizeof d2);
  d2.A = Hb; d2.B = W2; d2.C = Y2; d2.M = M; d2.N = C; d2.K = I; d2.lda = I; d2.ldb = I; d2.ldc = C; d2.batch = 1; d2.bias = b2; d2.alpha = 1.0f;
  d2.residual = X; d2.ldr = C; d2.dtype = dt; d2.lead_alpha = 1.0f; d2.workspace = ws; d2.workspace_bytes = ws_bytes; d2.workspace_zeroed = 1;
  auto pair = [&] { if (ln && !fused_stats) HK(hallo_row_stats(X, stats, M, C, 1e-5f, dt, nullptr)); HK(hallo_gemm(&d1, nullptr)); HK(hallo_gemm(&d2, nullptr)); };
  auto fusedk = [&] { HK(hallo_ff320(X, C, X, C, Y, C, dimg, b2, M, ln ? 1 : 0, 1e-5f, dt, nullptr)); };
  pair(); CK(hipDeviceSynchronize());
  // fp32 reference on the first and last rows
  const int rows = std::min(M, 256);
  float* Href = dalloc<float>((long)rows * I); float* Yref = dalloc<float>((long)rows * C);
  std::vector<float> href((size_t)rows * C); std::vector<uint16_t> out((size_t)M * C), out2((size_t)M * C), outp((size_t)M * C);
  CK(hipMemcpy(outp.data(), Y2, outp.size() * 2, hipMemcpyDeviceToHost));
  for (int variant : variants) {
    if (hallo_set_option("ff_fused", variant) != 0) { printf("ff_fused=%d needs a -DHALLO_ABLATIONS build\n", variant); continue; }
    CK(hipMemset(Y, 0xFF, (size_t)M * C * 2));
    long long* dbg = nullptr;
    if (variant == 9 && hallo_ff320_debug_buffer) { dbg = dalloc<long long>(64); CK(hipMemset(dbg, 0, 64 * 8)); hallo_ff320_debug_buffer(dbg); }
    fusedk(); CK(hipDeviceSynchronize());
    if (dbg) {
      fusedk(); CK(hipDeviceSynchronize());
      std::vector<long long> hs(64); CK(hipMemcpy(hs.data(), dbg, 64 * 8, hipMemcpyDeviceToHost));
      printf("ff stamps (s_memtime deltas, cycles): ");
      for (int i = 1; i < 64 && hs[i]; ++i) printf("%lld ", hs[i] - hs[i - 1]);
      printf(" | total %lld\n", hs[0] ? [&]{ long long last = 0; for (int i = 0; i < 64; ++i) if (hs[i]) last = hs[i]; return last - hs[0]; }() : 0LL);
    }
    CK(hipMemcpy(out.data(), Y, out.size() * 2, hipMemcpyDeviceToHost));
    double en = 0, rn = 0, ep = 0; long nan = 0; int nondet = 0;
    for (int part = 0; part < 2; ++part) {
      const int row0 = part ? M - rows : 0;
      hipLaunchKernelGGL(gemm_ref, dim3(((long)rows * I + 255) / 256), dim3(256), 0, 0, X, W1, b1, gamma, beta, (const uint16_t*)nullptr, Href, M, I, C, 1, (int)ln, dt, rows, row0);
      hipLaunchKernelGGL(ff2_ref, dim3(((long)rows * C + 255) / 256), dim3(256), 0, 0, Href, W2, b2, X, Yref, rows, row0, C, I, dt);
      CK(hipMemcpy(href.data(), Yref, href.size() * 4, hipMemcpyDeviceToHost));
      for (long i = 0; i < (long)rows * C; ++i) {
        const float x = host_to_f(out[(size_t)row0 * C + i], dt), xp = host_to_f(outp[(size_t)row0 * C + i], dt);
        double e = x - href[i]; en += e * e; rn += (double)href[i] * href[i]; e = xp - href[i]; ep += e * e;
      }
    }
    double dn = 0, pn = 0;
    for (size_t i = 0; i < out.size(); ++i) { const float x = host_to_f(out[i], dt), xp = host_to_f(outp[i], dt); if (x != x) ++nan; double e = x - xp; dn += e * e; pn += (double)xp * xp; }
    for (int r = 0; r < 3; ++r) { fusedk(); CK(hipDeviceSynchronize()); CK(hipMemcpy(out2.data(), Y, out2.size() * 2, hipMemcpyDeviceToHost)); if (memcmp(out.data(), out2.data(), out.size() * 2)) ++nondet; }
    tm.run(fusedk, 3);
    std::vector<float> t; for (int r = 0; r < g_rounds; ++r) t.push_back(tm.run(fusedk, g_iters));
    const float us = median(t);
    const double flop = 2.0 * M * (double)(2 * I) * C + 2.0 * M * (double)I * C;
    printf("ff M=%d dt=%s ln=%d ff_fused=%d: %.1f us  %.1f TFLOP/s  rel_l2(vs fp32, first+last %d rows)=%.2e  [two-GEMM path: %.2e]  rel_l2(vs two-GEMM, all rows)=%.2e nan=%ld nondet=%d\n",
           M, dt ? "bf16" : "f16", (int)ln, variant, us, flop / us / 1e6, rows, sqrt(en / rn), sqrt(ep / rn), sqrt(dn / pn), nan, nondet);
    fflush(stdout);
  }
  HK(hallo_set_option("ff_fused", 1));
  tm.run(pair, 3);
  std::vector<float> t; for (int r = 0; r < g_rounds; ++r) t.push_back(tm.run(pair, g_iters));
  printf("ff M=%d dt=%s ln=%d two-GEMM path (kernel %d fused_stats=%d): %.1f us\n", M, dt ? "bf16" : "f16", (int)ln, hallo_get_option("last_gemm_kernel"), (int)fused_stats, median(t));
  return 0;
}

static int cmd_gemm_suite(int argc, char** argv) {
  Timer tm;
  struct S { int M, N, K; bool geglu, ln, res; };
  // the 512x512x16f step's dominant GEMM shapes (gpurun_out/shape_breakdown.json of the r2 bench)
  const S suite[] = {
      {65536, 1280, 320, true, true, false}, {65536, 960, 320, false, true, false}, {73728, 1280, 320, true, true, false},
      {73728, 960, 320, false, true, false}, {16384, 2560, 640, true, true, false}, {4096, 5120, 1280, true, true, false},
      {65536, 320, 320, false, false, true}, {65536, 320, 1280, false, false, true}, {4608, 3840, 1280, false, true, false},
      {18432, 1920, 640, false, true, false}, {4096, 1280, 5120, false, false, true}, {18432, 640, 640, false, false, true},
      {4608, 1280, 1280, false, false, true}, {16384, 640, 2560, false, false, true}, {65536, 320, 968, false, false, true},
      {4096, 1920, 640, false, true, false}};
  for (const S& s : suite) {
    GemmOpts g; g.M = s.M; g.N = s.N; g.K = s.K; g.geglu = s.geglu; g.ln = s.ln; g.res = s.res;
    if (argc > 0 && !strcmp(argv[0], "nocheck")) g.check = false;
    run_gemm_case(g, tm);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: cbench attn-det | attn-time [v,v,..] | gemm M N K [flags] | gemm-suite\n"); return 64; }
  CK(hipSetDevice(0));
  if (getenv("CBENCH_QUICK")) { g_rounds = 1; g_iters = 3; }
  printf("cbench: libhallo_amd ABI %d\n", hallo_abi_version());
  std::string cmd = argv[1];
  if (cmd == "attn-det") return cmd_attn_det(argc - 2, argv + 2);
  if (cmd == "attn-nan") return cmd_attn_nan(argc - 2, argv + 2);
  if (cmd == "attn-time") return cmd_attn_time(argc - 2, argv + 2);
  if (cmd == "gemm" && argc >= 5) { Timer tm; run_gemm_case(parse_gemm(argc - 2, argv + 2), tm); return 0; }
  if (cmd == "gemm-suite") return cmd_gemm_suite(argc - 2, argv + 2);
  if (cmd == "ff") return cmd_ff(argc - 2, argv + 2);
  fprintf(stderr, "unknown command\n");
  return 64;
}
