// Definitions shared by svgd.hip and svgd_fused.hip.
#pragma once
#include "bde_common.hpp"

namespace bde {

// ws header (1.5 KB, so that the partial tiles behind it start on a 128-byte line): ws[0] = #partial tiles,
// ws[1] = padded M (8 or 16); the rest is unused (rounds 2-3 kept the hand-off words of a single-launch path there).
constexpr int kWsHeaderFloats = 32 * 12;
constexpr int kGramMaxBlocks = 1024;       // 4 workgroups per CU: best measured (tools/kexp.hip)

// Per-particle Adam scalars of the SHARED step counter (advanced once per particle).
struct AdamSteps {
  float step_size[BDE_MAX_PARTICLES];     // lr / (1 - beta1^t)
  float bc2_sqrt[BDE_MAX_PARTICLES];      // 1 / sqrt(1 - beta2^t)
};

static inline AdamSteps make_adam_steps(double lr, double beta1, double beta2, int64_t step0) {
  AdamSteps st;
  for (int i = 0; i < BDE_MAX_PARTICLES; ++i) {
    const double t = static_cast<double>(step0 + i + 1);
    st.step_size[i] = static_cast<float>(lr / (1.0 - std::pow(beta1, t)));
    st.bc2_sqrt[i] = static_cast<float>(1.0 / std::sqrt(1.0 - std::pow(beta2, t)));
  }
  return st;
}

// ---- gradients handed over as the tensors autograd produced (no copy into the flat gradient rows) ----
// The flat row of a particle is the concatenation of its parameter tensors, each starting on a float4 boundary
// (FlatLayout(align=4)).  "Segment" s = tensor s: columns [col0_s, col0_s + numel_s).  The gradient of particle j for
// segment s lives wherever autograd put it: seg_ptrs[s * M + j] (16-byte aligned; the caller substitutes the address
// of the segment inside a flat gradient row when a gradient is missing, unaligned or not contiguous, after copying /
// zeroing there).  The kernels walk a STATIC list of chunks -- at most 256 float4 columns of ONE segment, i.e. one
// float4 column per thread of a workgroup, the same 4 KB-per-row pieces the flat kernels touch per iteration -- so
// everything about a chunk (segment, position) is wave-uniform and comes through scalar loads.
using SegChunk = bde_seg_chunk;
static_assert(sizeof(SegChunk) == 32, "bde_seg_chunk is 32 bytes");

// The gradient float4 of particle row `gj` (already offset to this thread's column) with `valid` floats in bounds.
__device__ __forceinline__ f32x4 seg_load(const float* gj, int valid) {
  if (valid >= 4) return ld4_nt(gj);
  f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c)
    if (c < valid) g[c] = gj[c];
  return g;
}

static inline bool seg_args_ok(const void* const* seg_ptrs, const bde_seg_chunk* chunks, int64_t n_chunks, int64_t D) {
  return seg_ptrs && chunks && n_chunks >= 1 && n_chunks <= (int64_t{1} << 30) && (D & 3) == 0;
}

// ---- generic path (16 < M <= 64): particles in groups of 8, one 16-row Gram tile per pair of groups ----
static inline int svgd_groups(int M) { return (M + 7) / 8; }
static inline int svgd_pairs(int M) { const int g = svgd_groups(M); return g * (g + 1) / 2; }
// ws layout for M > 16: [header][pairs x kGramMaxBlocks x 256 partial tiles][M*M d2 matrix]
static inline size_t svgd_generic_d2_offset(int M) {
  return static_cast<size_t>(kWsHeaderFloats) + static_cast<size_t>(svgd_pairs(M)) * kGramMaxBlocks * 256;
}

static inline bool svgd_args_ok(const float* P, int M, int64_t D, int64_t ld) {
  return P && M >= 1 && M <= BDE_MAX_PARTICLES && D >= 1 && ld >= D && (ld & 3) == 0 && aligned16(P);
}

// One optimizer application to one element, torch.optim semantics, state carried in registers.
struct SgdParams {
  float lr, momentum, omd, wd;
  int nesterov, first;
};
__device__ __forceinline__ float sgd_apply(float p, float g, float& b, const SgdParams& k, bool first_particle) {
  if (k.wd != 0.f) g = __builtin_fmaf(k.wd, p, g);
  if (k.momentum != 0.f) {
    if (k.first && first_particle) b = g;              // torch initialises the buffer with the first gradient
    else b = k.momentum * b + k.omd * g;
    g = k.nesterov ? __builtin_fmaf(k.momentum, b, g) : b;
  }
  return p - k.lr * g;
}
struct AdamParams {
  float beta1, beta2, omb1, omb2, eps, wd;
};
__device__ __forceinline__ float adam_apply(float p, float g, float& m, float& v, const AdamParams& k, float step_size,
                                            float ibc2_sqrt) {
  if (k.wd != 0.f) g = __builtin_fmaf(k.wd, p, g);
  m = m + (g - m) * k.omb1;                             // exp_avg.lerp_(grad, 1 - beta1)
  v = __builtin_fmaf(k.omb2 * g, g, k.beta2 * v);       // mul_(beta2).addcmul_(g, g, value=1-beta2)
  // hardware sqrt / reciprocal (1 ulp each) instead of the IEEE expansions: the M applications per element
  // form one serial dependency chain (shared m, v), so their latency, not their count, sets the pace
  const float denom = __builtin_amdgcn_sqrtf(v) * ibc2_sqrt + k.eps;
  return p - step_size * (m * __builtin_amdgcn_rcpf(denom));
}

}  // namespace bde
