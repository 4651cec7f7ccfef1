// SVGD: posterior update + shared-state base-optimizer apply (+ next step's Gram) in ONE pass.
//
// Reference: src/algos/svgd.py:86-103 -- phi, then for every particle in order:
// model_param.grad = -phi[i], model_param.data = particle_i, base_optimizer.step(), with ONE
// optimizer whose state is keyed on the model's parameters and therefore shared by all
// particles and advanced M times per SVGD step.
//
// Unfused, that is combine (read P, G; write -phi: 12*M*D B) + apply (read P, -phi; write P:
// (12*M + 8)*D B) + next step's Gram (read P: 4*M*D B) = 28*M*D B per full step.  Here one
// thread owns a float4 column of all M particles: it forms the M rows of -phi in registers,
// walks the particles in order carrying the shared optimizer state in registers, writes the
// updated particles back in place and (GRAM) accumulates the mean-centred Gram matrix of the
// UPDATED particles, which is exactly what the next step's kernel statistics need (nothing
// touches the particles between two steps but this kernel):
//   bytes = 8*M*D (read P, G) + 4*M*D (write P) + optimizer state  ->  (12*M + 8)*D for SGD,
// -phi is never materialised.  Still HBM-bound: 2*M^2 (+ M(M+1)/2 for the Gram) FMAs per
// coordinate on the VALU, which idles otherwise.
#include "svgd_shared.hpp"

namespace bde {

// The in-place particle stream is loaded and stored non-temporally like the gradient stream: 0.482 ms vs 0.500 ms
// per full step with plain accesses (tools/kexp6.hip A/B, profiles/r02_small_step_timeline_v1_and_fused_ab.txt).
#ifndef BDE_FUSED_PLD
#define BDE_FUSED_PLD ld4_nt
#endif
#ifndef BDE_FUSED_PST
#define BDE_FUSED_PST st4_nt
#endif

constexpr int kFusedMaxBlocks = kGramMaxBlocks;   // the Gram partials go into the same ws slots

// SEG: the gradients come from the tensors autograd produced (svgd_shared.hpp: `G` is then unused, `seg_ptrs` /
// `chunks` describe them, and the loop runs over chunks of <= 256 float4 columns of one parameter tensor instead of
// over all float4 columns of the row; D % 4 == 0 there, so there is no scalar tail).
template <int M, int OPT /* 0 sgd, 1 adam */, bool GRAM, bool SEG>
__global__ __launch_bounds__(kBlock) void svgd_fused_kernel(float* __restrict__ P, const float* __restrict__ G,
                                                           float* __restrict__ s0, float* __restrict__ s1, int64_t D,
                                                           int64_t ld, int64_t ldg, const float* __restrict__ cgT,
                                                           const float* __restrict__ cpT, SgdParams sk, AdamParams ak,
                                                           AdamSteps st, float* __restrict__ ws_next,
                                                           const float* const* __restrict__ seg_ptrs,
                                                           const SegChunk* __restrict__ chunks, int n_chunks) {
  constexpr int NP = M * (M + 1) / 2;
  const int64_t n4 = D >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  float gacc[GRAM ? NP : 1];
#pragma unroll
  for (int k = 0; k < (GRAM ? NP : 1); ++k) gacc[k] = 0.f;

  // flat G: thread-strided walk over the float4 columns; SEG: workgroup-strided walk over the chunks
  const int64_t n_iter = SEG ? static_cast<int64_t>(n_chunks) : n4;
  const int64_t it0 = SEG ? static_cast<int64_t>(blockIdx.x) : static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t it_step = SEG ? static_cast<int64_t>(gridDim.x) : stride;
  // SEG: the descriptor of the NEXT chunk is requested before this chunk's work, so that only the (wave-uniform) load
  // of the M gradient pointers is on an iteration's critical path, and that one overlaps the particle loads
  SegChunk ch = {}, ch_next = {};
  if (SEG && it0 < n_iter) ch = chunks[it0];
  for (int64_t it = it0; it < n_iter; it += it_step, ch = ch_next) {
    int64_t i4 = it, l4 = 0;
    int valid = 4;
    const float* const* gp = nullptr;
    if (SEG) {
      if (it + it_step < n_iter) ch_next = chunks[it + it_step];
      valid = ch.nflt - 4 * static_cast<int>(threadIdx.x);
      if (valid <= 0) continue;
      gp = seg_ptrs + static_cast<int64_t>(ch.seg) * M;
      i4 = ch.c4 + threadIdx.x;
      l4 = ch.loc4 + threadIdx.x;
    }
    f32x4 p[M], u[M];
#pragma unroll
    for (int i = 0; i < M; ++i) u[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < M; ++j) {
      p[j] = BDE_FUSED_PLD(P + j * ld + 4 * i4);
      const f32x4 g = SEG ? seg_load(gp[j] + 4 * l4, valid) : ld4_nt(G + j * ldg + 4 * i4);
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const float a = cgT[j * M + i], b = cpT[j * M + i];
#pragma unroll
        for (int c = 0; c < 4; ++c) u[i][c] = __builtin_fmaf(b, p[j][c], __builtin_fmaf(a, g[c], u[i][c]));
      }
    }
    // u[i] = -phi_i: the gradient the reference hands to the base optimizer (svgd.py:95)
    f32x4 a0, a1;
    if (OPT == 0) {
      a0 = (sk.momentum != 0.f && !sk.first) ? ld4(s0 + 4 * i4) : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      a0 = ld4(s0 + 4 * i4);
      a1 = ld4(s1 + 4 * i4);
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float b = a0[c];
        if (OPT == 0) {
          p[i][c] = sgd_apply(p[i][c], u[i][c], b, sk, i == 0);
        } else {
          float v = a1[c];
          p[i][c] = adam_apply(p[i][c], u[i][c], b, v, ak, st.step_size[i], st.bc2_sqrt[i]);
          a1[c] = v;
        }
        a0[c] = b;
      }
      BDE_FUSED_PST(P + i * ld + 4 * i4, p[i]);
    }
    if (OPT == 0) {
      if (sk.momentum != 0.f) st4(s0 + 4 * i4, a0);
    } else {
      st4(s0 + 4 * i4, a0);
      st4(s1 + 4 * i4, a1);
    }
    if (GRAM) {
      f32x4 mean = p[0];
#pragma unroll
      for (int i = 1; i < M; ++i) mean += p[i];
      mean *= (1.0f / M);
#pragma unroll
      for (int i = 0; i < M; ++i) p[i] -= mean;
      int k = 0;
#pragma unroll
      for (int a = 0; a < M; ++a) {
#pragma unroll
        for (int b = a; b < M; ++b) {
#pragma unroll
          for (int c = 0; c < 4; ++c) gacc[k] = __builtin_fmaf(p[a][c], p[b][c], gacc[k]);
          ++k;
        }
      }
    }
  }

  // the D % 4 tail coordinates (block 0, one thread per coordinate)
  if (!SEG && blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < D) {
      float p[M], u[M];
#pragma unroll
      for (int i = 0; i < M; ++i) u[i] = 0.f;
#pragma unroll
      for (int j = 0; j < M; ++j) {
        p[j] = P[j * ld + e];
        const float g = G[j * ldg + e];
#pragma unroll
        for (int i = 0; i < M; ++i) u[i] = __builtin_fmaf(cpT[j * M + i], p[j], __builtin_fmaf(cgT[j * M + i], g, u[i]));
      }
      float b = 0.f, v = 0.f;
      if (OPT == 0) {
        if (sk.momentum != 0.f && !sk.first) b = s0[e];
      } else {
        b = s0[e];
        v = s1[e];
      }
#pragma unroll
      for (int i = 0; i < M; ++i) {
        p[i] = (OPT == 0) ? sgd_apply(p[i], u[i], b, sk, i == 0)
                          : adam_apply(p[i], u[i], b, v, ak, st.step_size[i], st.bc2_sqrt[i]);
        P[i * ld + e] = p[i];
      }
      if (OPT == 0) {
        if (sk.momentum != 0.f) s0[e] = b;
      } else {
        s0[e] = b;
        s1[e] = v;
      }
      if (GRAM) {
        float mean = 0.f;
#pragma unroll
        for (int i = 0; i < M; ++i) mean += p[i];
        mean *= (1.0f / M);
        int k = 0;
#pragma unroll
        for (int a = 0; a < M; ++a)
#pragma unroll
          for (int c = a; c < M; ++c) {
            gacc[k] = __builtin_fmaf(p[a] - mean, p[c] - mean, gacc[k]);
            ++k;
          }
      }
    }
  }

  if (GRAM) {
    // workgroup partial of the (padded 8x8) Gram tile, same ws layout as svgd_gram_kernel<2>
    __shared__ float red[kBlock / 64][NP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const float v = wave_sum(gacc[k]);
      if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int pi = threadIdx.x >> 3, pj = threadIdx.x & 7;
      float s = 0.f;
      if (pi < M && pj < M) {
        const int a = pi < pj ? pi : pj, b = pi < pj ? pj : pi;
        const int k = a * M - a * (a - 1) / 2 + (b - a);        // index of pair (a <= b) in row-major upper triangle
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) s += red[w][k];
      }
      ws_next[kWsHeaderFloats + static_cast<int64_t>(blockIdx.x) * 64 + threadIdx.x] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      ws_next[0] = static_cast<float>(gridDim.x);
      ws_next[1] = 8.0f;
    }
  }
}

struct SegArgs {                     // segmented gradients (null seg_ptrs: flat G)
  const float* const* seg_ptrs = nullptr;
  const SegChunk* chunks = nullptr;
  int n_chunks = 0;
};

template <int M, int OPT>
static int launch_fused(float* P, const float* G, float* s0, float* s1, int64_t D, int64_t ld, int64_t ldg,
                        const float* kstat, const SgdParams& sk, const AdamParams& ak, const AdamSteps& st,
                        float* ws_next, const SegArgs& sg, hipStream_t s) {
  const int n = M * M;
  const float* cg = kstat + 2 * n + M + 4;
  const float* cp = cg + n;
  const bool seg = sg.seg_ptrs != nullptr;
  const int grid = seg ? static_cast<int>(std::min<int64_t>(sg.n_chunks, kFusedMaxBlocks))
                       : stream_grid((D + 3) / 4, kBlock, kFusedMaxBlocks);
#define BDE_FUSED_LAUNCH(GRAM, SEG, WS)                                                                                \
  hipLaunchKernelGGL((svgd_fused_kernel<M, OPT, GRAM, SEG>), dim3(grid), dim3(kBlock), 0, s, P, G, s0, s1, D, ld, ldg, \
                     cg, cp, sk, ak, st, WS, sg.seg_ptrs, sg.chunks, sg.n_chunks)
  if constexpr (M <= 8) {
    if (ws_next) {
      if (seg) BDE_FUSED_LAUNCH(true, true, ws_next);
      else BDE_FUSED_LAUNCH(true, false, ws_next);
      return to_err(hipGetLastError());
    }
  }
  if (seg) BDE_FUSED_LAUNCH(false, true, static_cast<float*>(nullptr));
  else BDE_FUSED_LAUNCH(false, false, static_cast<float*>(nullptr));
#undef BDE_FUSED_LAUNCH
  return to_err(hipGetLastError());
}

template <int OPT>
static int dispatch_fused(int M, float* P, const float* G, float* s0, float* s1, int64_t D, int64_t ld, int64_t ldg,
                          const float* kstat, const SgdParams& sk, const AdamParams& ak, const AdamSteps& st,
                          float* ws_next, const SegArgs& sg, hipStream_t s) {
  switch (M) {
#define BDE_CASE(m) \
  case m:           \
    return launch_fused<m, OPT>(P, G, s0, s1, D, ld, ldg, kstat, sk, ak, st, ws_next, sg, s);
    BDE_CASE(1) BDE_CASE(2) BDE_CASE(3) BDE_CASE(4) BDE_CASE(5) BDE_CASE(6) BDE_CASE(7) BDE_CASE(8)
    BDE_CASE(9) BDE_CASE(10) BDE_CASE(11) BDE_CASE(12) BDE_CASE(13) BDE_CASE(14) BDE_CASE(15) BDE_CASE(16)
#undef BDE_CASE
  }
  return BDE_ERR_INVALID;
}

}  // namespace bde

using namespace bde;

extern "C" int bde_svgd_fused_gram_supported(int M) { return (M >= 1 && M <= 8) ? 1 : 0; }

extern "C" int bde_svgd_fused_sgd(float* P, const float* G, float* momentum_buf, int M, int64_t D, int64_t ld,
                                  int64_t ldg, const float* kstat, double lr, double momentum, double dampening,
                                  double weight_decay, int nesterov, int first, void* ws_next, void* stream) {
  if (!svgd_args_ok(P, M, D, ld) || M > BDE_FAST_PARTICLES || !G || !kstat || !aligned16(G) ||
      (momentum != 0.0 && !momentum_buf))
    return BDE_ERR_INVALID;
  if (ldg == 0) ldg = ld;
  if (ldg < D || (ldg & 3)) return BDE_ERR_INVALID;
  if (ws_next && (M > 8 || !aligned16(ws_next))) return BDE_ERR_INVALID;
  const SgdParams sk{static_cast<float>(lr), static_cast<float>(momentum), static_cast<float>(1.0 - dampening),
                     static_cast<float>(weight_decay), nesterov, first};
  return dispatch_fused<0>(M, P, G, momentum_buf, nullptr, D, ld, ldg, kstat, sk, AdamParams{}, AdamSteps{},
                           static_cast<float*>(ws_next), SegArgs{}, static_cast<hipStream_t>(stream));
}

extern "C" int bde_svgd_fused_sgd_seg(float* P, const void* const* seg_ptrs, const bde_seg_chunk* chunks,
                                      int64_t n_chunks, float* momentum_buf, int M, int64_t D, int64_t ld,
                                      const float* kstat, double lr, double momentum, double dampening,
                                      double weight_decay, int nesterov, int first, void* ws_next, void* stream) {
  if (!svgd_args_ok(P, M, D, ld) || M > BDE_FAST_PARTICLES || !seg_args_ok(seg_ptrs, chunks, n_chunks, D) || !kstat ||
      (momentum != 0.0 && !momentum_buf))
    return BDE_ERR_INVALID;
  if (ws_next && (M > 8 || !aligned16(ws_next))) return BDE_ERR_INVALID;
  const SgdParams sk{static_cast<float>(lr), static_cast<float>(momentum), static_cast<float>(1.0 - dampening),
                     static_cast<float>(weight_decay), nesterov, first};
  const SegArgs sg{reinterpret_cast<const float* const*>(seg_ptrs), chunks, static_cast<int>(n_chunks)};
  return dispatch_fused<0>(M, P, nullptr, momentum_buf, nullptr, D, ld, ld, kstat, sk, AdamParams{}, AdamSteps{},
                           static_cast<float*>(ws_next), sg, static_cast<hipStream_t>(stream));
}

extern "C" int bde_svgd_fused_adam(float* P, const float* G, float* exp_avg, float* exp_avg_sq, int M, int64_t D,
                                   int64_t ld, int64_t ldg, const float* kstat, double lr, double beta1, double beta2, double eps,
                                   double weight_decay, int64_t step0, void* ws_next, void* stream) {
  if (!svgd_args_ok(P, M, D, ld) || M > BDE_FAST_PARTICLES || !G || !kstat || !aligned16(G) || !exp_avg ||
      !exp_avg_sq || step0 < 0)
    return BDE_ERR_INVALID;
  if (ldg == 0) ldg = ld;
  if (ldg < D || (ldg & 3)) return BDE_ERR_INVALID;
  if (ws_next && (M > 8 || !aligned16(ws_next))) return BDE_ERR_INVALID;
  const AdamParams ak{static_cast<float>(beta1), static_cast<float>(beta2), static_cast<float>(1.0 - beta1),
                      static_cast<float>(1.0 - beta2), static_cast<float>(eps), static_cast<float>(weight_decay)};
  return dispatch_fused<1>(M, P, G, exp_avg, exp_avg_sq, D, ld, ldg, kstat, SgdParams{}, ak,
                           make_adam_steps(lr, beta1, beta2, step0), static_cast<float*>(ws_next), SegArgs{},
                           static_cast<hipStream_t>(stream));
}

extern "C" int bde_svgd_fused_adam_seg(float* P, const void* const* seg_ptrs, const bde_seg_chunk* chunks,
                                       int64_t n_chunks, float* exp_avg, float* exp_avg_sq, int M, int64_t D, int64_t ld,
                                       const float* kstat, double lr, double beta1, double beta2, double eps,
                                       double weight_decay, int64_t step0, void* ws_next, void* stream) {
  if (!svgd_args_ok(P, M, D, ld) || M > BDE_FAST_PARTICLES || !seg_args_ok(seg_ptrs, chunks, n_chunks, D) || !kstat ||
      !exp_avg || !exp_avg_sq || step0 < 0)
    return BDE_ERR_INVALID;
  if (ws_next && (M > 8 || !aligned16(ws_next))) return BDE_ERR_INVALID;
  const AdamParams ak{static_cast<float>(beta1), static_cast<float>(beta2), static_cast<float>(1.0 - beta1),
                      static_cast<float>(1.0 - beta2), static_cast<float>(eps), static_cast<float>(weight_decay)};
  const SegArgs sg{reinterpret_cast<const float* const*>(seg_ptrs), chunks, static_cast<int>(n_chunks)};
  return dispatch_fused<1>(M, P, nullptr, exp_avg, exp_avg_sq, D, ld, ld, kstat, SgdParams{}, ak,
                           make_adam_steps(lr, beta1, beta2, step0), static_cast<float*>(ws_next), sg,
                           static_cast<hipStream_t>(stream));
}

// bde_init(): load this translation unit's code object on the current device now (HIP otherwise uploads it at the
// first launch of one of its kernels).  Internal to the library (not exported).
extern "C" __attribute__((visibility("hidden"))) int bde_internal_load_svgd_fused(void) {
  hipFuncAttributes attr;
  return bde::to_err(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&bde::svgd_fused_kernel<8, 0, false, false>)));
}
