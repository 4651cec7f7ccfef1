// iVON: weight-noise draw from the precision and the fused natural-gradient
// mean / precision (Hessian) update.
//
// Reference: src/algos/ivorn.py:102-115 (sample_parameters) and :66-89 (update
// block of step): ~6 ATen launches per tensor per MC sample and ~14 per tensor
// per update inside a Python double loop.  Here each is one streaming pass
// over flat [D] buffers: sample 20 B/param (mean, prec read; param write;
// delta_sum RMW -- 16 B on the first MC sample), update 32 B/param (5 reads,
// 3 writes).  The arithmetic mirrors the reference's fp32 op order with
// separately rounded operations (-ffp-contract=off), so on identical inputs
// the results agree to the last bit wherever fp32 division/sqrt are IEEE.
#include "bde_common.hpp"

namespace bde {

template <bool RNG>
__global__ __launch_bounds__(kBlock) void ivon_sample_kernel(const float* __restrict__ mean,
                                                            const float* __restrict__ prec,
                                                            const float* __restrict__ eps, uint64_t seed,
                                                            uint64_t stream_id, float n_eff, int deterministic,
                                                            int first, float* __restrict__ param,
                                                            float* __restrict__ delta_sum, int64_t n) {
  const int64_t n4 = n >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 m = ld4(mean + 4 * i), pr = ld4(prec + 4 * i);
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    if (!deterministic) {
      const f32x4 e = RNG ? philox_normal4(seed, stream_id, static_cast<uint64_t>(i), kDomainDiag) : ld4_nt(eps + 4 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j)   // ivorn.py:108: 1 / (N * prec.clamp(min=1e-4)).sqrt() * eps
        d[j] = (1.0f / __builtin_sqrtf(n_eff * fmaxf(pr[j], 1e-4f))) * e[j];
    }
    st4_nt(param + 4 * i, m + d);                               // ivorn.py:111
    st4(delta_sum + 4 * i, first ? d : ld4(delta_sum + 4 * i) + d);   // ivorn.py:112-115
  }
  if (blockIdx.x == 0) {
    const int64_t k = (n4 << 2) + threadIdx.x;
    if (k < n) {
      float d = 0.f;
      if (!deterministic) {
        float e;
        if (RNG) {
          const f32x4 z = philox_normal4(seed, stream_id, static_cast<uint64_t>(n4), kDomainDiag);
          e = z[threadIdx.x & 3];
        } else {
          e = eps[k];
        }
        d = (1.0f / __builtin_sqrtf(n_eff * fmaxf(prec[k], 1e-4f))) * e;
      }
      param[k] = mean[k] + d;
      delta_sum[k] = first ? d : delta_sum[k] + d;
    }
  }
}

struct IvonScalars {
  float lam, n_eff, mc, beta1, omb1, omb2, c2, bc1, bc2, lr, damping;
};

__device__ __forceinline__ void ivon_elem(float& mean, float& mom, float& prec, float dsum, float acc, const IvonScalars& k) {
  const float gradient = acc / k.mc;                                           // ivorn.py:79
  const float g_mu = k.lam * mean + gradient;                                  // :80
  mom = k.beta1 * mom + k.omb1 * g_mu;                                         // :81
  const float g_s = ((k.lam - prec) + (((k.n_eff * prec) * dsum) / k.mc) * gradient) + k.damping;   // :82
  const float cm = mom / k.bc1;                                                // :84
  const float cp = prec / k.bc2;                                               // :85
  mean = mean - (k.lr * cm) / cp;                                              // :88
  prec = prec + (k.omb2 + ((k.c2 * g_s) / prec)) * g_s;                        // :89
}

__global__ __launch_bounds__(kBlock) void ivon_update_kernel(float* __restrict__ mean, float* __restrict__ momentum,
                                                            float* __restrict__ prec,
                                                            const float* __restrict__ delta_sum,
                                                            const float* __restrict__ acc_grad, IvonScalars k,
                                                            int64_t n) {
  const int64_t n4 = n >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 m = ld4(mean + 4 * i), mo = ld4(momentum + 4 * i), pr = ld4(prec + 4 * i);
    // read-once streams: non-temporal loads keep them from evicting the three in-place RMW streams
    // (+24 % measured, tools/kexp2.hip)
    const f32x4 ds = ld4_nt(delta_sum + 4 * i), ag = ld4_nt(acc_grad + 4 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = m[j], b = mo[j], c = pr[j];
      ivon_elem(a, b, c, ds[j], ag[j], k);
      m[j] = a;
      mo[j] = b;
      pr[j] = c;
    }
    st4(mean + 4 * i, m);
    st4(momentum + 4 * i, mo);
    st4(prec + 4 * i, pr);
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < n) {
      float a = mean[e], b = momentum[e], c = prec[e];
      ivon_elem(a, b, c, delta_sum[e], acc_grad[e], k);
      mean[e] = a;
      momentum[e] = b;
      prec[e] = c;
    }
  }
}

}  // namespace bde

using namespace bde;

extern "C" int bde_ivon_sample(const float* mean, const float* prec, const float* eps, uint64_t seed,
                               uint64_t stream_id, float n_eff, int deterministic, int first, float* param,
                               float* delta_sum, int64_t n, void* stream) {
  if (!mean || !prec || !param || !delta_sum || n <= 0) return BDE_ERR_INVALID;
  if (!aligned16(mean) || !aligned16(prec) || !aligned16(param) || !aligned16(delta_sum) || (eps && !aligned16(eps)))
    return BDE_ERR_INVALID;
  const int grid = stream_grid((n + 3) / 4);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (eps)
    hipLaunchKernelGGL(ivon_sample_kernel<false>, dim3(grid), dim3(kBlock), 0, s, mean, prec, eps, seed, stream_id,
                       n_eff, deterministic, first, param, delta_sum, n);
  else
    hipLaunchKernelGGL(ivon_sample_kernel<true>, dim3(grid), dim3(kBlock), 0, s, mean, prec, eps, seed, stream_id,
                       n_eff, deterministic, first, param, delta_sum, n);
  return to_err(hipGetLastError());
}

extern "C" int bde_ivon_update(float* mean, float* momentum, float* prec, const float* delta_sum,
                               const float* acc_grad, float lam, float n_eff, float mc, float beta1, float omb1,
                               float omb2, float c2, float bc1, float bc2, float lr, float damping, int64_t n,
                               void* stream) {
  if (!mean || !momentum || !prec || !delta_sum || !acc_grad || n <= 0) return BDE_ERR_INVALID;
  if (!aligned16(mean) || !aligned16(momentum) || !aligned16(prec) || !aligned16(delta_sum) || !aligned16(acc_grad))
    return BDE_ERR_INVALID;
  const IvonScalars k{lam, n_eff, mc, beta1, omb1, omb2, c2, bc1, bc2, lr, damping};
  const int grid = stream_grid((n + 3) / 4);
  hipLaunchKernelGGL(ivon_update_kernel, dim3(grid), dim3(kBlock), 0, static_cast<hipStream_t>(stream), mean, momentum,
                     prec, delta_sum, acc_grad, k, n);
  return to_err(hipGetLastError());
}

// bde_init(): load this translation unit's code object on the current device now (HIP otherwise uploads it at the
// first launch of one of its kernels).  Internal to the library (not exported).
extern "C" __attribute__((visibility("hidden"))) int bde_internal_load_ivon(void) {
  hipFuncAttributes attr;
  return bde::to_err(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&bde::ivon_update_kernel)));
}
