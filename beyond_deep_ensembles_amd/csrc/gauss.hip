// Mean-field Gaussian parameters (Bayes by Backprop): reparameterised draw,
// its backward, and the closed-form KL with fused analytic gradients.
//
// Reference: src/algos/util.py:151-186 (GaussianParameter, normal_like) and
// src/algos/bbb.py:18-21,71-80 (GaussianPrior.kl_divergence, the KL/L2
// collection loop of BBBOptimizer.step).  The reference runs ~4 ATen launches
// per tensor per draw plus an autograd graph, and ~8 launches + a reduction
// per tensor for the KL (again in backward).  Here mean/rho of all Gaussian
// parameters live in two flat buffers and each operation is one streaming
// pass: draw 12 B/param (8 with Philox noise), draw-backward 24 B/param (RMW)
// or 16 (overwrite), KL value + both gradients 16 B/param (overwrite) or 24
// (accumulate).  All HBM-bound.
#include "bde_common.hpp"

namespace bde {

constexpr int kReduceMaxBlocks = 1024;
constexpr int kReduceHeader = 2;   // doubles: [0] = #partials

// ------------------------------------------------------------------ draw --
template <bool RNG>
__global__ __launch_bounds__(kBlock) void gauss_draw_fwd_kernel(const float* __restrict__ mean,
                                                               const float* __restrict__ rho,
                                                               const float* __restrict__ eps, uint64_t seed,
                                                               uint64_t stream_id, float* __restrict__ w,
                                                               float* __restrict__ eps_out, int64_t n) {
  const int64_t n4 = n >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 m = ld4_nt(mean + 4 * i), r = ld4_nt(rho + 4 * i);
    const f32x4 e = RNG ? philox_normal4(seed, stream_id, static_cast<uint64_t>(i), kDomainDiag) : ld4_nt(eps + 4 * i);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = m[j] + e[j] * softplus(r[j]);   // util.py:171: mean + eps * std
    BDE_OUT_ST(w + 4 * i, o);
    if (RNG && eps_out) st4_nt(eps_out + 4 * i, e);
  }
  if (blockIdx.x == 0) {
    const int64_t k = (n4 << 2) + threadIdx.x;
    if (k < n) {
      float e;
      if (RNG) {
        const f32x4 z = philox_normal4(seed, stream_id, static_cast<uint64_t>(n4), kDomainDiag);
        e = z[threadIdx.x & 3];
        if (eps_out) eps_out[k] = e;
      } else {
        e = eps[k];
      }
      w[k] = mean[k] + e * softplus(rho[k]);
    }
  }
}

template <bool RNG, bool ACC>
__global__ __launch_bounds__(kBlock) void gauss_draw_bwd_kernel(const float* __restrict__ g,
                                                               const float* __restrict__ rho,
                                                               const float* __restrict__ eps, uint64_t seed,
                                                               uint64_t stream_id, float* __restrict__ gmean,
                                                               float* __restrict__ grho, int64_t n) {
  const int64_t n4 = n >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 go = ld4_nt(g + 4 * i), r = ld4_nt(rho + 4 * i);
    const f32x4 e = RNG ? philox_normal4(seed, stream_id, static_cast<uint64_t>(i), kDomainDiag) : ld4_nt(eps + 4 * i);
    f32x4 gm = go, gr;
#pragma unroll
    for (int j = 0; j < 4; ++j) gr[j] = (go[j] * e[j]) * sigmoidf(r[j]);   // d softplus = sigmoid
    if (ACC) {
      gm = gm + ld4(gmean + 4 * i);
      gr = gr + ld4(grho + 4 * i);
    }
    st4(gmean + 4 * i, gm);
    st4(grho + 4 * i, gr);
  }
  if (blockIdx.x == 0) {
    const int64_t k = (n4 << 2) + threadIdx.x;
    if (k < n) {
      float e;
      if (RNG) {
        const f32x4 z = philox_normal4(seed, stream_id, static_cast<uint64_t>(n4), kDomainDiag);
        e = z[threadIdx.x & 3];
      } else {
        e = eps[k];
      }
      const float gm = g[k], gr = (g[k] * e) * sigmoidf(rho[k]);
      gmean[k] = ACC ? gmean[k] + gm : gm;
      grho[k] = ACC ? grho[k] + gr : gr;
    }
  }
}

// Scalar variants for operands that are not 16-byte aligned (per-tensor views into a flat
// buffer, e.g. the small rank-1 vectors): one element per thread, same Philox indexing.
template <bool RNG>
__global__ __launch_bounds__(kBlock) void gauss_draw_fwd_scalar_kernel(const float* __restrict__ mean,
                                                                      const float* __restrict__ rho,
                                                                      const float* __restrict__ eps, uint64_t seed,
                                                                      uint64_t stream_id, float* __restrict__ w,
                                                                      float* __restrict__ eps_out, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n; k += stride) {
    float e;
    if (RNG) {
      e = philox_normal4(seed, stream_id, static_cast<uint64_t>(k >> 2), kDomainDiag)[k & 3];
      if (eps_out) eps_out[k] = e;
    } else {
      e = eps[k];
    }
    w[k] = mean[k] + e * softplus(rho[k]);
  }
}

template <bool RNG, bool ACC>
__global__ __launch_bounds__(kBlock) void gauss_draw_bwd_scalar_kernel(const float* __restrict__ g,
                                                                      const float* __restrict__ rho,
                                                                      const float* __restrict__ eps, uint64_t seed,
                                                                      uint64_t stream_id, float* __restrict__ gmean,
                                                                      float* __restrict__ grho, int64_t n) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n; k += stride) {
    const float e = RNG ? philox_normal4(seed, stream_id, static_cast<uint64_t>(k >> 2), kDomainDiag)[k & 3] : eps[k];
    const float gm = g[k], gr = (g[k] * e) * sigmoidf(rho[k]);
    gmean[k] = ACC ? gmean[k] + gm : gm;
    grho[k] = ACC ? grho[k] + gr : gr;
  }
}

// -------------------------------------------------- local reparameterisation --
// Epilogue of the mean-field layers (bbb_layers.py:70-80 of the reference): the two GEMMs / convs give the
// pre-activation mean and variance, then  out = mean + sqrt(var) * eps  (4 ATen launches + autograd nodes
// there: sqrt, normal_, mul, add).  One pass forward (12 B/element with in-kernel noise), one pass backward:
//   g_mean = g,   g_var = g * eps / (2 sqrt(var)).
template <bool RNG>
__global__ __launch_bounds__(kBlock) void local_reparam_fwd_kernel(const float* __restrict__ mean,
                                                                  const float* __restrict__ var,
                                                                  const float* __restrict__ eps, uint64_t seed,
                                                                  uint64_t stream_id, float* __restrict__ out,
                                                                  int64_t n) {
  const int64_t n4 = n >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 m = ld4_nt(mean + 4 * i), v = ld4_nt(var + 4 * i);
    const f32x4 e = RNG ? philox_normal4(seed, stream_id, static_cast<uint64_t>(i), kDomainDiag) : ld4_nt(eps + 4 * i);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = m[j] + __builtin_sqrtf(v[j]) * e[j];
    BDE_OUT_ST(out + 4 * i, o);
  }
  if (blockIdx.x == 0) {
    const int64_t k = (n4 << 2) + threadIdx.x;
    if (k < n) {
      const float e = RNG ? philox_normal4(seed, stream_id, static_cast<uint64_t>(n4), kDomainDiag)[threadIdx.x & 3] : eps[k];
      out[k] = mean[k] + __builtin_sqrtf(var[k]) * e;
    }
  }
}

template <bool RNG>
__global__ __launch_bounds__(kBlock) void local_reparam_bwd_kernel(const float* __restrict__ g,
                                                                  const float* __restrict__ var,
                                                                  const float* __restrict__ eps, uint64_t seed,
                                                                  uint64_t stream_id, float* __restrict__ gvar,
                                                                  int64_t n) {
  const int64_t n4 = n >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 go = ld4_nt(g + 4 * i), v = ld4_nt(var + 4 * i);
    const f32x4 e = RNG ? philox_normal4(seed, stream_id, static_cast<uint64_t>(i), kDomainDiag) : ld4_nt(eps + 4 * i);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (go[j] * e[j]) / (2.0f * __builtin_sqrtf(v[j]));
    st4_nt(gvar + 4 * i, o);
  }
  if (blockIdx.x == 0) {
    const int64_t k = (n4 << 2) + threadIdx.x;
    if (k < n) {
      const float e = RNG ? philox_normal4(seed, stream_id, static_cast<uint64_t>(n4), kDomainDiag)[threadIdx.x & 3] : eps[k];
      gvar[k] = (g[k] * e) / (2.0f * __builtin_sqrtf(var[k]));
    }
  }
}

// Operands of the variance product of the local-reparameterisation layers (bbb_layers.py:66-67,71,150-153):
//   MODE 0: clamp(x^2, 1e-4)                      backward: g * 2 x * [x^2 >= 1e-4]
//   MODE 1: clamp(softplus(rho)^2, 1e-4)          backward: g * [sigma^2 >= 1e-4] * 2 sigma sigmoid(rho)
//   MODE 2: softplus(rho)^2 (BBBConv2d's bias)    backward: g * 2 sigma sigmoid(rho)
// -- two to three ATen launches each in the reference (pow, clamp; softplus, pow, clamp), and twice that in autograd.
constexpr float kVarClamp = 1e-4f;
template <int MODE>
__device__ __forceinline__ float var_operand(float v) {
  if (MODE == 0) return fmaxf(v * v, kVarClamp);
  const float s = softplus(v);
  return MODE == 1 ? fmaxf(s * s, kVarClamp) : s * s;
}
template <int MODE>
__device__ __forceinline__ float var_operand_grad(float g, float v) {
  if (MODE == 0) return v * v >= kVarClamp ? g * (2.0f * v) : 0.f;
  const SoftplusSigmoid ss = softplus_sigmoid(v);
  const bool keep = MODE == 2 || ss.sp * ss.sp >= kVarClamp;
  return keep ? g * (2.0f * ss.sp * ss.sg) : 0.f;
}
template <int MODE, bool BWD>
__global__ __launch_bounds__(kBlock) void var_operand_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                            float* __restrict__ out, int64_t n) {
  const int64_t n4 = n >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 a = ld4_nt(v + 4 * i);
    f32x4 go = {0.f, 0.f, 0.f, 0.f}, o;
    if (BWD) go = ld4_nt(g + 4 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = BWD ? var_operand_grad<MODE>(go[j], a[j]) : var_operand<MODE>(a[j]);
    st4(out + 4 * i, o);                                          // read again right away by the product that follows
  }
  if (blockIdx.x == 0) {
    const int64_t k = (n4 << 2) + threadIdx.x;
    if (k < n) out[k] = BWD ? var_operand_grad<MODE>(g[k], v[k]) : var_operand<MODE>(v[k]);
  }
}

// -------------------------------------------------------------------- KL --
// Per element (bbb.py:20): 0.5 * (2 ln(sp/s) - 1 + (s/sp)^2 + ((mp - m)/sp)^2).
// One exp, two hardware logs and two hardware reciprocals per element; the divisions by the
// prior sigma are multiplications by host-computed reciprocals (<= 1 ulp from the reference's
// divides, inside the stated 3e-6 tolerance) so that the kernel stays HBM-bound.
struct KlConsts {
  float pmu, psig, ipsig, ipsig2, c;
};
__device__ __forceinline__ float kl_elem(float m, float r, const KlConsts& k, bool want_grad, float& gm, float& gr) {
  const SoftplusSigmoid ss = softplus_sigmoid(r);
  const float s = ss.sp;
  const float rs = __builtin_amdgcn_rcpf(s);
  const float a = s * k.ipsig;
  const float b = (k.pmu - m) * k.ipsig;
  const float kl = 0.5f * (2.0f * __logf(k.psig * rs) - 1.0f + a * a + b * b);
  if (want_grad) {
    gm = k.c * ((m - k.pmu) * k.ipsig2);
    gr = k.c * ((s * k.ipsig2 - rs) * ss.sg);
  }
  return kl;
}

template <bool GRAD, bool ACC>
__global__ __launch_bounds__(kBlock) void gauss_kl_kernel(const float* __restrict__ mean, const float* __restrict__ rho,
                                                         float pmu, float psig, float ipsig, float ipsig2,
                                                         float grad_scale, const float* __restrict__ grad_scale_dev,
                                                         float* __restrict__ gmean, float* __restrict__ grho,
                                                         double* __restrict__ partials, int64_t n) {
  __shared__ double smem[kBlock / 64];
  const KlConsts kc{pmu, psig, ipsig, ipsig2, grad_scale * (grad_scale_dev ? grad_scale_dev[0] : 1.0f)};
  const int64_t n4 = n >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  float local = 0.f;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 m = ld4_nt(mean + 4 * i), r = ld4_nt(rho + 4 * i);
    f32x4 gm, gr;
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = 0.f, b = 0.f;
      part += kl_elem(m[j], r[j], kc, GRAD, a, b);
      gm[j] = a;
      gr[j] = b;
    }
    local += part;
    if (GRAD) {
      if (ACC) {
        gm = gm + ld4(gmean + 4 * i);
        gr = gr + ld4(grho + 4 * i);
      }
      st4(gmean + 4 * i, gm);
      st4(grho + 4 * i, gr);
    }
  }
  double acc = static_cast<double>(local);
  if (blockIdx.x == 0) {
    const int64_t k = (n4 << 2) + threadIdx.x;
    if (k < n) {
      float gm, gr;
      acc += static_cast<double>(kl_elem(mean[k], rho[k], kc, GRAD, gm, gr));
      if (GRAD) {
        gmean[k] = ACC ? gmean[k] + gm : gm;
        grho[k] = ACC ? grho[k] + gr : gr;
      }
    }
  }
  const double tot = block_sum(acc, smem);
  if (threadIdx.x == 0) {
    partials[kReduceHeader + blockIdx.x] = tot;
    if (blockIdx.x == 0) partials[0] = static_cast<double>(gridDim.x);
  }
}

// ------------------------------------------------------------ mixture prior --
// MixturePrior.kl_divergence (bbb.py:23-37): -sum log p(mean) with
//   p = pi N(0, s1) + (1 - pi) N(0, s2),  each component's log-density clamped to [-23, 0] before the logaddexp.
// It depends on the means only (rho gets no gradient from it).  Value and gradient in one pass: 4 B/param read,
// 4 or 8 B/param for the gradient (overwrite / accumulate).  Arithmetic in torch's order: Normal.log_prob =
// -(x^2) / (2 var) - log(scale) - log(sqrt(2 pi)) with IEEE divides; logaddexp = max + log1p(exp(-|a - b|)).
struct MixConsts {
  float two_var1, two_var2, log_s1, log_s2, log_w1, log_w2, c;
};
__device__ __forceinline__ float mixture_elem(float x, const MixConsts& k, bool want_grad, float& g) {
  constexpr float kHalfLog2Pi = 0.91893853320467274178f;          // log(sqrt(2 pi))
  const float x2 = x * x;
  const float lp1 = ((-x2) / k.two_var1 - k.log_s1) - kHalfLog2Pi;
  const float lp2 = ((-x2) / k.two_var2 - k.log_s2) - kHalfLog2Pi;
  const float a = k.log_w1 + fminf(fmaxf(lp1, -23.0f), 0.0f);
  const float b = k.log_w2 + fminf(fmaxf(lp2, -23.0f), 0.0f);
  const float m = fmaxf(a, b);
  const float out = m + log1pf(expf(-fabsf(a - b)));
  if (want_grad) {
    // d out / d a = exp(a - out); the clamp passes gradients inside [-23, 0] (bounds included); d lp / d x = -2 x / (2 var)
    const float wa = (lp1 >= -23.0f && lp1 <= 0.0f) ? expf(a - out) : 0.0f;
    const float wb = (lp2 >= -23.0f && lp2 <= 0.0f) ? expf(b - out) : 0.0f;
    const float dlogp = wa * ((-2.0f * x) / k.two_var1) + wb * ((-2.0f * x) / k.two_var2);
    g = k.c * (-dlogp);                                           // the "KL" is MINUS the log-density
  }
  return -out;
}

template <bool GRAD, bool ACC>
__global__ __launch_bounds__(kBlock) void mixture_nll_kernel(const float* __restrict__ mean, MixConsts k,
                                                            const float* __restrict__ grad_scale_dev,
                                                            float* __restrict__ gmean, double* __restrict__ partials,
                                                            int64_t n) {
  __shared__ double smem[kBlock / 64];
  k.c = k.c * (grad_scale_dev ? grad_scale_dev[0] : 1.0f);
  const int64_t n4 = n >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  float local = 0.f;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 m = ld4_nt(mean + 4 * i);
    f32x4 gm;
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = 0.f;
      part += mixture_elem(m[j], k, GRAD, a);
      gm[j] = a;
    }
    local += part;
    if (GRAD) {
      if (ACC) gm = gm + ld4(gmean + 4 * i);
      st4(gmean + 4 * i, gm);
    }
  }
  double acc = static_cast<double>(local);
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < n) {
      float g = 0.f;
      acc += static_cast<double>(mixture_elem(mean[e], k, GRAD, g));
      if (GRAD) gmean[e] = ACC ? gmean[e] + g : g;
    }
  }
  const double tot = block_sum(acc, smem);
  if (threadIdx.x == 0) {
    partials[kReduceHeader + blockIdx.x] = tot;
    if (blockIdx.x == 0) partials[0] = static_cast<double>(gridDim.x);
  }
}

template <bool GRAD, bool ACC>
__global__ __launch_bounds__(kBlock) void l2_kernel(const float* __restrict__ p, float l2_scale, float grad_scale,
                                                   const float* __restrict__ grad_scale_dev, float* __restrict__ g,
                                                   double* __restrict__ partials, int64_t n) {
  __shared__ double smem[kBlock / 64];
  const float c = grad_scale * (grad_scale_dev ? grad_scale_dev[0] : 1.0f) * l2_scale;
  const int64_t n4 = n >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  float local = 0.f;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = ld4_nt(p + 4 * i);
    local += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    if (GRAD) {
      f32x4 o = c * v;
      if (ACC) o = o + ld4(g + 4 * i);
      st4(g + 4 * i, o);
    }
  }
  double acc = static_cast<double>(local);
  if (blockIdx.x == 0) {
    const int64_t k = (n4 << 2) + threadIdx.x;
    if (k < n) {
      acc += static_cast<double>(p[k] * p[k]);
      if (GRAD) g[k] = ACC ? g[k] + c * p[k] : c * p[k];
    }
  }
  const double tot = block_sum(acc, smem);
  if (threadIdx.x == 0) {
    partials[kReduceHeader + blockIdx.x] = tot;
    if (blockIdx.x == 0) partials[0] = static_cast<double>(gridDim.x);
  }
}

// Fixed-order finish: one workgroup, fp64.  out[0] = factor * sum.
__global__ __launch_bounds__(kBlock) void reduce_finish_kernel(const double* __restrict__ partials, float factor,
                                                              float* __restrict__ out) {
  __shared__ double smem[kBlock];
  const int nb = static_cast<int>(partials[0]);
  double s = 0.0;
  for (int b = threadIdx.x; b < nb; b += kBlock) s += partials[kReduceHeader + b];
  smem[threadIdx.x] = s;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) smem[threadIdx.x] += smem[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = static_cast<float>(static_cast<double>(factor) * smem[0]);
}

}  // namespace bde

using namespace bde;

extern "C" size_t bde_reduce_ws_bytes(void) { return sizeof(double) * (kReduceHeader + kReduceMaxBlocks); }

extern "C" int bde_gauss_draw_fwd(const float* mean, const float* rho, const float* eps, uint64_t seed,
                                  uint64_t stream_id, float* w, float* eps_out, int64_t n, void* stream) {
  if (!mean || !rho || !w || n <= 0) return BDE_ERR_INVALID;
  const bool vec = aligned16(mean) && aligned16(rho) && aligned16(w) && (!eps || aligned16(eps)) &&
                   (!eps_out || aligned16(eps_out));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vec) {
    const int grid = stream_grid((n + 3) / 4);
    if (eps)
      hipLaunchKernelGGL(gauss_draw_fwd_kernel<false>, dim3(grid), dim3(kBlock), 0, s, mean, rho, eps, seed, stream_id,
                         w, eps_out, n);
    else
      hipLaunchKernelGGL(gauss_draw_fwd_kernel<true>, dim3(grid), dim3(kBlock), 0, s, mean, rho, eps, seed, stream_id,
                         w, eps_out, n);
  } else {
    const int grid = stream_grid(n);
    if (eps)
      hipLaunchKernelGGL(gauss_draw_fwd_scalar_kernel<false>, dim3(grid), dim3(kBlock), 0, s, mean, rho, eps, seed,
                         stream_id, w, eps_out, n);
    else
      hipLaunchKernelGGL(gauss_draw_fwd_scalar_kernel<true>, dim3(grid), dim3(kBlock), 0, s, mean, rho, eps, seed,
                         stream_id, w, eps_out, n);
  }
  return to_err(hipGetLastError());
}

extern "C" int bde_gauss_draw_bwd(const float* g, const float* rho, const float* eps, uint64_t seed,
                                  uint64_t stream_id, float* gmean, float* grho, int accumulate, int64_t n,
                                  void* stream) {
  if (!g || !rho || !gmean || !grho || n <= 0) return BDE_ERR_INVALID;
  const bool vec = aligned16(g) && aligned16(rho) && aligned16(gmean) && aligned16(grho) && (!eps || aligned16(eps));
  const int grid = vec ? stream_grid((n + 3) / 4) : stream_grid(n);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define BDE_LAUNCH(KERNEL, R, A) \
  hipLaunchKernelGGL((KERNEL<R, A>), dim3(grid), dim3(kBlock), 0, s, g, rho, eps, seed, stream_id, gmean, grho, n)
#define BDE_DISPATCH(KERNEL)                                       \
  if (eps) {                                                       \
    if (accumulate) BDE_LAUNCH(KERNEL, false, true);               \
    else BDE_LAUNCH(KERNEL, false, false);                         \
  } else {                                                         \
    if (accumulate) BDE_LAUNCH(KERNEL, true, true);                \
    else BDE_LAUNCH(KERNEL, true, false);                          \
  }
  if (vec) { BDE_DISPATCH(gauss_draw_bwd_kernel) } else { BDE_DISPATCH(gauss_draw_bwd_scalar_kernel) }
#undef BDE_DISPATCH
#undef BDE_LAUNCH
  return to_err(hipGetLastError());
}

extern "C" int bde_gauss_kl(const float* mean, const float* rho, float prior_mu, float prior_sigma, float grad_scale,
                            const float* grad_scale_dev, float* gmean, float* grho, int accumulate, float* kl_out,
                            void* ws, int64_t n, void* stream) {
  if (!mean || !rho || !ws || n <= 0 || !(prior_sigma > 0.f)) return BDE_ERR_INVALID;
  if ((gmean == nullptr) != (grho == nullptr)) return BDE_ERR_INVALID;
  if (!aligned16(mean) || !aligned16(rho) || (gmean && (!aligned16(gmean) || !aligned16(grho)))) return BDE_ERR_INVALID;
  const int grid = stream_grid((n + 3) / 4, kBlock, kReduceMaxBlocks);
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* part = static_cast<double*>(ws);
  const float ipsig = static_cast<float>(1.0 / static_cast<double>(prior_sigma));
  const float ipsig2 = static_cast<float>(1.0 / (static_cast<double>(prior_sigma) * static_cast<double>(prior_sigma)));
#define BDE_LAUNCH(G, A)                                                                                         \
  hipLaunchKernelGGL((gauss_kl_kernel<G, A>), dim3(grid), dim3(kBlock), 0, s, mean, rho, prior_mu, prior_sigma, \
                     ipsig, ipsig2, grad_scale, grad_scale_dev, gmean, grho, part, n)
  if (!gmean) BDE_LAUNCH(false, false);
  else if (accumulate) BDE_LAUNCH(true, true);
  else BDE_LAUNCH(true, false);
#undef BDE_LAUNCH
  int rc = to_err(hipGetLastError());
  if (rc || !kl_out) return rc;
  hipLaunchKernelGGL(reduce_finish_kernel, dim3(1), dim3(kBlock), 0, s, part, 1.0f, kl_out);
  return to_err(hipGetLastError());
}

extern "C" int bde_mixture_nll(const float* mean, float pi, float sigma1, float sigma2, float grad_scale,
                               const float* grad_scale_dev, float* gmean, int accumulate, float* val_out, void* ws,
                               int64_t n, void* stream) {
  if (!mean || !ws || n <= 0 || !(sigma1 > 0.f) || !(sigma2 > 0.f) || !(pi > 0.f) || !(pi < 1.f)) return BDE_ERR_INVALID;
  if (!aligned16(mean) || (gmean && !aligned16(gmean))) return BDE_ERR_INVALID;
  // torch forms var = scale ** 2 and log(scale), log(pi), log(1 - pi) on float32 tensors
  const float var1 = sigma1 * sigma1, var2 = sigma2 * sigma2;
  const float one_minus_pi = 1.0f - pi;
  const MixConsts k{2.0f * var1, 2.0f * var2, logf(sigma1), logf(sigma2), logf(pi), logf(one_minus_pi), grad_scale};
  const int grid = stream_grid((n + 3) / 4, kBlock, kReduceMaxBlocks);
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* part = static_cast<double*>(ws);
#define BDE_LAUNCH(G, A) \
  hipLaunchKernelGGL((mixture_nll_kernel<G, A>), dim3(grid), dim3(kBlock), 0, s, mean, k, grad_scale_dev, gmean, part, n)
  if (!gmean) BDE_LAUNCH(false, false);
  else if (accumulate) BDE_LAUNCH(true, true);
  else BDE_LAUNCH(true, false);
#undef BDE_LAUNCH
  int rc = to_err(hipGetLastError());
  if (rc || !val_out) return rc;
  hipLaunchKernelGGL(reduce_finish_kernel, dim3(1), dim3(kBlock), 0, s, part, 1.0f, val_out);
  return to_err(hipGetLastError());
}

extern "C" int bde_l2(const float* p, float l2_scale, float grad_scale, const float* grad_scale_dev, float* g,
                      int accumulate, float* val_out, void* ws, int64_t n, void* stream) {
  if (!p || !ws || n <= 0 || !aligned16(p) || (g && !aligned16(g))) return BDE_ERR_INVALID;
  const int grid = stream_grid((n + 3) / 4, kBlock, kReduceMaxBlocks);
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* part = static_cast<double*>(ws);
#define BDE_LAUNCH(G, A) \
  hipLaunchKernelGGL((l2_kernel<G, A>), dim3(grid), dim3(kBlock), 0, s, p, l2_scale, grad_scale, grad_scale_dev, g, part, n)
  if (!g) BDE_LAUNCH(false, false);
  else if (accumulate) BDE_LAUNCH(true, true);
  else BDE_LAUNCH(true, false);
#undef BDE_LAUNCH
  int rc = to_err(hipGetLastError());
  if (rc || !val_out) return rc;
  hipLaunchKernelGGL(reduce_finish_kernel, dim3(1), dim3(kBlock), 0, s, part, 0.5f * l2_scale, val_out);
  return to_err(hipGetLastError());
}

extern "C" int bde_local_reparam_fwd(const float* mean, const float* var, const float* eps, uint64_t seed,
                                     uint64_t stream_id, float* out, int64_t n, void* stream) {
  if (!mean || !var || !out || n <= 0) return BDE_ERR_INVALID;
  if (!aligned16(mean) || !aligned16(var) || !aligned16(out) || (eps && !aligned16(eps))) return BDE_ERR_INVALID;
  const int grid = stream_grid((n + 3) / 4);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (eps)
    hipLaunchKernelGGL(local_reparam_fwd_kernel<false>, dim3(grid), dim3(kBlock), 0, s, mean, var, eps, seed, stream_id, out, n);
  else
    hipLaunchKernelGGL(local_reparam_fwd_kernel<true>, dim3(grid), dim3(kBlock), 0, s, mean, var, eps, seed, stream_id, out, n);
  return to_err(hipGetLastError());
}

extern "C" int bde_local_reparam_bwd(const float* g, const float* var, const float* eps, uint64_t seed,
                                     uint64_t stream_id, float* gvar, int64_t n, void* stream) {
  if (!g || !var || !gvar || n <= 0) return BDE_ERR_INVALID;
  if (!aligned16(g) || !aligned16(var) || !aligned16(gvar) || (eps && !aligned16(eps))) return BDE_ERR_INVALID;
  const int grid = stream_grid((n + 3) / 4);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (eps)
    hipLaunchKernelGGL(local_reparam_bwd_kernel<false>, dim3(grid), dim3(kBlock), 0, s, g, var, eps, seed, stream_id, gvar, n);
  else
    hipLaunchKernelGGL(local_reparam_bwd_kernel<true>, dim3(grid), dim3(kBlock), 0, s, g, var, eps, seed, stream_id, gvar, n);
  return to_err(hipGetLastError());
}

extern "C" int bde_var_operand_fwd(const float* v, int mode, float* out, int64_t n, void* stream) {
  if (!v || !out || n <= 0 || mode < 0 || mode > 2 || !aligned16(v) || !aligned16(out)) return BDE_ERR_INVALID;
  const int grid = stream_grid((n + 3) / 4);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* none = nullptr;
  if (mode == 0) hipLaunchKernelGGL((var_operand_kernel<0, false>), dim3(grid), dim3(kBlock), 0, s, none, v, out, n);
  else if (mode == 1) hipLaunchKernelGGL((var_operand_kernel<1, false>), dim3(grid), dim3(kBlock), 0, s, none, v, out, n);
  else hipLaunchKernelGGL((var_operand_kernel<2, false>), dim3(grid), dim3(kBlock), 0, s, none, v, out, n);
  return to_err(hipGetLastError());
}

extern "C" int bde_var_operand_bwd(const float* g, const float* v, int mode, float* gv, int64_t n, void* stream) {
  if (!g || !v || !gv || n <= 0 || mode < 0 || mode > 2 || !aligned16(g) || !aligned16(v) || !aligned16(gv))
    return BDE_ERR_INVALID;
  const int grid = stream_grid((n + 3) / 4);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (mode == 0) hipLaunchKernelGGL((var_operand_kernel<0, true>), dim3(grid), dim3(kBlock), 0, s, g, v, gv, n);
  else if (mode == 1) hipLaunchKernelGGL((var_operand_kernel<1, true>), dim3(grid), dim3(kBlock), 0, s, g, v, gv, n);
  else hipLaunchKernelGGL((var_operand_kernel<2, true>), dim3(grid), dim3(kBlock), 0, s, g, v, gv, n);
  return to_err(hipGetLastError());
}

// bde_init(): load this translation unit's code object on the current device now (HIP otherwise uploads it at the
// first launch of one of its kernels).  Internal to the library (not exported).
extern "C" __attribute__((visibility("hidden"))) int bde_internal_load_gauss(void) {
  hipFuncAttributes attr;
  return bde::to_err(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&bde::gauss_draw_fwd_kernel<true>)));
}
