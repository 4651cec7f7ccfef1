// Local-reparameterisation forward of a mean-field linear layer, fused.
//
// Reference: BBBLinear.forward, sampling="activations" (src/algos/bbb_layers.py:61-80):
//   activation_mean = x W_mu^T + b_mu
//   activation_var  = clamp(x^2, 1e-4) clamp(softplus(W_rho)^2, 1e-4)^T + clamp(softplus(b_rho)^2, 1e-4)
//   out             = activation_mean + sqrt(activation_var) * eps
// i.e. ~14 ATen launches (softplus, square, clamp for W and b, square and clamp for x, two GEMMs with bias, sqrt,
// normal_, mul, add) that write and re-read sigma^2, its clamp and x^2 -- six extra passes over the [O, I]
// weight-shaped tensors.  For the batch sizes this path sees (tens of rows) the layer is a weight-streaming
// problem: W_mu and W_rho should be read ONCE.  Here:
//
//   lrt_partial_kernel   one wave per (32-output tile, K-slice): streams its slice of W_mu / W_rho once (float4 per
//                        lane, 32 rows x 8 columns per wave-load), forms sigma^2 on the fly (softplus from ONE
//                        exponential, bde_common.hpp) and feeds TWO accumulator tiles on the f32 MFMA
//                        (v_mfma_f32_32x32x2_f32): mean += x W_mu^T and var += clamp(x^2) clamp(sigma^2)^T, for up to
//                        4 batch tiles of 32 rows held in registers; split-K partials go to a workspace
//   lrt_wide_kernel      the same products for wide layers (O*I >= 2^20): coalesced loads staged through per-wave LDS
//                        tiles, four K-quarters per workgroup summed in LDS (see the comment at the kernel)
//   lrt_finish_kernel    fixed-order sum of the K-slices (bit-reproducible, no atomics), bias terms, sqrt, noise
//                        (caller-supplied or in-kernel Philox), writes out and the variance the backward needs
//
// HBM traffic 8*O*I (weights, once) + O(B*(I + O)) instead of ~32*O*I.  The backward pass is lrt_bwd.hip.
#include "bde_common.hpp"

namespace bde {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// tools/lrt_ab.py only (never defined in the product build): where the time of the wide-layer kernels goes.
//   BDE_EXP_NOMFMA   the matrix products replaced by one VALU FMA each (memory + LDS + issue side alone)
//   BDE_EXP_ALIAS    every tile reads / writes the FIRST tile's rows of the weight-shaped arrays (no HBM stream)
#ifdef BDE_EXP_NOMFMA
#define BDE_MFMA32(a, b, c) ([&] { auto c_ = (c); c_[0] = __builtin_fmaf((a), (b), c_[0]); return c_; }())
#else
#define BDE_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#endif
#ifdef BDE_EXP_ALIAS
#define BDE_EXP_ROW(r) ((r) & 31)
#else
#define BDE_EXP_ROW(r) (r)
#endif

constexpr int kLrtMinKSlice = 64;     // columns of W per wave at least (8 k-steps of 8)
constexpr int kLrtTargetWaves = 2048; // (o-tile, K-slice) units wanted: 2 waves per SIMD on 256 CUs
constexpr int kLrtWavesPerWG = 4;
constexpr float kLrtClamp = 1e-4f;    // bbb_layers.py:71-72

// float4 of row `r` at columns [k, k+4); rows / columns outside the matrix read as 0.  ALIGNED: I % 4 == 0 and a
// 16-byte aligned base, so the four columns are one load and either all valid or all invalid.
template <bool ALIGNED>
__device__ __forceinline__ f32x4 lrt_load4(const float* __restrict__ base, int64_t ld, int r, int rows, int k, int cols) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (r < rows) {
    const float* p = base + static_cast<int64_t>(r) * ld + k;
    if (ALIGNED) {
      if (k < cols) v = ld4(p);
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (k + c < cols) v[c] = p[c];
    }
  }
  return v;
}

// NB = batch tiles of 32 rows handled by one wave (B <= 32 * NB).
template <int NB, bool ALIGNED>
__global__ __launch_bounds__(kLrtWavesPerWG * 64) void lrt_partial_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ w_mu, const float* __restrict__ w_rho, int B,
    int I, int O, int n_slices, int kslice, float* __restrict__ ws) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int unit = blockIdx.x * kLrtWavesPerWG + wave;            // (o-tile, K-slice)
  const int o_tiles = (O + 31) >> 5;
  if (unit >= o_tiles * n_slices) return;
  const int ot = unit / n_slices, sl = unit % n_slices;
  const int r = lane & 31, h = lane >> 5;                         // row within the tile, k-half
  const int k0 = sl * kslice, k1 = min(I, k0 + kslice);

  f32x16 accm[NB], accv[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) accm[t] = accv[t] = f32x16{};

  // register double buffering: the loads of k-step n+1 are issued before the softplus / MFMA work of step n
  struct Operands {
    f32x4 wm, wr, xs[NB];
  };
  auto load = [&](Operands& q, int k) {
    q.wm = lrt_load4<ALIGNED>(w_mu, I, ot * 32 + r, O, k, k1);
    q.wr = lrt_load4<ALIGNED>(w_rho, I, ot * 32 + r, O, k, k1);
#pragma unroll
    for (int t = 0; t < NB; ++t) q.xs[t] = lrt_load4<ALIGNED>(x, ldx, t * 32 + r, B, k, k1);
  };
  Operands cur, nxt;
  const int kend = k1 + 4 * h;                                    // both halves run the same number of steps
  int k = k0 + 4 * h;
  if (k < kend) load(cur, k);
  for (; k < kend; k += 8) {
    if (k + 8 < kend) load(nxt, k + 8);
    f32x4 s2;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float s = softplus(cur.wr[c]);
      s2[c] = fmaxf(s * s, kLrtClamp);                            // clamp(softplus(rho)^2, 1e-4)
    }
    const bool live = (ot * 32 + r < O) && (k < k1);              // padding rows / columns contribute nothing
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool ok = live && (ALIGNED || k + c < k1);
      const float bm = ok ? cur.wm[c] : 0.f, bv = ok ? s2[c] : 0.f;
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const float a = cur.xs[t][c];
        const bool xok = (t * 32 + r < B) && (k < k1) && (ALIGNED || k + c < k1);
        const float a2 = xok ? fmaxf(a * a, kLrtClamp) : 0.f;     // clamp(x^2, 1e-4)
        accm[t] = BDE_MFMA32(a, bm, accm[t]);
        accv[t] = BDE_MFMA32(a2, bv, accv[t]);
      }
    }
    cur = nxt;
  }
  // C[b][o]: lane holds column o = lane & 31, rows b = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int o_pad = o_tiles * 32, b_pad = NB * 32;
  float* base = ws + static_cast<int64_t>(sl) * 2 * b_pad * o_pad;
#pragma unroll
  for (int t = 0; t < NB; ++t) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int b = t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      base[static_cast<int64_t>(b) * o_pad + ot * 32 + r] = accm[t][reg];
      base[static_cast<int64_t>(b_pad + b) * o_pad + ot * 32 + r] = accv[t][reg];
    }
  }
}

// Wide layers (I % 4 == 0, O * I >= 2^20): the row-per-lane loads of lrt_partial_kernel touch 32 cache lines per
// instruction and use 16 bytes of each (89 us for 2 x 67 MB at B = 64; this kernel: 71 us, of which the 16.7 M
// softplus evaluations are ~25 us of VALU time and the fp32 MFMA products ~27 us -- the layer is co-bound, not HBM-bound).
// Here one workgroup = 4 waves owns (32-output tile, K-group); wave w streams its quarter of the group in chunks of
// 32 columns with COALESCED float4 loads (8 lanes cover 128 bytes of a row, 8 rows per instruction), forms sigma^2
// in that layout, parks the chunk in its private LDS tiles (row stride 36 floats: 16-byte aligned, conflict-free
// b128 reads in the MFMA layout) and reads the MFMA operands back from there; the next chunk's global loads are in
// flight while the current chunk's 16 * NB MFMAs run.  No barrier inside the loop (the tiles are per wave); at the
// end the four accumulator sets are summed through LDS in wave order (fixed order, bit-reproducible) and ONE partial
// per (tile, group) goes to the workspace: 4 x fewer partial bytes than one partial per wave.
constexpr int kWideChunk = 32, kWideLd = 36, kWideSub = 4;

// PRE: `w_rho` is not rho but the cached clamp(softplus(rho)^2, 1e-4) of this weight version (bde_lrt_sigma_cache):
// same bytes, no transcendental in the loop.
template <int NB, bool PRE>
__global__ __launch_bounds__(kWideSub * 64) void lrt_wide_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ w_mu, const float* __restrict__ w_rho, int B,
    int I, int O, int n_groups, int kgroup, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int kTileFloats = (64 + NB * 32) * kWideLd;            // W_mu, sigma^2 and NB x tiles of 32 rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ot = blockIdx.x / n_groups, gp = blockIdx.x % n_groups;
  const int ksub = kgroup / kWideSub;                               // host: kgroup % (4 * 32) == 0
  const int k0 = min(I, gp * kgroup + wave * ksub), k1 = min(I, k0 + ksub);
  float* Wm = lds + wave * kTileFloats;
  float* S2 = Wm + 32 * kWideLd;
  float* X = S2 + 32 * kWideLd;
  const int lr = lane >> 3, lc = 4 * (lane & 7);                   // coalesced layout: row 8 j + lr, columns lc .. lc+3
  const int r = lane & 31, h = lane >> 5;                          // MFMA layout

  struct Stage {
    f32x4 wm[4], wr[4], xs[NB * 4];
  };
  auto gload = [&](Stage& q, int kc) {
    const int col = (kc + lc < k1) ? kc + lc : k0;                 // past the slice: any valid address, masked at the stash
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = BDE_EXP_ROW(min(ot * 32 + 8 * j + lr, O - 1));
      q.wm[j] = ld4(w_mu + static_cast<int64_t>(row) * I + col);
      q.wr[j] = ld4(w_rho + static_cast<int64_t>(row) * I + col);
    }
#pragma unroll
    for (int j = 0; j < NB * 4; ++j) q.xs[j] = ld4(x + static_cast<int64_t>(min(8 * j + lr, B - 1)) * ldx + col);
  };
  auto stash = [&](const Stage& q, int kc) {
    const bool c_ok = kc + lc < k1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = c_ok && (ot * 32 + 8 * j + lr < O);          // padding rows / columns of W contribute nothing
      f32x4 m, v;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        m[c] = ok ? q.wm[j][c] : 0.f;
        if (PRE) {
          v[c] = ok ? q.wr[j][c] : 0.f;
        } else {
          const float sp = softplus(q.wr[j][c]);
          v[c] = ok ? fmaxf(sp * sp, kLrtClamp) : 0.f;             // clamp(softplus(rho)^2, 1e-4)
        }
      }
      *reinterpret_cast<f32x4*>(Wm + (8 * j + lr) * kWideLd + lc) = m;
      *reinterpret_cast<f32x4*>(S2 + (8 * j + lr) * kWideLd + lc) = v;
    }
#pragma unroll
    for (int j = 0; j < NB * 4; ++j) *reinterpret_cast<f32x4*>(X + (8 * j + lr) * kWideLd + lc) = q.xs[j];
  };

  f32x16 accm[NB], accv[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) accm[t] = accv[t] = f32x16{};
  auto products = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bm = *reinterpret_cast<const f32x4*>(Wm + r * kWideLd + 8 * q + 4 * h);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(S2 + r * kWideLd + 8 * q + 4 * h);
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(X + (t * 32 + r) * kWideLd + 8 * q + 4 * h);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          accm[t] = BDE_MFMA32(a[c], bm[c], accm[t]);
          accv[t] = BDE_MFMA32(fmaxf(a[c] * a[c], kLrtClamp), bv[c], accv[t]);
        }
      }
    }
  };
  Stage st;
  if (k0 < k1) gload(st, k0);
  for (int kc = k0; kc < k1; kc += kWideChunk) {
    stash(st, kc);
    if (kc + kWideChunk < k1) gload(st, kc + kWideChunk);
    products();
  }
  // the four K-quarters of this group, summed in wave order
  __syncthreads();
  float* red = lds;                                                // [3][2 * NB * 16][64]
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        red[((wave - 1) * (2 * NB * 16) + (2 * t) * 16 + reg) * 64 + lane] = accm[t][reg];
        red[((wave - 1) * (2 * NB * 16) + (2 * t + 1) * 16 + reg) * 64 + lane] = accv[t][reg];
      }
  }
  __syncthreads();
  if (wave != 0) return;
  const int o_tiles = (O + 31) >> 5, o_pad = o_tiles * 32, b_pad = NB * 32;
  float* base = ws + static_cast<int64_t>(gp) * 2 * b_pad * o_pad;
#pragma unroll
  for (int t = 0; t < NB; ++t) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      float m = accm[t][reg], v = accv[t][reg];
#pragma unroll
      for (int w = 0; w < kWideSub - 1; ++w) {
        m += red[(w * (2 * NB * 16) + (2 * t) * 16 + reg) * 64 + lane];
        v += red[(w * (2 * NB * 16) + (2 * t + 1) * 16 + reg) * 64 + lane];
      }
      const int b = t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;   // C[b][o]: lane holds column o = lane & 31
      base[static_cast<int64_t>(b) * o_pad + ot * 32 + r] = m;
      base[static_cast<int64_t>(b_pad + b) * o_pad + ot * 32 + r] = v;
    }
  }
}

template <bool RNG>
__global__ __launch_bounds__(kBlock) void lrt_finish_kernel(const float* __restrict__ ws, int n_slices, int b_pad,
                                                           int o_pad, const float* __restrict__ b_mu,
                                                           const float* __restrict__ b_rho, int clamp_bias,
                                                           const float* __restrict__ eps, uint64_t seed,
                                                           uint64_t stream_id, float* __restrict__ out,
                                                           float* __restrict__ var_out, int B, int O) {
  const int64_t n = static_cast<int64_t>(B) * O;
  const int64_t slice_stride = static_cast<int64_t>(2) * b_pad * o_pad;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(e / O), o = static_cast<int>(e % O);
    const float* pm = ws + static_cast<int64_t>(b) * o_pad + o;
    const float* pv = ws + static_cast<int64_t>(b_pad + b) * o_pad + o;
    float m = 0.f, v = 0.f;
    for (int s = 0; s < n_slices; ++s) {                          // fixed order
      m += pm[s * slice_stride];
      v += pv[s * slice_stride];
    }
    if (b_mu) m += b_mu[o];
    if (b_rho) {
      const float sb = softplus(b_rho[o]);
      const float vb = sb * sb;
      v += clamp_bias ? fmaxf(vb, kLrtClamp) : vb;                // BBBLinear clamps the bias variance, BBBConv2d does not
    }
    const float z = RNG ? philox_normal4(seed, stream_id, static_cast<uint64_t>(e >> 2), kDomainDiag)[e & 3] : eps[e];
    if (var_out) var_out[e] = v;
    out[e] = m + __builtin_sqrtf(v) * z;
  }
}

}  // namespace bde

using namespace bde;

// Split of the reduction dimension: enough (o-tile, K-slice) units to fill the chip, but no more K-slices than that
// needs (every slice costs a [2, B, O] partial in the workspace); slices are multiples of 8 columns.
struct LrtPlan {
  int n_slices, kslice;
};
static inline LrtPlan lrt_plan(int I, int O) {
  const int o_tiles = (O + 31) / 32;
  int want = (kLrtTargetWaves + o_tiles - 1) / o_tiles;
  const int most = (I + kLrtMinKSlice - 1) / kLrtMinKSlice;
  if (want > most) want = most;
  if (want < 1) want = 1;
  int kslice = ((I + want - 1) / want + 7) / 8 * 8;
  if (kslice < kLrtMinKSlice) kslice = kLrtMinKSlice;
  return LrtPlan{(I + kslice - 1) / kslice, kslice};
}

// Wide path: (o-tile, K-group) workgroups, about two per CU; groups are multiples of 128 columns (4 waves x 32).
#ifndef BDE_LRT_WIDE_TARGET_WGS
#define BDE_LRT_WIDE_TARGET_WGS 512
#endif
#ifndef BDE_LRT_WIDE
#define BDE_LRT_WIDE 1
#endif
struct LrtWidePlan {
  int n_groups, kgroup;
};
static inline bool lrt_wide_shape(int I, int O) {
  return BDE_LRT_WIDE && I % 4 == 0 && I >= 512 && static_cast<int64_t>(I) * O >= (int64_t{1} << 20);
}
static inline LrtWidePlan lrt_wide_plan(int I, int O) {
  const int o_tiles = (O + 31) / 32;
  int want = (BDE_LRT_WIDE_TARGET_WGS + o_tiles / 2) / o_tiles;
  const int unit = kWideSub * kWideChunk;
  const int most = (I + unit - 1) / unit;
  if (want > most) want = most;
  if (want < 1) want = 1;
  const int kgroup = ((I + want - 1) / want + unit - 1) / unit * unit;
  return LrtWidePlan{(I + kgroup - 1) / kgroup, kgroup};
}
template <int NB, bool PRE>
static int lrt_wide_launch(const float* x, int64_t ldx, const float* w_mu, const float* w_rho, int B, int I, int O,
                           const LrtWidePlan& plan, float* ws, hipStream_t s) {
  constexpr int tile_bytes = kWideSub * (64 + NB * 32) * kWideLd * 4, red_bytes = 3 * 2 * NB * 16 * 64 * 4;
  constexpr int lds_bytes = tile_bytes > red_bytes ? tile_bytes : red_bytes;
  static int attr_rc = to_err(hipFuncSetAttribute(reinterpret_cast<const void*>(&lrt_wide_kernel<NB, PRE>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  if (attr_rc) return attr_rc;
  const int o_tiles = (O + 31) / 32;
  hipLaunchKernelGGL((lrt_wide_kernel<NB, PRE>), dim3(o_tiles * plan.n_groups), dim3(kWideSub * 64), lds_bytes, s, x, ldx,
                     w_mu, w_rho, B, I, O, plan.n_groups, plan.kgroup, ws);
  return to_err(hipGetLastError());
}

// sigma^2 = clamp(softplus(rho)^2, 1e-4) and d sigma^2 / d rho = [sigma^2 >= 1e-4] * 2 sigma sigmoid(rho) of a weight
// matrix, once per weight VERSION: the wide-layer kernels of all Monte-Carlo forward / backward passes of an optimizer
// step (bbb.py:63-67 runs mc_samples of them between two base_optimizer.step() calls) then read these instead of
// evaluating softplus / sigmoid per weight per pass.  Same expressions as the kernels: bit-identical results.
__global__ __launch_bounds__(kBlock) void lrt_sigma_cache_kernel(const float* __restrict__ rho, float* __restrict__ s2,
                                                                float* __restrict__ ds2, int64_t n) {
  const int64_t n4 = n >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 r = ld4_nt(rho + 4 * i);
    f32x4 a, b;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const SoftplusSigmoid sp = softplus_sigmoid(r[c]);
      const float keep = sp.sp * sp.sp >= kLrtClamp ? 1.f : 0.f;
      a[c] = fmaxf(sp.sp * sp.sp, kLrtClamp);
      b[c] = keep * (2.0f * sp.sp * sp.sg);
    }
    st4(s2 + 4 * i, a);
    if (ds2) st4(ds2 + 4 * i, b);
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < n) {
      const SoftplusSigmoid sp = softplus_sigmoid(rho[e]);
      const float keep = sp.sp * sp.sp >= kLrtClamp ? 1.f : 0.f;
      s2[e] = fmaxf(sp.sp * sp.sp, kLrtClamp);
      if (ds2) ds2[e] = keep * (2.0f * sp.sp * sp.sg);
    }
  }
}

extern "C" int bde_lrt_linear_supported(int B, int I, int O) {
  return B >= 1 && B <= 128 && I >= 1 && O >= 1 && static_cast<int64_t>(I) * O <= (int64_t{1} << 31);
}

extern "C" size_t bde_lrt_linear_ws_bytes(int B, int I, int O) {
  if (!bde_lrt_linear_supported(B, I, O)) return 0;
  const int nb = (B + 31) / 32 == 3 ? 4 : (B + 31) / 32;
  const size_t o_pad = static_cast<size_t>((O + 31) / 32) * 32, b_pad = static_cast<size_t>(nb) * 32;
  int slices = lrt_plan(I, O).n_slices;
  if (lrt_wide_shape(I, O)) slices = std::max(slices, lrt_wide_plan(I, O).n_groups);
  return sizeof(float) * static_cast<size_t>(slices) * 2 * b_pad * o_pad;
}

extern "C" int bde_lrt_sigma_cache_wanted(int I, int O) { return lrt_wide_shape(I, O) ? 1 : 0; }

extern "C" int bde_lrt_sigma_cache(const float* w_rho, float* s2, float* ds2, int64_t n, void* stream) {
  if (!w_rho || !s2 || n < 1 || !aligned16(w_rho) || !aligned16(s2) || (ds2 && !aligned16(ds2))) return BDE_ERR_INVALID;
  hipLaunchKernelGGL(lrt_sigma_cache_kernel, dim3(stream_grid((n + 3) / 4)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                     w_rho, s2, ds2, n);
  return to_err(hipGetLastError());
}

extern "C" int bde_lrt_linear_fwd(const float* x, int64_t ldx, const float* w_mu, const float* w_rho, const float* w_s2,
                                  const float* b_mu, const float* b_rho, int clamp_bias_var, const float* eps,
                                  uint64_t seed, uint64_t stream_id, float* out, float* var_out, int B, int I, int O,
                                  void* ws, void* stream) {
  if (!x || !w_mu || !w_rho || !out || !ws || !bde_lrt_linear_supported(B, I, O) || ldx < I) return BDE_ERR_INVALID;
  if ((b_mu == nullptr) != (b_rho == nullptr)) return BDE_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const LrtPlan plan = lrt_plan(I, O);
  int n_slices = plan.n_slices;
  const int o_tiles = (O + 31) / 32;
  const int nbt = (B + 31) / 32;
  const int nb = nbt == 3 ? 4 : nbt;
  const bool aligned = (I % 4 == 0) && (ldx % 4 == 0) && aligned16(x) && aligned16(w_mu) && aligned16(w_rho);
  float* wsf = static_cast<float*>(ws);
  const bool wide = aligned && lrt_wide_shape(I, O);
  int rc = 0;
  if (wide) {
    const LrtWidePlan wp = lrt_wide_plan(I, O);
    n_slices = wp.n_groups;
    if (w_s2 && aligned16(w_s2))      // the cached sigma^2 of this weight version instead of rho
      rc = nb == 1 ? lrt_wide_launch<1, true>(x, ldx, w_mu, w_s2, B, I, O, wp, wsf, s)
                   : nb == 2 ? lrt_wide_launch<2, true>(x, ldx, w_mu, w_s2, B, I, O, wp, wsf, s)
                             : lrt_wide_launch<4, true>(x, ldx, w_mu, w_s2, B, I, O, wp, wsf, s);
    else
      rc = nb == 1 ? lrt_wide_launch<1, false>(x, ldx, w_mu, w_rho, B, I, O, wp, wsf, s)
                   : nb == 2 ? lrt_wide_launch<2, false>(x, ldx, w_mu, w_rho, B, I, O, wp, wsf, s)
                             : lrt_wide_launch<4, false>(x, ldx, w_mu, w_rho, B, I, O, wp, wsf, s);
    if (rc) return rc;
  } else {
    const int units = o_tiles * n_slices;
    const int grid = (units + kLrtWavesPerWG - 1) / kLrtWavesPerWG;
#define BDE_LRT(NB, AL)                                                                                              \
  hipLaunchKernelGGL((lrt_partial_kernel<NB, AL>), dim3(grid), dim3(kLrtWavesPerWG * 64), 0, s, x, ldx, w_mu, w_rho, B, \
                     I, O, n_slices, plan.kslice, wsf)
    if (nb == 1) { if (aligned) BDE_LRT(1, true); else BDE_LRT(1, false); }
    else if (nb == 2) { if (aligned) BDE_LRT(2, true); else BDE_LRT(2, false); }
    else { if (aligned) BDE_LRT(4, true); else BDE_LRT(4, false); }
#undef BDE_LRT
    rc = to_err(hipGetLastError());
    if (rc) return rc;
  }
  const int fgrid = stream_grid(static_cast<int64_t>(B) * O);
  const int o_pad = o_tiles * 32, b_pad = nb * 32;
  if (eps)
    hipLaunchKernelGGL(lrt_finish_kernel<false>, dim3(fgrid), dim3(kBlock), 0, s, wsf, n_slices, b_pad, o_pad, b_mu, b_rho,
                       clamp_bias_var, eps, seed, stream_id, out, var_out, B, O);
  else
    hipLaunchKernelGGL(lrt_finish_kernel<true>, dim3(fgrid), dim3(kBlock), 0, s, wsf, n_slices, b_pad, o_pad, b_mu, b_rho,
                       clamp_bias_var, eps, seed, stream_id, out, var_out, B, O);
  return to_err(hipGetLastError());
}

// bde_init(): load this translation unit's code object on the current device now (HIP otherwise uploads it at the
// first launch of one of its kernels).  Internal to the library (not exported).
extern "C" __attribute__((visibility("hidden"))) int bde_internal_load_lrt(void) {
  hipFuncAttributes attr;
  return bde::to_err(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&bde::lrt_finish_kernel<true>)));
}
