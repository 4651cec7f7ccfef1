// Shared device/host helpers for libbde_hip (gfx950 / CDNA4 only).
//
// Everything on this path is an HBM-bound stream over flat fp32 parameter
// buffers, so the helpers are about: 16-byte coalesced accesses, enough
// workgroups to fill 256 CUs with grid-stride loops, deterministic two-stage
// reductions (wave shuffles -> LDS -> fixed-order fp64 finish), and a
// counter-based RNG whose output does not depend on the launch geometry.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>

#include "../../include/bde_hip.h"

namespace bde {

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kCUs = 256;            // MI355X
constexpr int kBlock = 256;          // 4 waves per workgroup for streaming kernels
constexpr int kMaxStreamBlocks = kCUs * 8;   // 8 workgroups of 256 per CU = 32 waves/CU

using f32x4 = __attribute__((ext_vector_type(4))) float;

static inline int to_err(hipError_t e) { return e == hipSuccess ? 0 : -static_cast<int>(e); }

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Grid for a grid-stride streaming kernel over `n_items` per-thread work items.
static inline int stream_grid(int64_t n_items, int block = kBlock, int max_blocks = kMaxStreamBlocks) {
  int64_t g = (n_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return static_cast<int>(g);
}

__device__ __forceinline__ f32x4 ld4(const float* __restrict__ p) {
  return *reinterpret_cast<const f32x4*>(p);
}
__device__ __forceinline__ void st4(float* __restrict__ p, f32x4 v) {
  *reinterpret_cast<f32x4*>(p) = v;
}
// Streaming store: the line is not re-read by this kernel.
__device__ __forceinline__ void st4_nt(float* __restrict__ p, f32x4 v) {
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
}
// How the single-output sampling kernels (swag_sample, gauss_draw_fwd, local_reparam_fwd) store their result row:
// with PLAIN stores.  The sampled weights / activations are read again by the very next kernels (the model's
// forward), so they are not a streaming output; measured +2-4 % on the 12*D kernels and neutral on swag_sample
// (A/B over alternating processes, profiles/r02_output_store_ab.txt).
#ifndef BDE_OUT_ST
#define BDE_OUT_ST st4
#endif
__device__ __forceinline__ f32x4 ld4_nt(const float* __restrict__ p) {
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
}

// ---- reductions ---------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

// Sum over the workgroup; result valid in thread 0.  `smem` >= blockDim/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* smem) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = wave_sum(v);
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += smem[i];
  }
  __syncthreads();
  return r;
}

// ---- Philox4x32-10 + Box-Muller ----------------------------------------
// Counter layout: (lo32(idx), hi32(idx), lo32(stream), hi32(stream) ^ domain),
// key = seed.  `idx` is the float4 group index of the element, so the normal
// attached to element e of stream s is a pure function of (seed, s, e).
// The bijection is the published Philox4x32-10 (Salmon et al., SC'11; the
// Random123 known-answer vectors are checked in tests/ through bde_philox_bits).
// Per round: two v_mad_u64_u32 (hi and lo of a 32x32 product in one
// instruction) and two v_bitop3_b32 (hi ^ c ^ key as ONE gfx950 instruction;
// the compiler emits two v_xor_b32 for the same expression) -- 40 VALU
// instructions per 128 random bits instead of 59.
// Rounds: Philox4x32-10 is the published default (and what every draw of the BBB / iVON / layer kernels uses);
// Philox4x32-7 is the fewest rounds Salmon et al. report as Crush-resistant (it passes BigCrush; the other three are
// their safety margin).  The SWAG samplers use 7 (kSwagPhiloxRounds): their epilogue generates S x D normals per pass
// and is co-bound by exactly these multiplies -- 14 instead of 20 v_mad_u64_u32 per 128 bits moved the batched
// sampler from 0.68 to 0.71 of the HBM peak (profiles/r03_swag_layout_ab.txt).
constexpr int kPhiloxRounds = 10;
#ifndef BDE_SWAG_PHILOX_ROUNDS
#define BDE_SWAG_PHILOX_ROUNDS 7
#endif
constexpr int kSwagPhiloxRounds = BDE_SWAG_PHILOX_ROUNDS;
struct Philox {
  static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  __device__ __forceinline__ static uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
  }
  template <int ROUNDS>
  __device__ __forceinline__ static uint4 rounds(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const uint64_t p0 = static_cast<uint64_t>(M0) * c.x, p1 = static_cast<uint64_t>(M1) * c.z;
      const uint32_t hi0 = static_cast<uint32_t>(p0 >> 32), lo0 = static_cast<uint32_t>(p0);
      const uint32_t hi1 = static_cast<uint32_t>(p1 >> 32), lo1 = static_cast<uint32_t>(p1);
      c = make_uint4(xor3(hi1, c.y, k.x), lo1, xor3(hi0, c.w, k.y), lo0);
      k.x += W0;      // the key schedule is wave-uniform: scalar adds
      k.y += W1;
    }
    return c;
  }
};

enum : uint32_t { kDomainDiag = 0x0u, kDomainLowRank = 0x80000000u };

template <int ROUNDS = kPhiloxRounds>
__device__ __forceinline__ uint4 philox_bits4(uint64_t seed, uint64_t stream_id, uint64_t idx4, uint32_t domain) {
  const uint4 c = make_uint4(static_cast<uint32_t>(idx4), static_cast<uint32_t>(idx4 >> 32),
                             static_cast<uint32_t>(stream_id), static_cast<uint32_t>(stream_id >> 32) ^ domain);
  const uint2 k = make_uint2(static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
  return Philox::rounds<ROUNDS>(c, k);
}

// Box-Muller on the 24 high bits of each word.  u0, u2 in [2^-24, 1]: never zero or denormal,
// so the bare hardware v_log_f32 / v_sqrt_f32 (1 ulp; the libm wrappers add ~12 instructions
// of denormal scaling and Newton fix-up per call) are exact enough and the radius is at most
// sqrt(2 * 24 ln 2) = 5.77.  v_sin_f32 / v_cos_f32 take their argument in revolutions.
__device__ __forceinline__ f32x4 box_muller4(uint4 r) {
  const float u0 = (static_cast<float>(r.x >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u1 = static_cast<float>(r.y >> 8) * (1.0f / 16777216.0f);
  const float u2 = (static_cast<float>(r.z >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u3 = static_cast<float>(r.w >> 8) * (1.0f / 16777216.0f);
  constexpr float kM2Ln2 = -1.3862943611198906f;                 // -2 ln 2: -2 ln(u) = kM2Ln2 * log2(u)
  const float r0 = __builtin_amdgcn_sqrtf(kM2Ln2 * __builtin_amdgcn_logf(u0));
  const float r1 = __builtin_amdgcn_sqrtf(kM2Ln2 * __builtin_amdgcn_logf(u2));
  f32x4 z;
  z.x = r0 * __builtin_amdgcn_cosf(u1);
  z.y = r0 * __builtin_amdgcn_sinf(u1);
  z.z = r1 * __builtin_amdgcn_cosf(u3);
  z.w = r1 * __builtin_amdgcn_sinf(u3);
  return z;
}

// Four standard normals for float4 group `idx4` of stream `stream_id`.
template <int ROUNDS = kPhiloxRounds>
__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint64_t stream_id, uint64_t idx4, uint32_t domain) {
  return box_muller4(philox_bits4<ROUNDS>(seed, stream_id, idx4, domain));
}

// softplus(rho) = torch.nn.functional.softplus (beta = 1, threshold = 20; util.py:183) and
// sigmoid(rho) = d softplus / d rho, from ONE exponential:
//   e = exp(-|x|) in (0, 1], u = 1 + e, softplus = max(x, 0) + log1p(e), sigmoid = x >= 0 ? 1/u : e/u,
// with log1p(e) = log(u) - ((u - 1) - e) / u (the rounding of 1 + e corrected to first order).
// Measured on gfx950 against fp64 over [-30, 25] (tools/acc.hip): softplus <= 2.5e-7, sigmoid
// <= 2.1e-7 relative -- the same class as log1pf(expf(x)) (1.3e-7) at about a third of the
// instructions (v_log_f32 / v_rcp_f32 are 1-ulp hardware ops; expf is the accurate ocml one).
// For x > 20 the sum max(x, 0) + log1p(e) rounds to x exactly, as torch's threshold branch returns.
struct SoftplusSigmoid {
  float sp, sg;
};
__device__ __forceinline__ SoftplusSigmoid softplus_sigmoid(float x) {
  const float e = expf(-fabsf(x));
  const float u = 1.0f + e;
  const float r = __builtin_amdgcn_rcpf(u);
  const float l = __logf(u) - ((u - 1.0f) - e) * r;
  SoftplusSigmoid o;
  o.sp = fmaxf(x, 0.0f) + l;
  o.sg = x >= 0.0f ? r : e * r;
  return o;
}
__device__ __forceinline__ float softplus(float x) { return softplus_sigmoid(x).sp; }
__device__ __forceinline__ float sigmoidf(float x) { return softplus_sigmoid(x).sg; }

}  // namespace bde
