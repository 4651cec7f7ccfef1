// SWAG: running-moment collection and low-rank + diagonal Gaussian sampling.
//
// Reference: src/algos/swag.py:91-114 (+ torch LowRankMultivariateNormal.rsample).
// The reference keeps mean/sq/deviations on the CPU, copies the flattened
// weights GPU->CPU every update, physically rolls a [D, K] matrix, and ships
// everything back to the device to sample.  Here the statistics stay in HBM,
// the deviation matrix is a [K, ld] ring (one coalesced row per iterate; rows a leading dimension apart -- round 3's
// interleaved 16 KB pieces did not win in any bench.py record and are gone) and
// both operations are single streaming passes:
//   update : 24 B / parameter   (theta r, mean rw, sq rw, one ring row w)
//   sample : 4 (K + 3) B / parameter with in-kernel Philox noise
//            (+4 B when eps_d is supplied for parity).
// Both are HBM-bound (<= 0.5 flop/B); no LDS tiling is useful beyond staging
// the K noise weights once per workgroup.
#include "bde_common.hpp"

namespace bde {

// ---------------------------------------------------------------- update --
// Bit-exact with the reference's CPU fp32 arithmetic: separately rounded
// multiply, add and IEEE divide (this file is built with -ffp-contract=off).
__global__ __launch_bounds__(kBlock) void swag_update_kernel(const float* __restrict__ theta,
                                                            float* __restrict__ mean, float* __restrict__ sq,
                                                            float* __restrict__ dev_row, float n, float np1,
                                                            int64_t D) {
  const int64_t n4 = D >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int64_t o = 4 * i;
    const f32x4 t = ld4_nt(theta + 4 * i);
    f32x4 m = ld4(mean + o);
    f32x4 s = ld4(sq + o);
    m = (n * m + t) / np1;            // swag.py:101
    s = (n * s + t * t) / np1;        // swag.py:102
    st4(mean + o, m);
    st4(sq + o, s);
    st4_nt(dev_row + o, t - m);       // swag.py:104 (deviation from the UPDATED mean)
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < D) {
      const int64_t o = e;
      const float t = theta[e];
      const float m = (n * mean[o] + t) / np1;
      const float s = (n * sq[o] + t * t) / np1;
      mean[o] = m;
      sq[o] = s;
      dev_row[o] = t - m;
    }
  }
}

// ---------------------------------------------------------------- sample --
template <int ROUNDS>
__device__ __forceinline__ float lowrank_noise(const float* __restrict__ eps_w, uint64_t seed, uint64_t stream_id, int c) {
  if (eps_w) return eps_w[c];
  const f32x4 z = philox_normal4<ROUNDS>(seed, stream_id, static_cast<uint64_t>(c >> 2), kDomainLowRank);
  return z[c & 3];
}

__device__ __forceinline__ f32x4 diag_std(f32x4 m, f32x4 s) {
  // swag.py:112: 0.5 * (relu(sq - mean^2) + 1e-6), then rsample's sqrt
  f32x4 v = s - m * m;
  f32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = __builtin_sqrtf(0.5f * (fmaxf(v[j], 0.0f) + 1e-6f));
  return r;
}

template <bool RNG>
__global__ __launch_bounds__(kBlock) void swag_sample_kernel(const float* __restrict__ mean,
                                                            const float* __restrict__ sq,
                                                            const float* __restrict__ dev, int K, int64_t ld,
                                                            int head, const float* __restrict__ eps_w,
                                                            const float* __restrict__ eps_d, uint64_t seed,
                                                            uint64_t stream_id, float* __restrict__ out,
                                                            int64_t D) {
  __shared__ float w[BDE_MAX_RANK];   // noise weight of each PHYSICAL ring row
  const float denom = __builtin_sqrtf(2.0f * static_cast<float>(K - 1));   // swag.py:113
  for (int r = threadIdx.x; r < K; r += blockDim.x) {
    int c = r - head;
    if (c < 0) c += K;
    w[r] = lowrank_noise<kSwagPhiloxRounds>(eps_w, seed, stream_id, c) / denom;
  }
  __syncthreads();

  const int64_t n4 = D >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  // Two float4 columns (one grid-stride apart) per thread and 5 ring rows per batch: 10 independent 16-byte
  // loads in flight per lane.  Measured on MI355X (tools/kexp5.hip, profiles/r02_probes_and_variants_before.txt): 6.2-6.3 TB/s against
  // 6.0 for one column x 10 rows and 5.4-5.9 for contiguous per-workgroup chunks; a kernel that only reads the
  // same K + 2 rows and writes one reaches 5.75.
  constexpr int U = 2;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride * U) {
    f32x4 acc[U];
    int64_t off[U];                                        // where the column sits inside a row: once per column
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int64_t c = i + u * stride;
      off[u] = 4 * (c < n4 ? c : i);
    }
#pragma unroll 5
    for (int r = 0; r < K; ++r) {
      const float wr = w[r];
      const float* row = dev + static_cast<int64_t>(r) * ld;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (i + u * stride < n4) {
          const f32x4 d = ld4_nt(row + off[u]);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[u][j] = __builtin_fmaf(d[j], wr, acc[u][j]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t c = i + u * stride;
      if (c < n4) {
        const f32x4 m = ld4_nt(mean + off[u]);
        const f32x4 s = ld4_nt(sq + off[u]);
        const f32x4 z = RNG ? philox_normal4<kSwagPhiloxRounds>(seed, stream_id, static_cast<uint64_t>(c), kDomainDiag)
                            : ld4_nt(eps_d + 4 * c);
        BDE_OUT_ST(out + 4 * c, (m + acc[u]) + diag_std(m, s) * z);
      }
    }
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < D) {
      float acc = 0.f;
      const int64_t o = e;
      for (int r = 0; r < K; ++r) acc = __builtin_fmaf(dev[static_cast<int64_t>(r) * ld + o], w[r], acc);
      const float m = mean[o], s = sq[o];
      float z;
      if (RNG) {
        const f32x4 zz = philox_normal4<kSwagPhiloxRounds>(seed, stream_id, static_cast<uint64_t>(n4), kDomainDiag);
        z = zz[threadIdx.x & 3];
      } else {
        z = eps_d[e];
      }
      out[e] = (m + acc) + __builtin_sqrtf(0.5f * (fmaxf(s - m * m, 0.0f) + 1e-6f)) * z;
    }
  }
}

// The Philox normals written out (tests; callers that want the noise).
template <int ROUNDS>
__global__ __launch_bounds__(kBlock) void philox_normal_kernel(uint64_t seed, uint64_t stream_id, float* __restrict__ eps_w,
                                                              int K, float* __restrict__ eps_d, int64_t D) {
  const int64_t gid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  if (eps_w) {
    for (int64_t c = gid; c < K; c += stride) eps_w[c] = lowrank_noise<ROUNDS>(nullptr, seed, stream_id, static_cast<int>(c));
  }
  if (eps_d) {
    const int64_t n4 = (D + 3) >> 2;
    for (int64_t i = gid; i < n4; i += stride) {
      const f32x4 z = philox_normal4<ROUNDS>(seed, stream_id, static_cast<uint64_t>(i), kDomainDiag);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * i + j < D) eps_d[4 * i + j] = z[j];
    }
  }
}

template <int ROUNDS>
__global__ __launch_bounds__(kBlock) void philox_bits_kernel(uint64_t seed, uint64_t stream_id, uint32_t domain,
                                                            uint64_t idx0, uint32_t* __restrict__ out, int64_t n_groups) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; g < n_groups; g += stride) {
    const uint4 r = philox_bits4<ROUNDS>(seed, stream_id, idx0 + static_cast<uint64_t>(g), domain);
    out[4 * g + 0] = r.x;
    out[4 * g + 1] = r.y;
    out[4 * g + 2] = r.z;
    out[4 * g + 3] = r.w;
  }
}

}  // namespace bde

using namespace bde;

extern "C" int bde_swag_update(const float* theta, float* mean, float* sq, float* dev_row, int64_t n, int64_t D, void* stream) {
  if (!theta || !mean || !sq || !dev_row || D <= 0 || n < 1) return BDE_ERR_INVALID;
  if (!aligned16(theta) || !aligned16(mean) || !aligned16(sq) || !aligned16(dev_row)) return BDE_ERR_INVALID;
  const int grid = stream_grid((D + 3) / 4);
  hipLaunchKernelGGL(swag_update_kernel, dim3(grid), dim3(kBlock), 0, static_cast<hipStream_t>(stream), theta, mean, sq, dev_row,
                     static_cast<float>(n), static_cast<float>(n + 1), D);
  return to_err(hipGetLastError());
}

extern "C" int bde_swag_sample(const float* mean, const float* sq, const float* dev, int K, int64_t ld, int head,
                               const float* eps_w, const float* eps_d, uint64_t seed, uint64_t stream_id, float* out,
                               int64_t D, void* stream) {
  if (!mean || !sq || !dev || !out || D <= 0 || K < 1 || K > BDE_MAX_RANK) return BDE_ERR_INVALID;
  if (head < 0 || head >= K || (ld & 3) || ld < D) return BDE_ERR_INVALID;
  if (!aligned16(mean) || !aligned16(sq) || !aligned16(dev) || !aligned16(out) || (eps_d && !aligned16(eps_d)))
    return BDE_ERR_INVALID;
  const int grid = stream_grid((D + 3) / 4);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (eps_d)
    hipLaunchKernelGGL(swag_sample_kernel<false>, dim3(grid), dim3(kBlock), 0, s, mean, sq, dev, K, ld, head, eps_w, eps_d, seed,
                       stream_id, out, D);
  else
    hipLaunchKernelGGL(swag_sample_kernel<true>, dim3(grid), dim3(kBlock), 0, s, mean, sq, dev, K, ld, head, eps_w, eps_d, seed,
                       stream_id, out, D);
  return to_err(hipGetLastError());
}

extern "C" int bde_swag_philox_rounds(void) { return kSwagPhiloxRounds; }

extern "C" int bde_philox_normal(uint64_t seed, uint64_t stream_id, float* eps_w, int K, float* eps_d, int64_t D,
                                 int rounds, void* stream) {
  if ((!eps_w && !eps_d) || (eps_w && K < 1) || (eps_d && D < 1) || (rounds != kPhiloxRounds && rounds != kSwagPhiloxRounds))
    return BDE_ERR_INVALID;
  const int grid = stream_grid(eps_d ? (D + 3) / 4 : K);
  if (rounds == kPhiloxRounds)
    hipLaunchKernelGGL(philox_normal_kernel<kPhiloxRounds>, dim3(grid), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       seed, stream_id, eps_w, K, eps_d, D);
  else
    hipLaunchKernelGGL(philox_normal_kernel<kSwagPhiloxRounds>, dim3(grid), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), seed, stream_id, eps_w, K, eps_d, D);
  return to_err(hipGetLastError());
}

extern "C" int bde_philox_bits(uint64_t seed, uint64_t stream_id, uint32_t domain, uint64_t idx0, uint32_t* out,
                               int64_t n_groups, int rounds, void* stream) {
  if (!out || n_groups < 1 || (rounds != kPhiloxRounds && rounds != kSwagPhiloxRounds)) return BDE_ERR_INVALID;
  if (rounds == kPhiloxRounds)
    hipLaunchKernelGGL(philox_bits_kernel<kPhiloxRounds>, dim3(stream_grid(n_groups)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), seed, stream_id, domain, idx0, out, n_groups);
  else
    hipLaunchKernelGGL(philox_bits_kernel<kSwagPhiloxRounds>, dim3(stream_grid(n_groups)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), seed, stream_id, domain, idx0, out, n_groups);
  return to_err(hipGetLastError());
}

// bde_init(): load this translation unit's code object on the current device now (HIP otherwise uploads it at the
// first launch of one of its kernels).  Internal to the library (not exported).
extern "C" __attribute__((visibility("hidden"))) int bde_internal_load_swag(void) {
  hipFuncAttributes attr;
  return bde::to_err(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&bde::swag_update_kernel)));
}
