// SWAG: S posterior samples in ONE pass over the statistics.
//
// Reference: DeepEnsemble.predict (src/algos/ensemble.py:37-42) calls
// SwagOptimizer.sample_parameters() S times; each call re-reads the whole
// [D, K] deviation matrix (swag.py:57).  Batched, the statistics are read once:
//   bytes = 4 * D * (K + 2 + S)   instead of   S * 4 * D * (K + 3),
// and the deviation-matrix x noise product  C[S, D] = Wn[S, K] . Dev[K, D]
// (2*K*S flop per parameter: 5.8 flop/B at K = 20, S = 30) runs on the f32
// MFMA (v_mfma_f32_32x32x2_f32) so the VALU stays free for the Philox/
// Box-Muller noise of the diagonal term.  Still HBM-bound.
//
// Tile: one wave owns 128 consecutive parameters x 32 samples.  Each lane
// loads a float4 of one ring row (lanes 0-31 row r, lanes 32-63 row r+1 --
// 512 B contiguous per row); component c of the float4 feeds accumulator tile
// c, so that a lane ends up with 4 CONSECUTIVE parameters of one sample in
// (acc0[reg] .. acc3[reg]) and the epilogue stores float4s, 512 B contiguous
// per sample row.
#include "bde_common.hpp"

namespace bde {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float lowrank_noise_b(const float* __restrict__ eps_w, uint64_t seed, uint64_t stream_id, int c) {
  if (eps_w) return eps_w[c];
  const f32x4 z = philox_normal4<kSwagPhiloxRounds>(seed, stream_id, static_cast<uint64_t>(c >> 2), kDomainLowRank);
  return z[c & 3];
}

constexpr int kBatchCH = 10;   // k-steps (2 ring rows each) whose loads are issued back to back

template <bool RNG>
__global__ __launch_bounds__(kBlock, 3) void swag_sample_batched_kernel(
    const float* __restrict__ mean, const float* __restrict__ sq, const float* __restrict__ dev, int K, int64_t ld,
    int head, const float* __restrict__ eps_w, const float* __restrict__ eps_d, int64_t ld_eps, uint64_t seed,
    uint64_t stream0, float* __restrict__ out, int64_t ld_out, int S, int64_t D) {
  extern __shared__ __attribute__((aligned(16))) float w[];   // [K + (K & 1)][32]: weight of ring row r for sample s
  const int kpad = K + (K & 1);
  const int ksteps = kpad >> 1;
  const float denom = __builtin_sqrtf(2.0f * static_cast<float>(K - 1));   // swag.py:113
  for (int idx = threadIdx.x; idx < kpad * 32; idx += blockDim.x) {
    const int r = idx >> 5, s = idx & 31;
    float v = 0.f;
    if (r < K && s < S) {
      int c = r - head;
      if (c < 0) c += K;
      v = lowrank_noise_b(eps_w ? eps_w + static_cast<int64_t>(s) * K : nullptr, seed, stream0 + s, c) / denom;
    }
    w[idx] = v;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, half = lane >> 5;
  const int64_t n4 = D >> 2;                                  // full float4 groups
  const int64_t n_tiles = (n4 + 31) / 32;                     // 32 float4 = 128 parameters per tile
  const int64_t waves_total = static_cast<int64_t>(gridDim.x) * (kBlock / 64);

  for (int64_t t = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wave; t < n_tiles; t += waves_total) {
    const int64_t g4 = t * 32 + j;                            // this lane's float4 group
    const bool ok = g4 < n4;
    const int64_t so = 4 * g4, oo = 4 * g4;
    const float* col = dev + so;
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    // The whole [K, 128] slab of the ring is requested before the first MFMA waits on it: a
    // load -> MFMA -> load chain exposed one HBM latency per k-step (2.7x off the roofline).
    for (int c0 = 0; c0 < ksteps; c0 += kBatchCH) {
      f32x4 b[kBatchCH];
#pragma unroll
      for (int u = 0; u < kBatchCH; ++u) {
        const int r = 2 * (c0 + u) + half;
        b[u] = (ok && r < K) ? ld4_nt(col + static_cast<int64_t>(r) * ld) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < kBatchCH; ++u) {
        if (c0 + u < ksteps) {                                // wave-uniform
          const float a = w[(2 * (c0 + u) + half) * 32 + j];
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[u][0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[u][1], acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[u][2], acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[u][3], acc3, 0, 0, 0);
        }
      }
    }
    if (ok) {
      const f32x4 m = ld4(mean + so);
      const f32x4 v = ld4(sq + so) - m * m;
      f32x4 sd;
#pragma unroll
      for (int c = 0; c < 4; ++c) sd[c] = __builtin_sqrtf(0.5f * (fmaxf(v[c], 0.0f) + 1e-6f));   // swag.py:112
      // rolled on purpose: unrolled, the 16 Philox chains cost > 240 VGPRs (2 waves/SIMD, spills at 3)
#pragma unroll 1
      for (int reg = 0; reg < 16; ++reg) {
        const int s = (reg & 3) + 8 * (reg >> 2) + 4 * half;   // C/D row of the 32x32 tile
        if (s < S) {
#ifdef BDE_BATCHED_NO_RNG   // tools/swag_piece_sweep.py only: the memory + MFMA side of the kernel without the noise epilogue
          const f32x4 z = {1.f, 1.f, 1.f, 1.f};
#else
          const f32x4 z = RNG ? philox_normal4<kSwagPhiloxRounds>(seed, stream0 + s, static_cast<uint64_t>(g4), kDomainDiag)
                              : ld4_nt(eps_d + static_cast<int64_t>(s) * ld_eps + 4 * g4);
#endif
          const f32x4 lr = {acc0[reg], acc1[reg], acc2[reg], acc3[reg]};
          st4_nt(out + static_cast<int64_t>(s) * ld_out + oo, (m + lr) + sd * z);
        }
      }
    }
  }

  // the D % 4 tail parameters: plain dot products (block 0 only)
  if (blockIdx.x == 0) {
    const int rem = static_cast<int>(D - (n4 << 2));
    for (int idx = threadIdx.x; idx < rem * S; idx += blockDim.x) {
      const int s = idx / rem, k = idx % rem;
      const int64_t e = (n4 << 2) + k;
      const int64_t so = e;
      float acc = 0.f;
      for (int r = 0; r < K; ++r) acc = __builtin_fmaf(dev[static_cast<int64_t>(r) * ld + so], w[r * 32 + s], acc);
      const float m = mean[so];
      float z;
      if (RNG) {
        const f32x4 zz = philox_normal4<kSwagPhiloxRounds>(seed, stream0 + s, static_cast<uint64_t>(n4), kDomainDiag);
        z = zz[k];
      } else {
        z = eps_d[static_cast<int64_t>(s) * ld_eps + e];
      }
      out[static_cast<int64_t>(s) * ld_out + e] =
          (m + acc) + __builtin_sqrtf(0.5f * (fmaxf(sq[so] - m * m, 0.0f) + 1e-6f)) * z;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Round 4: the same tile, software-pipelined through the LDS-DMA.
//
// The register kernel above runs load -> MFMA -> epilogue per tile and wave; a wave in its epilogue (16 x (Philox +
// Box-Muller + 512-byte store), ~80 VALU instructions each) has nothing in flight, so with 3 waves per SIMD only about a
// third of them keep memory busy (~40 KB in flight per CU against the ~64 KB that 8 TB/s x ~2 us of loaded latency
// need).  Holding the NEXT tile's ring rows in registers during the epilogue would cost 40 more VGPRs on a kernel that sits
// at its 168-register budget.  global_load_lds_dwordx4 has no register destination: each wave owns an 11 KB LDS slab
// ([20 ring rows + mean + sq][128 floats], lane-linear = exactly the order its lanes read the MFMA B operand back with
// ds_read_b128), and the slab of tile t+1 is requested as soon as tile t's operands sit in registers -- BEFORE its 40
// MFMAs and its whole epilogue (K <= 20: one slab per tile; more ring rows run the register kernel above).  Every wave then has 11 KB in
// flight all the time (132 KB per CU).  No barrier: a slab is private to its wave, the only waits are the wave's own
// s_waitcnt vmcnt (DMA landed) and lgkmcnt (operands read before the slab is refilled).  hipcc does not order a ds_read
// behind an LDS-DMA in flight (checked in the ISA), so those waits are explicit.
//
// The epilogue runs two independent Philox chains per trip (the 64-bit multiplies of one chain fill the latency of the
// other's) and reads mean / second moment from the slab as well, so there is no ordinary global load inside the loop
// (hipcc would drain the DMA queue with vmcnt(0) at its first use).
constexpr int kSlabRows = 2 * kBatchCH + 2;            // 20 ring rows + mean + sq
constexpr int kSlabFloats = kSlabRows * 128;

// 64 lanes x 16 bytes: lane l's source is base + off_bytes[l], its destination lds_uniform_base + 16 l (the LDS
// address of an LDS-DMA is wave-uniform + lane * size).  Uniform 64-bit base + 32-bit lane offset = the saddr form of
// the instruction: ONE VGPR of addressing for all rows of a tile.
__device__ __forceinline__ void dma16(const float* __restrict__ base_uniform, uint32_t off_bytes, float* lds_uniform_base) {
  const char* src = reinterpret_cast<const char*>(base_uniform) + off_bytes;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),
                                   (__attribute__((address_space(3))) void*)(lds_uniform_base), 16, 0, 2 /* nt */);
}

template <bool RNG, int ROUNDS>
__global__ __launch_bounds__(kBlock, 3) void swag_sample_batched_dma_kernel(
    const float* __restrict__ mean, const float* __restrict__ sq, const float* __restrict__ dev, int K, int64_t ld,
    int head, const float* __restrict__ eps_w, const float* __restrict__ eps_d, int64_t ld_eps, uint64_t seed,
    uint64_t stream0, float* __restrict__ out, int64_t ld_out, int S, int64_t D) {
  extern __shared__ __attribute__((aligned(16))) float w[];   // [kpad][32] weights, then one slab per wave
  const int kpad = K + (K & 1);
  const int ksteps = kpad >> 1;                               // <= kBatchCH: one slab holds all ring rows of a tile
  const float denom = __builtin_sqrtf(2.0f * static_cast<float>(K - 1));   // swag.py:113
  for (int idx = threadIdx.x; idx < kpad * 32; idx += blockDim.x) {
    const int r = idx >> 5, s = idx & 31;
    float v = 0.f;
    if (r < K && s < S) {
      int c = r - head;
      if (c < 0) c += K;
      v = lowrank_noise_b(eps_w ? eps_w + static_cast<int64_t>(s) * K : nullptr, seed, stream0 + s, c) / denom;
    }
    w[idx] = v;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, half = lane >> 5;
  float* slab = w + kpad * 32 + wave * kSlabFloats;          // wave-uniform
  const float* my = slab + 4 * lane;                          // where this lane's float4 of a row pair lands
  const int64_t n4 = D >> 2;                                  // full float4 groups
  const int64_t n_tiles = (n4 + 31) / 32;                     // 32 float4 = 128 parameters per tile
  const int64_t waves_total = static_cast<int64_t>(gridDim.x) * (kBlock / 64);
  const uint32_t row_bytes = static_cast<uint32_t>(ld) * 4u;  // ld < 2^30 floats (checked by the launcher)

  // Request the slab of a tile: ring rows 2 u + half, u < ksteps, then mean | sq.  All
  // addressing is a wave-uniform base (tile, row pair) + ONE per-lane byte offset: (half, float4 j) -- lanes past the
  // end of the vector (last tile) read the last valid float4 instead; their columns are never stored.  The odd row K
  // of an odd K re-reads row K - 1 (its weight is zero).
  auto request = [&](int64_t tile) {
    constexpr int c0 = 0;
    const int64_t e0 = tile * 128;                                            // uniform
    const int jlast = static_cast<int>(min<int64_t>(31, n4 - 1 - tile * 32));  // uniform
    const uint32_t joff = 16u * static_cast<uint32_t>(min(j, jlast));
    const float* base = dev + e0;                                             // uniform
#pragma unroll
    for (int u = 0; u < kBatchCH; ++u) {
      if (c0 + u < ksteps) {                                                  // wave-uniform
        const int r0 = 2 * (c0 + u);
        const uint32_t lane_off = (r0 + 1 < K) ? joff + (half ? row_bytes : 0u) : joff;
        dma16(base + static_cast<int64_t>(r0) * ld, lane_off, slab + u * 256);
      }
    }
    dma16(half ? sq + e0 : mean + e0, joff, slab + kBatchCH * 256);
  };

  int64_t t = static_cast<int64_t>(blockIdx.x) * (kBlock / 64) + wave;
  if (t < n_tiles) request(t);
  for (; t < n_tiles; t += waves_total) {
    const int64_t g4 = t * 32 + j;                            // this lane's float4 group
    const bool ok = g4 < n4;
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    f32x4 m = {}, q = {};
    {
      constexpr int c0 = 0;
      f32x4 b[kBatchCH];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this chunk's slab has landed
#pragma unroll
      for (int u = 0; u < kBatchCH; ++u) b[u] = *reinterpret_cast<const f32x4*>(my + u * 256);
      m = *reinterpret_cast<const f32x4*>(slab + kBatchCH * 256 + 4 * j);
      q = *reinterpret_cast<const f32x4*>(slab + kBatchCH * 256 + 128 + 4 * j);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // operands in registers: the slab may be refilled
      if (t + waves_total < n_tiles) request(t + waves_total);
#pragma unroll
      for (int u = 0; u < kBatchCH; ++u) {
        if (c0 + u < ksteps) {                                // wave-uniform
          const float a = w[(2 * (c0 + u) + half) * 32 + j];
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[u][0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[u][1], acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[u][2], acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[u][3], acc3, 0, 0, 0);
        }
      }
    }
    if (ok) {
      const int64_t oo = 4 * g4;
      const f32x4 v = q - m * m;
      f32x4 sd;
#pragma unroll
      for (int c = 0; c < 4; ++c) sd[c] = __builtin_sqrtf(0.5f * (fmaxf(v[c], 0.0f) + 1e-6f));   // swag.py:112
      // two samples (= two independent Philox chains) per trip; rolled: all 16 chains at once cost > 240 VGPRs
#pragma unroll 1
      for (int reg = 0; reg < 16; reg += 2) {
        const int s0 = (reg & 3) + 8 * (reg >> 2) + 4 * half;  // C/D rows of the 32x32 tile: reg and reg + 1
        f32x4 z0, z1;
#ifdef BDE_BATCHED_NO_RNG
        z0 = z1 = f32x4{1.f, 1.f, 1.f, 1.f};
#else
        if (RNG) {
          z0 = philox_normal4<ROUNDS>(seed, stream0 + s0, static_cast<uint64_t>(g4), kDomainDiag);
          z1 = philox_normal4<ROUNDS>(seed, stream0 + s0 + 1, static_cast<uint64_t>(g4), kDomainDiag);
        } else {
          z0 = s0 < S ? ld4_nt(eps_d + static_cast<int64_t>(s0) * ld_eps + 4 * g4) : f32x4{};
          z1 = s0 + 1 < S ? ld4_nt(eps_d + static_cast<int64_t>(s0 + 1) * ld_eps + 4 * g4) : f32x4{};
        }
#endif
        // both chains complete BEFORE the (lane-masked) stores: otherwise the compiler sinks each chain into the branch
        // of its own store and they run one after the other
        asm volatile("" : "+v"(z0), "+v"(z1));
        const f32x4 lr0 = {acc0[reg], acc1[reg], acc2[reg], acc3[reg]};
        const f32x4 lr1 = {acc0[reg + 1], acc1[reg + 1], acc2[reg + 1], acc3[reg + 1]};
        if (s0 < S) st4_nt(out + static_cast<int64_t>(s0) * ld_out + oo, (m + lr0) + sd * z0);
        if (s0 + 1 < S) st4_nt(out + static_cast<int64_t>(s0 + 1) * ld_out + oo, (m + lr1) + sd * z1);
      }
    }
  }

  // the D % 4 tail parameters: plain dot products (block 0 only)
  if (blockIdx.x == 0) {
    const int rem = static_cast<int>(D - (n4 << 2));
    for (int idx = threadIdx.x; idx < rem * S; idx += blockDim.x) {
      const int s = idx / rem, k = idx % rem;
      const int64_t e = (n4 << 2) + k;
      const int64_t so = e;
      float acc = 0.f;
      for (int r = 0; r < K; ++r) acc = __builtin_fmaf(dev[static_cast<int64_t>(r) * ld + so], w[r * 32 + s], acc);
      const float mm = mean[so];
      float z;
      if (RNG) {
        const f32x4 zz = philox_normal4<ROUNDS>(seed, stream0 + s, static_cast<uint64_t>(n4), kDomainDiag);
        z = zz[k];
      } else {
        z = eps_d[static_cast<int64_t>(s) * ld_eps + e];
      }
      out[static_cast<int64_t>(s) * ld_out + e] =
          (mm + acc) + __builtin_sqrtf(0.5f * (fmaxf(sq[so] - mm * mm, 0.0f) + 1e-6f)) * z;
    }
  }
}

}  // namespace bde

using namespace bde;

extern "C" int bde_swag_sample_batched(const float* mean, const float* sq, const float* dev, int K, int64_t ld, int head,
                                       const float* eps_w, const float* eps_d, int64_t ld_eps, uint64_t seed,
                                       uint64_t stream_id0, float* out, int64_t ld_out, int S, int64_t D, void* stream) {
  if (!mean || !sq || !dev || !out || D <= 0 || K < 1 || K > BDE_MAX_RANK || S < 1 || S > BDE_MAX_BATCH)
    return BDE_ERR_INVALID;
  if (head < 0 || head >= K || (ld & 3) || (ld_out & 3) || ld < D || ld_out < D || (eps_d && (ld_eps < D || (ld_eps & 3))))
    return BDE_ERR_INVALID;
  if (!aligned16(mean) || !aligned16(sq) || !aligned16(dev) || !aligned16(out) || (eps_d && !aligned16(eps_d)))
    return BDE_ERR_INVALID;
  const int64_t n_tiles = ((D >> 2) + 31) / 32;
  // 12 workgroups of 4 waves per CU = 3 waves per SIMD (the register budget of both kernels): the grid-stride loop keeps
  // every one of them busy
  const int grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>((n_tiles + 3) / 4, kCUs * 12)));
  const size_t lds = sizeof(float) * static_cast<size_t>(K + (K & 1)) * 32;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // K <= 20 (one 20-row slab per tile; BASELINE's K): the LDS-DMA pipelined kernel.  More ring rows: the register kernel
  // (a multi-slab DMA variant spilled at the 168-register budget and lost 7 % to it, profiles/r04_swag_batched_ab.txt).
  // mean + 127 floats past a partial last tile stay inside the row (ld >= D; the DMA clamps its lanes to the last float4).
  if ((K + (K & 1)) / 2 <= kBatchCH && (D >> 2) >= 1 && ld < (int64_t{1} << 30)) {
    const size_t lds_dma = lds + sizeof(float) * (kBlock / 64) * kSlabFloats;
    if (eps_d)
      hipLaunchKernelGGL((swag_sample_batched_dma_kernel<false, kSwagPhiloxRounds>), dim3(grid), dim3(kBlock), lds_dma, s, mean,
                         sq, dev, K, ld, head, eps_w, eps_d, ld_eps, seed, stream_id0, out, ld_out, S, D);
    else
      hipLaunchKernelGGL((swag_sample_batched_dma_kernel<true, kSwagPhiloxRounds>), dim3(grid), dim3(kBlock), lds_dma, s, mean,
                         sq, dev, K, ld, head, eps_w, eps_d, ld_eps, seed, stream_id0, out, ld_out, S, D);
    return to_err(hipGetLastError());
  }
  if (eps_d)
    hipLaunchKernelGGL(swag_sample_batched_kernel<false>, dim3(grid), dim3(kBlock), lds, s, mean, sq, dev, K, ld, head,
                       eps_w, eps_d, ld_eps, seed, stream_id0, out, ld_out, S, D);
  else
    hipLaunchKernelGGL(swag_sample_batched_kernel<true>, dim3(grid), dim3(kBlock), lds, s, mean, sq, dev, K, ld, head,
                       eps_w, eps_d, ld_eps, seed, stream_id0, out, ld_out, S, D);
  return to_err(hipGetLastError());
}

// bde_init(): load this translation unit's code object on the current device now (HIP otherwise uploads it at the
// first launch of one of its kernels).  Internal to the library (not exported).
extern "C" __attribute__((visibility("hidden"))) int bde_internal_load_swag_batched(void) {
  hipFuncAttributes attr;
  return bde::to_err(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&bde::swag_sample_batched_kernel<true>)));
}
