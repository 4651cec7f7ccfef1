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
    uint64_t stream0, float* __restrict__ out, int64_t ld_out, int S, int64_t D, RowPiecesRt L, RowPiecesRt Lo) {
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
    // a tile is 128 consecutive parameters and a piece a multiple of that, so a tile lies inside ONE piece
    const int64_t so = piece_off_rt(4 * g4, L), oo = piece_off_rt(4 * g4, Lo);
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
      const int64_t so = piece_off_rt(e, L);
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
      out[static_cast<int64_t>(s) * ld_out + piece_off_rt(e, Lo)] =
          (m + acc) + __builtin_sqrtf(0.5f * (fmaxf(sq[so] - m * m, 0.0f) + 1e-6f)) * z;
    }
  }
}

}  // namespace bde

using namespace bde;

extern "C" int bde_swag_sample_batched(const float* mean, const float* sq, const float* dev, int K, int64_t ld, int head,
                                       const float* eps_w, const float* eps_d, int64_t ld_eps, uint64_t seed,
                                       uint64_t stream_id0, float* out, int64_t ld_out, int S, int64_t D, int log2_piece,
                                       int64_t piece_stride, int log2_piece_out, int64_t piece_stride_out, void* stream) {
  if (!mean || !sq || !dev || !out || D <= 0 || K < 1 || K > BDE_MAX_RANK || S < 1 || S > BDE_MAX_BATCH)
    return BDE_ERR_INVALID;
  if (!pieces_ok(log2_piece, piece_stride) || !pieces_ok(log2_piece_out, piece_stride_out)) return BDE_ERR_INVALID;
  if (head < 0 || head >= K || (ld & 3) || (ld_out & 3) || (log2_piece == 0 ? ld < D : ld < (int64_t{1} << log2_piece)) ||
      (log2_piece_out == 0 ? ld_out < D : ld_out < (int64_t{1} << log2_piece_out)) ||
      (eps_d && (ld_eps < D || (ld_eps & 3))))
    return BDE_ERR_INVALID;
  const RowPiecesRt L{log2_piece, piece_stride}, Lo{log2_piece_out, piece_stride_out};
  if (!aligned16(mean) || !aligned16(sq) || !aligned16(dev) || !aligned16(out) || (eps_d && !aligned16(eps_d)))
    return BDE_ERR_INVALID;
  const int64_t n_tiles = ((D >> 2) + 31) / 32;
  // 12 workgroups of 4 waves per CU = the 3 waves per SIMD that 168 VGPRs allow: the grid-stride loop keeps every one of
  // them busy (4 per CU, one wave per SIMD, left the load -> MFMA -> Philox epilogue chain of a tile un-overlapped:
  // 1.02 -> 0.98 ms with in-kernel noise, 1.70 -> 1.43 ms with supplied noise; tools/kexp5.hip batched)
  const int grid = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>((n_tiles + 3) / 4, kCUs * 12)));
  const size_t lds = sizeof(float) * static_cast<size_t>(K + (K & 1)) * 32;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (eps_d)
    hipLaunchKernelGGL(swag_sample_batched_kernel<false>, dim3(grid), dim3(kBlock), lds, s, mean, sq, dev, K, ld, head,
                       eps_w, eps_d, ld_eps, seed, stream_id0, out, ld_out, S, D, L, Lo);
  else
    hipLaunchKernelGGL(swag_sample_batched_kernel<true>, dim3(grid), dim3(kBlock), lds, s, mean, sq, dev, K, ld, head,
                       eps_w, eps_d, ld_eps, seed, stream_id0, out, ld_out, S, D, L, Lo);
  return to_err(hipGetLastError());
}

// bde_init(): load this translation unit's code object on the current device now (HIP otherwise uploads it at the
// first launch of one of its kernels).  Internal to the library (not exported).
extern "C" __attribute__((visibility("hidden"))) int bde_internal_load_swag_batched(void) {
  hipFuncAttributes attr;
  return bde::to_err(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&bde::swag_sample_batched_kernel<true>)));
}
