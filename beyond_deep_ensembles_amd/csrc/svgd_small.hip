// SVGD posterior update for SMALL models in ONE launch (CIFAR ResNet-20 scale: D = 273,610).
//
// Reference: src/algos/svgd.py:14-32,86-89 -- the same arithmetic as svgd.hip.  At a few hundred
// thousand parameters the three-launch path (gram -> kstats -> combine) is not bandwidth-bound
// any more: 35 MB of traffic is ~6 us of HBM time, but two launch boundaries, a one-workgroup
// statistics kernel and the combine's cold scalar loads made it 27 us (round-1 profile).  Here
// the whole update is one persistent launch, one workgroup per CU:
//
//   phase 1  every workgroup owns <= 512 consecutive float4 columns of P: centred Gram partial
//            of its slice on the f32 MFMA (same tiles as svgd_gram_kernel<2>), published with
//            write-through (sc1) stores;
//   hand-off one agent-scope atomic add per workgroup on one of 8 SHARDED arrive counters (a single
//            counter serialises 256 adds at ~12 ns each and is hammered by 256 pollers on top); the
//            last arriver of a shard adds to a top counter, the last of those raises the 8 go flags;
//            one lane per workgroup polls its shard's flag with sc1 loads (MI355X_MICROARCH.md,
//            inter-workgroup visibility: sc1 stores + drained vmcnt + counter add on the producer,
//            sc1 poll + workgroup barrier + sc1 loads on the consumer) -- while it waits, the
//            workgroup's P and G columns for phase 2 are already in flight into registers (P from
//            the XCD's L2, where phase 1 just put it);
//   phase 2  EVERY workgroup reduces all partials in the same fixed order (fp64) and evaluates the
//            kernel statistics redundantly (a few hundred scalar operations), keeps the 2 M^2
//            coefficients in LDS, and combines its own columns.
//
// P is read once from HBM (4 M D), G once (4 M D), out written once (4 M D): 12 M D bytes, no
// second pass over P.  The counters are reset by the last workgroup to leave, so the caller only
// has to hand in a workspace that was zeroed once.  All workgroups must be co-resident (they are:
// <= 256 workgroups of 256 threads, <= 1 per CU worth of registers/LDS each on a 256-CU device).
#include "svgd_gram.hpp"

namespace bde {

// development aid (tools/kexp6.hip): per-workgroup phase timestamps, 100 MHz wall clock
#ifdef BDE_SMALL_TIMING
__device__ unsigned long long g_small_ts[256 * 16];
#define BDE_TS(k) if (threadIdx.x == 0) g_small_ts[blockIdx.x * 16 + (k)] = wall_clock64();
#else
#define BDE_TS(k)
#endif

constexpr int kSmallBlock = 256;
constexpr int kSmallMaxTilesPerWG = 16;            // 16 tiles x 32 float4 columns = 512 columns = 2 per thread
constexpr int kSmallTile4 = kGramU * 8;            // float4 columns per Gram tile (PACK = 2)
constexpr int kSmallMaxGrid = 256;

__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int M, bool HAS_G>
__global__ __launch_bounds__(kSmallBlock, 1) void svgd_step_small_kernel(const float* __restrict__ P, const float* G,
                                                                        float* out, int64_t D, int64_t ld,
                                                                        int tiles_per_wg, StatParams sp,
                                                                        float* __restrict__ ws,
                                                                        float* __restrict__ kstat) {
  constexpr int MP = 8, MP2 = 64;
  __shared__ float tile[kSmallBlock / 64][16][17];
  __shared__ double red[kSmallBlock];
  __shared__ double gmat[MP2];
  __shared__ __attribute__((aligned(16))) float cgT[MP2];
  __shared__ __attribute__((aligned(16))) float cpT[MP2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwg = gridDim.x;
  unsigned* words = reinterpret_cast<unsigned*>(ws);
  const int shard = blockIdx.x & (kWsShards - 1);
  unsigned* arrive = words + kWsArriveWord + 32 * shard;
  unsigned* top = words + kWsTopWord;
  unsigned* go = words + kWsGoWord + 32 * shard;
  unsigned* depart = words + kWsDepartWord;
  float* part = ws + kWsHeaderFloats;

  BDE_TS(0)
  // ---------------- phase 1: centred Gram partial of this workgroup's columns ----------------
  const int r16 = lane & 15, kq = lane >> 4;
  const int c4 = (r16 >> 3) * 4 + kq;
  const int prow = r16 & 7;
  const bool valid = prow < M;
  const float inv_m = 1.0f / static_cast<float>(M);
  const float* rowp = P + static_cast<int64_t>(valid ? prow : 0) * ld;
  const int64_t n4c = (D + 3) >> 2;                              // float4 columns incl. a partial last one
  const int64_t n_tiles = (n4c + kSmallTile4 - 1) / kSmallTile4;
  const int64_t t0 = static_cast<int64_t>(blockIdx.x) * tiles_per_wg;
  const int64_t t1 = (t0 + tiles_per_wg < n_tiles) ? t0 + tiles_per_wg : n_tiles;

  f32x4acc acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  {
    // all of this wave's tiles are requested before the first DPP/MFMA chain waits on them
    constexpr int TW = kSmallMaxTilesPerWG / (kSmallBlock / 64);   // tiles per wave, at most
    f32x4 v[TW][kGramU];
#pragma unroll
    for (int k = 0; k < TW; ++k) {
      const int64_t t = t0 + wave + 4 * k;
      if (t < t1) gram_load_tile<8, false>(v[k], rowp, valid, t, kSmallTile4, c4, n4c, D);
    }
#pragma unroll
    for (int k = 0; k < TW; ++k) {
      const int64_t t = t0 + wave + 4 * k;
      if (t < t1) {                                                // wave-uniform
#pragma unroll
        for (int u = 0; u < kGramU; ++u) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x = valid ? v[k][u][j] : 0.f;
            const float s = group_sum<2>(x);
            const float q = valid ? (x - s * inv_m) : 0.f;
            if ((j & 1) == 0)
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(q, q, acc0, 0, 0, 0);
            else
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q, q, acc1, 0, 0, 0);
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[wave][4 * kq + r][r16] = acc0[r] + acc1[r];
  __syncthreads();
  BDE_TS(1)
  if (tid < MP2) {                                                 // one wave, one store instruction per 128-B line
    const int pi = tid / MP, pj = tid % MP;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kSmallBlock / 64; ++w) s += tile[w][pi][pj] + tile[w][pi + 8][pj + 8];
    st_sc1(part + static_cast<int64_t>(blockIdx.x) * MP2 + tid, s);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the write-through stores have left this CU
  __syncthreads();
  BDE_TS(2)
  if (tid == 0) {
    const unsigned in_shard = static_cast<unsigned>((nwg - shard + kWsShards - 1) / kWsShards);
    const unsigned shards = static_cast<unsigned>(nwg < kWsShards ? nwg : kWsShards);
    if (__hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1u) {
      // last of this shard: its counter can go back to zero, and the shard reports to the top counter
      __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1u) {
        // last of all: every partial tile is published
        __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (unsigned sh = 0; sh < shards; ++sh)
          __hip_atomic_store(words + kWsGoWord + 32 * sh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  BDE_TS(3)

  // ---------------- phase 2 operands: requested now, consumed after the hand-off ----------------
  const int64_t n4 = D >> 2;                                       // full float4 columns
  const int64_t col0 = t0 * kSmallTile4;
  const int64_t cA = col0 + tid, cB = col0 + kSmallBlock + tid;
  const int64_t colEnd = (t1 * kSmallTile4 < n4) ? t1 * kSmallTile4 : n4;
  const bool hasA = cA < colEnd, hasB = cB < colEnd;
  f32x4 pA[M], gA[M], pB[M], gB[M];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    pA[j] = gA[j] = pB[j] = gB[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (hasA) {
      pA[j] = ld4(P + j * ld + 4 * cA);
      if (HAS_G) gA[j] = ld4_nt(G + j * ld + 4 * cA);
    }
    if (hasB) {
      pB[j] = ld4(P + j * ld + 4 * cB);
      if (HAS_G) gB[j] = ld4_nt(G + j * ld + 4 * cB);
    }
  }

  // ---------------- hand-off ----------------
  if (tid == 0) {
    while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
  }
  BDE_TS(4)
  __syncthreads();

  // fixed-order fp64 reduction of ALL partial tiles (every workgroup computes the same bits)
  {
    const int e = tid & (MP2 - 1), slice = tid >> 6;               // 4 slices
    double s8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int b = slice;
    for (; b + 28 < nwg; b += 32) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = ld_sc1(part + static_cast<int64_t>(b + 4 * u) * MP2 + e);
#pragma unroll
      for (int u = 0; u < 8; ++u) s8[u] += static_cast<double>(x[u]);
    }
    for (int u = 0; b < nwg; b += 4, ++u) s8[u] += static_cast<double>(ld_sc1(part + static_cast<int64_t>(b) * MP2 + e));
    red[tid] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  }
  __syncthreads();
  if (tid < MP2) gmat[tid] = (red[tid] + red[MP2 + tid]) + (red[2 * MP2 + tid] + red[3 * MP2 + tid]);
  __syncthreads();
  BDE_TS(5)
  svgd_stats_core(gmat, M, MP, sp, blockIdx.x == 0 ? kstat : nullptr, cgT, cpT);

  BDE_TS(6)
  // every workgroup has passed its go flag once all have added here: the last one lowers the flags again
  if (tid == 0) {
    const unsigned left = __hip_atomic_fetch_add(depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left == static_cast<unsigned>(nwg) - 1u) {
      for (int sh = 0; sh < kWsShards; ++sh)
        __hip_atomic_store(words + kWsGoWord + 32 * sh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (blockIdx.x == 0) {
      ws[0] = static_cast<float>(nwg);
      ws[1] = static_cast<float>(MP);
    }
  }

  // ---------------- phase 2: out = CG . G + CP . P for this workgroup's columns ----------------
  f32x4 oA[M], oB[M];
#pragma unroll
  for (int i = 0; i < M; ++i) oA[i] = oB[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < M; ++j) {
    if (HAS_G) {
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const float a = cgT[j * M + i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          oA[i][c] = __builtin_fmaf(a, gA[j][c], oA[i][c]);
          oB[i][c] = __builtin_fmaf(a, gB[j][c], oB[i][c]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const float b = cpT[j * M + i];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        oA[i][c] = __builtin_fmaf(b, pA[j][c], oA[i][c]);
        oB[i][c] = __builtin_fmaf(b, pB[j][c], oB[i][c]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < M; ++i) {
    if (hasA) st4_nt(out + i * ld + 4 * cA, oA[i]);
    if (hasB) st4_nt(out + i * ld + 4 * cB, oB[i]);
  }
  BDE_TS(7)
  // the D % 4 leftover coordinates (last workgroup)
  if (blockIdx.x == nwg - 1) {
    const int64_t e = (n4 << 2) + tid;
    if (e < D) {
      float acc[M];
#pragma unroll
      for (int i = 0; i < M; ++i) acc[i] = 0.f;
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const float p = P[j * ld + e];
        const float g = HAS_G ? G[j * ld + e] : 0.f;
#pragma unroll
        for (int i = 0; i < M; ++i) {
          if (HAS_G) acc[i] = __builtin_fmaf(cgT[j * M + i], g, acc[i]);
          acc[i] = __builtin_fmaf(cpT[j * M + i], p, acc[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < M; ++i) out[i * ld + e] = acc[i];
    }
  }
}

template <int M>
static int launch_small(const float* P, const float* G, float* out, int64_t D, int64_t ld, int grid, int tpw,
                        const StatParams& sp, float* ws, float* kstat, hipStream_t s) {
  if (G)
    hipLaunchKernelGGL((svgd_step_small_kernel<M, true>), dim3(grid), dim3(kSmallBlock), 0, s, P, G, out, D, ld, tpw, sp,
                       ws, kstat);
  else
    hipLaunchKernelGGL((svgd_step_small_kernel<M, false>), dim3(grid), dim3(kSmallBlock), 0, s, P, G, out, D, ld, tpw,
                       sp, ws, kstat);
  return to_err(hipGetLastError());
}

}  // namespace bde

using namespace bde;

extern "C" int bde_svgd_small_supported(int M, int64_t D) {
  if (M < 1 || M > 8 || D < 1) return 0;
  const int64_t n_tiles = (((D + 3) >> 2) + kSmallTile4 - 1) / kSmallTile4;
  return n_tiles <= static_cast<int64_t>(kSmallMaxGrid) * kSmallMaxTilesPerWG;
}

extern "C" int bde_svgd_step_small(const float* P, const float* G, float* out, int M, int64_t D, int64_t ld,
                                   float l2_reg, float kernel_grad_scale, float dataset_size, float sign,
                                   float h_override, int mode, void* ws, float* kstat, void* stream) {
  if (!svgd_args_ok(P, M, D, ld) || !out || !ws || !kstat || !aligned16(out) || !aligned16(ws) || (G && !aligned16(G)) ||
      out == P || (mode != 0 && mode != 1) || (mode == 0 && !G))
    return BDE_ERR_INVALID;
  if (!bde_svgd_small_supported(M, D)) return BDE_ERR_INVALID;
  const int64_t n_tiles = (((D + 3) >> 2) + kSmallTile4 - 1) / kSmallTile4;
  const int tpw = static_cast<int>((n_tiles + kSmallMaxGrid - 1) / kSmallMaxGrid);
  const int grid = static_cast<int>((n_tiles + tpw - 1) / tpw);
  const StatParams sp{l2_reg, kernel_grad_scale, dataset_size, sign, h_override,
                      static_cast<float>(std::log(static_cast<double>(M) + 1.0)), mode};
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* wsf = static_cast<float*>(ws);
  switch (M) {
#define BDE_CASE(m) \
  case m:           \
    return launch_small<m>(P, mode == 0 ? G : nullptr, out, D, ld, grid, tpw, sp, wsf, kstat, s);
    BDE_CASE(1) BDE_CASE(2) BDE_CASE(3) BDE_CASE(4) BDE_CASE(5) BDE_CASE(6) BDE_CASE(7) BDE_CASE(8)
#undef BDE_CASE
  }
  return BDE_ERR_INVALID;
}
