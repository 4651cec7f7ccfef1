// SVGD posterior update for SMALL models (CIFAR ResNet-20 scale: D = 273,610) as TWO launches of one kernel.
//
// Reference: src/algos/svgd.py:14-32,86-89 -- the same arithmetic as svgd.hip.  At a few hundred
// thousand parameters the three-launch path (gram -> kstats -> combine) is not bandwidth-bound
// any more: 35 MB of traffic is ~6 us of HBM time, but a one-workgroup statistics kernel between two
// launch boundaries and the combine's cold scalar loads made it 27 us (round-1 profile).  Here:
//
//   launch 1 (phase kSmallGramOnly)   every workgroup owns <= 512 consecutive float4 columns of P: centred Gram
//            partial of its slice on the f32 MFMA (same tiles as svgd_gram_kernel<2>), written to the workspace;
//   launch 2 (phase kSmallAfterGram)  EVERY workgroup reduces all partials in the same fixed order (fp64) and
//            evaluates the kernel statistics redundantly (a few hundred scalar operations, one wave, shuffles), keeps
//            the 2 M^2 coefficients in LDS, and combines its own columns -- or, OPT != 0, continues through the M
//            shared-state base-optimizer applications and writes the updated particles.
//
// P is read from HBM once (the second launch finds its columns in L2), G once, out written once: 12 M D bytes.
//
// Rounds 2 and 3 also ran both halves as ONE persistent launch with an in-kernel hand-off between the workgroups
// (round 2: unbounded wait, 10.7 us, could hang a shared device; round 3: bounded wait + a compare-and-swap
// COMMIT / ABORT outcome word + redo path, 14.3 us against 14.7 us for these two launches).  A protocol, a
// shared-device hazard and a recovery path in the shell for 0.3 us: removed in round 4.
#include "svgd_gram.hpp"
#include <atomic>
#include <cstdlib>

namespace bde {

// development aid (tools/kexp6.hip): per-workgroup phase timestamps, 100 MHz wall clock
#ifdef BDE_SMALL_TIMING
__device__ unsigned long long g_small_ts[256 * 16];
#define BDE_TS(k) if (threadIdx.x == 0) g_small_ts[blockIdx.x * 16 + (k)] = wall_clock64();
#else
#define BDE_TS(k)
#endif

constexpr int kSmallBlock = 512;                   // 8 waves: the Gram tiles of a workgroup are spread over more MFMA pipes
constexpr int kSmallWaves = kSmallBlock / 64;
constexpr int kSmallMaxTilesPerWG = 16;            // 16 tiles x 32 float4 columns = 512 columns = 1 per thread
constexpr int kSmallTile4 = kGramU * 8;            // float4 columns per Gram tile (PACK = 2)
constexpr int kSmallMaxGrid = 256;

__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

enum : int { kSmallGramOnly = 1, kSmallAfterGram = 2 };

// The kernel statistics of svgd_stats_core for M <= 8 (M * M <= 64 entries), evaluated by ONE wave with
// cross-lane shuffles instead of LDS round trips and workgroup barriers (2.0 us -> well under 1 us on the
// critical path of the single-launch kernel).  Same arithmetic, same results.  Lane e < M * M owns entry
// (i, j) = (e / M, e % M).  All 64 lanes of the wave must call it.
__device__ __forceinline__ void svgd_stats_wave(const double* gmat, int M, const StatParams sp,
                                                float* __restrict__ kstat, float* lds_cg, float* lds_cp) {
  const int lane = threadIdx.x & 63;
  const int n = M * M;
  const bool act = lane < n;
  const int i = act ? lane / M : 0, j = act ? lane % M : 0;
  float d2 = 0.f;
  if (act) {
    double d = gmat[i * 8 + i] + gmat[j * 8 + j] - 2.0 * gmat[i * 8 + j];   // svgd.py:15
    if (d < 0.0 || i == j) d = 0.0;
    d2 = static_cast<float>(d);
  }
  // torch.quantile(d2, 0.5), 'linear' interpolation, fp32 like the reference (svgd.py:18; the M diagonal zeros count):
  // the values at ranks floor / ceil of (n - 1) / 2.  Every lane counts the entries below and not above its own value
  // from an LDS copy of the 64 values (16 broadcast b128 reads; inactive lanes hold +inf): the entry of rank r is any
  // lane with below <= r < not_above.  (64 v_readlane broadcasts took 1.1 of the statistics' 1.75 us; a radix select
  // over the bits with one ballot per bit was slower still, 2.6 us: 31 dependent VALU -> SALU round trips.)
  const float pos = 0.5f * static_cast<float>(n - 1);
  const float lo = floorf(pos);
  const float wgt = pos - lo;
  const int r_lo = static_cast<int>(lo), r_hi = static_cast<int>(ceilf(pos));
  lds_cg[lane] = act ? d2 : __builtin_inff();                     // scratch: the coefficients are written at the end
  int below = 0, not_above = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const f32x4 o = *reinterpret_cast<const f32x4*>(lds_cg + 4 * q);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      below += o[c] < d2 ? 1 : 0;
      not_above += o[c] <= d2 ? 1 : 0;
    }
  }
  const unsigned long long m_lo = __ballot(act && below <= r_lo && r_lo < not_above);
  const unsigned long long m_hi = __ballot(act && below <= r_hi && r_hi < not_above);
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), __builtin_ctzll(m_lo)));
  const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), __builtin_ctzll(m_hi)));
  const float med = (fabsf(wgt) < 0.5f) ? a + wgt * (b - a) : b - (b - a) * (1.0f - wgt);   // at::lerp
  float h = __builtin_sqrtf((0.5f * med) / sp.log_m1) + 1e-8f;                              // svgd.py:18
  if (sp.h_override > 0.f) h = sp.h_override;
  const float k = act ? expf(-d2 / (2.0f * (h * h))) : 0.f;                                // svgd.py:21
  float rowsum = 0.f;                                                                      // sum_j K[i][j], j ascending
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const float kj = __shfl(k, i * M + (jj < M ? jj : 0), 64);
    if (jj < M) rowsum += kj;
  }
  const double h2 = static_cast<double>(h) * static_cast<double>(h);
  const double s_rep = static_cast<double>(sp.kernel_grad_scale) / (static_cast<double>(sp.dataset_size) * h2);
  if (act) {
    const double kij = k;
    const double rep = ((i == j) ? static_cast<double>(rowsum) : 0.0) - kij;
    double cg, cp;
    if (sp.mode == 0) {
      cg = static_cast<double>(sp.sign) * (-kij);
      cp = static_cast<double>(sp.sign) * (-kij * (0.5 * static_cast<double>(sp.l2_reg)) + s_rep * rep);
    } else {
      cg = 0.0;
      cp = rep / h2;
    }
    lds_cg[j * M + i] = static_cast<float>(cg);
    lds_cp[j * M + i] = static_cast<float>(cp);
    if (kstat) {
      const int oK = 0, oD2 = n, oRow = 2 * n, oMisc = 2 * n + M, oCG = oMisc + 4, oCP = oCG + n;
      kstat[oK + lane] = k;
      kstat[oD2 + lane] = d2;
      kstat[oCG + j * M + i] = static_cast<float>(cg);
      kstat[oCP + j * M + i] = static_cast<float>(cp);
      if (j == 0) kstat[oRow + i] = rowsum;
      if (lane == 0) {
        kstat[oMisc + 0] = h;
        kstat[oMisc + 1] = med;
        kstat[oMisc + 2] = static_cast<float>(s_rep);
        kstat[oMisc + 3] = static_cast<float>(M);
      }
    }
  }
}

// OPT = 0: out = sign * phi (or grad_kernel in mode 1).  OPT = 1 / 2: the shared-state SGD / Adam applications of
// svgd.py:92-103 follow in registers (svgd_fused.hip's loop) and the updated particles are written back over P
// (`out` must be P; s0 / s1 are the optimizer state) -- the FULL SVGDOptimizer.step minus forward/backward in one
// launch, (12 M + 8) D bytes.
template <int M, bool HAS_G, int OPT>
__global__ __launch_bounds__(kSmallBlock, 1) void svgd_step_small_kernel(const float* P, const float* G, float* out,
                                                                        int64_t D, int64_t ld, int tiles_per_wg,
                                                                        StatParams sp, float* __restrict__ ws,
                                                                        float* __restrict__ kstat, float* s0, float* s1,
                                                                        SgdParams sk, AdamParams ak, AdamSteps ast,
                                                                        int phase) {
  constexpr int MP = 8, MP2 = 64;
  __shared__ float tile[kSmallWaves][16][17];
  __shared__ double red[(kSmallBlock / (MP2 / 2)) * MP2];           // [16 slices][64]
  __shared__ double gmat[MP2];
  __shared__ __attribute__((aligned(16))) float cgT[MP2];
  __shared__ __attribute__((aligned(16))) float cpT[MP2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwg = gridDim.x;
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
  if (phase != kSmallAfterGram) {
    // all of this wave's tiles are requested before the first DPP/MFMA chain waits on them
    constexpr int TW = kSmallMaxTilesPerWG / kSmallWaves;          // tiles per wave, at most
    f32x4 v[TW][kGramU];
#pragma unroll
    for (int k = 0; k < TW; ++k) {
      const int64_t t = t0 + wave + kSmallWaves * k;
      if (t < t1) gram_load_tile<8, false>(v[k], rowp, valid, t, kSmallTile4, c4, n4c, D);
    }
#pragma unroll
    for (int k = 0; k < TW; ++k) {
      const int64_t t = t0 + wave + kSmallWaves * k;
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
  if (phase != kSmallAfterGram) {
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[wave][4 * kq + r][r16] = acc0[r] + acc1[r];
    __syncthreads();
  }
  BDE_TS(1)
  if (phase != kSmallAfterGram && tid < MP2) {                     // one wave, one store instruction per 128-B line
    const int pi = tid / MP, pj = tid % MP;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kSmallWaves; ++w) s += tile[w][pi][pj] + tile[w][pi + 8][pj + 8];
    st_sc1(part + static_cast<int64_t>(blockIdx.x) * MP2 + tid, s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the write-through stores have left this CU
    BDE_TS(2)
  }
  BDE_TS(3)
  if (phase == kSmallGramOnly) {                                   // first of two launches: the partials are the result
    if (blockIdx.x == 0 && tid == 0) {
      ws[0] = static_cast<float>(nwg);
      ws[1] = static_cast<float>(MP);
    }
    return;
  }

  // ---------------- second launch: this workgroup's operands, requested before the partials are reduced ----------------
  const int64_t n4 = D >> 2;                                       // full float4 columns
  const int64_t cA = t0 * kSmallTile4 + tid;
  const int64_t colEnd = (t1 * kSmallTile4 < n4) ? t1 * kSmallTile4 : n4;
  const bool hasA = cA < colEnd;
  f32x4 pA[M], gA[M];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    pA[j] = gA[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (hasA) {
      pA[j] = ld4(P + j * ld + 4 * cA);
      if (HAS_G) gA[j] = ld4_nt(G + j * ld + 4 * cA);
    }
  }
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  if (OPT != 0 && hasA) {
    if (OPT == 2 || (sk.momentum != 0.f && !sk.first)) a0 = ld4(s0 + 4 * cA);
    if (OPT == 2) a1 = ld4(s1 + 4 * cA);
  }

  BDE_TS(4)
  // fixed-order fp64 reduction of ALL partial tiles (every workgroup computes the same bits): 8-byte sc1 loads, all
  // of a thread's loads in flight before the first add, unconditional (a predicated load gets a basic block and an
  // s_waitcnt of its own: 8 us instead of 1) -- slots past the grid re-read the last partial and are dropped
  {
    constexpr int PAIRS = MP2 / 2;                                 // 32 float pairs per partial tile
    constexpr int SLICES = kSmallBlock / PAIRS;                    // 16
    constexpr int NL = kSmallMaxGrid / SLICES;                     // partials per thread, at most (16)
    const int q = tid & (PAIRS - 1), slice = tid / PAIRS;
    unsigned long long x[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int b = slice + SLICES * u;
      x[u] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(part + static_cast<int64_t>(b < nwg ? b : nwg - 1) * MP2) + q,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const bool in = slice + SLICES * u < nwg;
      s0 += in ? static_cast<double>(__uint_as_float(static_cast<unsigned>(x[u]))) : 0.0;
      s1 += in ? static_cast<double>(__uint_as_float(static_cast<unsigned>(x[u] >> 32))) : 0.0;
    }
    red[slice * MP2 + 2 * q] = s0;
    red[slice * MP2 + 2 * q + 1] = s1;
  }
  __syncthreads();
  if (wave == 0) {
    double s = 0.0;
#pragma unroll
    for (int sl = 0; sl < kSmallBlock / (MP2 / 2); ++sl) s += red[sl * MP2 + lane];
    gmat[lane] = s;
  }
  BDE_TS(5)
  if (wave == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");         // gmat written above is read across lanes below
    svgd_stats_wave(gmat, M, sp, blockIdx.x == 0 ? kstat : nullptr, cgT, cpT);
  }
  __syncthreads();
  BDE_TS(6)

  // ---------------- phase 2: out = CG . G + CP . P for this workgroup's columns ----------------
  f32x4 oA[M];
#pragma unroll
  for (int i = 0; i < M; ++i) oA[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < M; ++j) {
    if (HAS_G) {
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const float a = cgT[j * M + i];
#pragma unroll
        for (int c = 0; c < 4; ++c) oA[i][c] = __builtin_fmaf(a, gA[j][c], oA[i][c]);
      }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const float b = cpT[j * M + i];
#pragma unroll
      for (int c = 0; c < 4; ++c) oA[i][c] = __builtin_fmaf(b, pA[j][c], oA[i][c]);
    }
  }
  if (hasA) {
    if (OPT == 0) {
#pragma unroll
      for (int i = 0; i < M; ++i) st4_nt(out + i * ld + 4 * cA, oA[i]);
    } else {
      // oA[i] = -phi_i is the gradient the reference hands to the base optimizer (svgd.py:95); the particles are
      // walked in order with the SHARED optimizer state in registers
#pragma unroll
      for (int i = 0; i < M; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float b = a0[c];
          if (OPT == 1) {
            pA[i][c] = sgd_apply(pA[i][c], oA[i][c], b, sk, i == 0);
          } else {
            float v = a1[c];
            pA[i][c] = adam_apply(pA[i][c], oA[i][c], b, v, ak, ast.step_size[i], ast.bc2_sqrt[i]);
            a1[c] = v;
          }
          a0[c] = b;
        }
        st4(out + i * ld + 4 * cA, pA[i]);
      }
      if (OPT == 2 || sk.momentum != 0.f) st4(s0 + 4 * cA, a0);
      if (OPT == 2) st4(s1 + 4 * cA, a1);
    }
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
      if (OPT == 0) {
#pragma unroll
        for (int i = 0; i < M; ++i) out[i * ld + e] = acc[i];
      } else {
        float b = 0.f, v = 0.f;
        if (OPT == 2 || (sk.momentum != 0.f && !sk.first)) b = s0[e];
        if (OPT == 2) v = s1[e];
#pragma unroll
        for (int i = 0; i < M; ++i) {
          const float p = P[i * ld + e];
          out[i * ld + e] = (OPT == 1) ? sgd_apply(p, acc[i], b, sk, i == 0)
                                       : adam_apply(p, acc[i], b, v, ak, ast.step_size[i], ast.bc2_sqrt[i]);
        }
        if (OPT == 2 || sk.momentum != 0.f) s0[e] = b;
        if (OPT == 2) s1[e] = v;
      }
    }
  }
  if (blockIdx.x == 0 && tid == 0) {
    ws[0] = static_cast<float>(nwg);
    ws[1] = static_cast<float>(MP);
  }
}

struct SmallOpt {
  int kind = 0;                    // 0 none, 1 sgd, 2 adam
  float* s0 = nullptr;
  float* s1 = nullptr;
  SgdParams sk{};
  AdamParams ak{};
  AdamSteps ast{};
};

template <int M>
static int launch_small(const float* P, const float* G, float* out, int64_t D, int64_t ld, int grid, int tpw,
                        const StatParams& sp, float* ws, float* kstat, const SmallOpt& o, hipStream_t s) {
#define BDE_SMALL_LAUNCH(HG, OPT, PHASE)                                                                             \
  hipLaunchKernelGGL((svgd_step_small_kernel<M, HG, OPT>), dim3(grid), dim3(kSmallBlock), 0, s, P, G, out, D, ld, tpw, \
                     sp, ws, kstat, o.s0, o.s1, o.sk, o.ak, o.ast, PHASE)
#define BDE_SMALL_VARIANT(PHASE)                   \
  do {                                             \
    if (o.kind == 1) BDE_SMALL_LAUNCH(true, 1, PHASE);      \
    else if (o.kind == 2) BDE_SMALL_LAUNCH(true, 2, PHASE); \
    else if (G) BDE_SMALL_LAUNCH(true, 0, PHASE);           \
    else BDE_SMALL_LAUNCH(false, 0, PHASE);                 \
  } while (0)
  BDE_SMALL_VARIANT(kSmallGramOnly);
  BDE_SMALL_VARIANT(kSmallAfterGram);
#undef BDE_SMALL_VARIANT
#undef BDE_SMALL_LAUNCH
  return to_err(hipGetLastError());
}

// One workgroup per CU: the grid is what the CURRENT device holds at once (CUs x workgroups per CU at this kernel's
// register / LDS footprint, at most 256), queried once per device -- a partition of the chip (CPX mode: 32 CUs) gets a
// smaller grid and a smaller limit on D; without a device (the build container) the full-chip figure applies.
static int small_resident_limit() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return kSmallMaxGrid;
  int v = cache[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  int cus = 0, per_cu = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, svgd_step_small_kernel<8, true, 2>, kSmallBlock, 0) != hipSuccess ||
      cus < 1 || per_cu < 1)
    v = 1;                                                          // unknown: never more than one workgroup
  else
    v = static_cast<int>(std::min<int64_t>(static_cast<int64_t>(cus) * per_cu, kSmallMaxGrid));
  cache[dev].store(v, std::memory_order_relaxed);
  return v;
}

static int small_dispatch(const float* P, const float* G, float* out, int M, int64_t D, int64_t ld, const StatParams& sp,
                          float* ws, float* kstat, const SmallOpt& o, hipStream_t s) {
  const int64_t n_tiles = (((D + 3) >> 2) + kSmallTile4 - 1) / kSmallTile4;
  const int max_grid = small_resident_limit();
  const int tpw = static_cast<int>((n_tiles + max_grid - 1) / max_grid);
  if (tpw > kSmallMaxTilesPerWG) return BDE_ERR_INVALID;             // bde_svgd_small_supported() says no on this device
  const int grid = static_cast<int>((n_tiles + tpw - 1) / tpw);
  switch (M) {
#define BDE_CASE(m) \
  case m:           \
    return launch_small<m>(P, G, out, D, ld, grid, tpw, sp, ws, kstat, o, s);
    BDE_CASE(1) BDE_CASE(2) BDE_CASE(3) BDE_CASE(4) BDE_CASE(5) BDE_CASE(6) BDE_CASE(7) BDE_CASE(8)
#undef BDE_CASE
  }
  return BDE_ERR_INVALID;
}

}  // namespace bde

using namespace bde;

extern "C" int bde_svgd_small_supported(int M, int64_t D) {
  if (M < 1 || M > 8 || D < 1) return 0;
  const int64_t n_tiles = (((D + 3) >> 2) + kSmallTile4 - 1) / kSmallTile4;
  return n_tiles <= static_cast<int64_t>(small_resident_limit()) * kSmallMaxTilesPerWG;
}

static StatParams small_stat_params(int M, float l2_reg, float kernel_grad_scale, float dataset_size, float sign,
                                    float h_override, int mode) {
  return StatParams{l2_reg, kernel_grad_scale, dataset_size, sign, h_override,
                    static_cast<float>(std::log(static_cast<double>(M) + 1.0)), mode};
}

extern "C" int bde_svgd_step_small(const float* P, const float* G, float* out, int M, int64_t D, int64_t ld,
                                   float l2_reg, float kernel_grad_scale, float dataset_size, float sign,
                                   float h_override, int mode, void* ws, float* kstat, void* stream) {
  if (!svgd_args_ok(P, M, D, ld) || !out || !ws || !kstat || !aligned16(out) || !aligned16(ws) || (G && !aligned16(G)) ||
      out == P || (mode != 0 && mode != 1) || (mode == 0 && !G))
    return BDE_ERR_INVALID;
  if (!bde_svgd_small_supported(M, D)) return BDE_ERR_INVALID;
  return small_dispatch(P, mode == 0 ? G : nullptr, out, M, D, ld,
                        small_stat_params(M, l2_reg, kernel_grad_scale, dataset_size, sign, h_override, mode),
                        static_cast<float*>(ws), kstat, SmallOpt{}, static_cast<hipStream_t>(stream));
}

extern "C" int bde_svgd_step_small_sgd(float* P, const float* G, float* momentum_buf, int M, int64_t D, int64_t ld,
                                       float l2_reg, float kernel_grad_scale, float dataset_size, double lr,
                                       double momentum, double dampening, double weight_decay, int nesterov, int first,
                                       void* ws, float* kstat, void* stream) {
  if (!svgd_args_ok(P, M, D, ld) || !G || !aligned16(G) || !ws || !aligned16(ws) || !kstat ||
      (momentum != 0.0 && (!momentum_buf || !aligned16(momentum_buf))) || !bde_svgd_small_supported(M, D))
    return BDE_ERR_INVALID;
  SmallOpt o;
  o.kind = 1;
  o.s0 = momentum_buf;
  o.sk = SgdParams{static_cast<float>(lr), static_cast<float>(momentum), static_cast<float>(1.0 - dampening),
                   static_cast<float>(weight_decay), nesterov, first};
  return small_dispatch(P, G, P, M, D, ld, small_stat_params(M, l2_reg, kernel_grad_scale, dataset_size, -1.f, 0.f, 0),
                        static_cast<float*>(ws), kstat, o, static_cast<hipStream_t>(stream));
}

extern "C" int bde_svgd_step_small_adam(float* P, const float* G, float* exp_avg, float* exp_avg_sq, int M, int64_t D,
                                        int64_t ld, float l2_reg, float kernel_grad_scale, float dataset_size, double lr,
                                        double beta1, double beta2, double eps, double weight_decay, int64_t step0,
                                        void* ws, float* kstat, void* stream) {
  if (!svgd_args_ok(P, M, D, ld) || !G || !aligned16(G) || !ws || !aligned16(ws) || !kstat || !exp_avg || !exp_avg_sq ||
      !aligned16(exp_avg) || !aligned16(exp_avg_sq) || step0 < 0 || !bde_svgd_small_supported(M, D))
    return BDE_ERR_INVALID;
  SmallOpt o;
  o.kind = 2;
  o.s0 = exp_avg;
  o.s1 = exp_avg_sq;
  o.ak = AdamParams{static_cast<float>(beta1), static_cast<float>(beta2), static_cast<float>(1.0 - beta1),
                    static_cast<float>(1.0 - beta2), static_cast<float>(eps), static_cast<float>(weight_decay)};
  o.ast = make_adam_steps(lr, beta1, beta2, step0);
  return small_dispatch(P, G, P, M, D, ld, small_stat_params(M, l2_reg, kernel_grad_scale, dataset_size, -1.f, 0.f, 0),
                        static_cast<float*>(ws), kstat, o, static_cast<hipStream_t>(stream));
}

// bde_init(): load this translation unit's code object on the current device now (HIP otherwise uploads it at the
// first launch of one of its kernels).  Internal to the library (not exported).
extern "C" __attribute__((visibility("hidden"))) int bde_internal_load_svgd_small(void) {
  hipFuncAttributes attr;
  return bde::to_err(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&bde::svgd_step_small_kernel<8, true, 1>)));
}
