// Pieces of the SVGD Gram / kernel-statistics computation shared by the streaming path (svgd.hip:
// gram -> kstats -> combine as three launches) and the single-launch path for small models
// (svgd_small.hip).  Reference: src/algos/svgd.py:14-32.
#pragma once
#include "svgd_shared.hpp"

namespace bde {

// A/B switches (tools/kexp6.hip) for the three-stage Gram pass: BDE_GRAM_NT = flavour of the loads behind the
// nt/cacheable split point (GramRows::nt_split), BDE_GRAM_REVERSE = direction of the walk.
#ifndef BDE_GRAM_NT
#define BDE_GRAM_NT false
#endif
#ifndef BDE_GRAM_REVERSE
#define BDE_GRAM_REVERSE 0
#endif
#if BDE_GRAM_REVERSE
#define BDE_GRAM_TILE(t) (n_tiles - 1 - (t))
#else
#define BDE_GRAM_TILE(t) (t)
#endif

constexpr int kGramBlock = 256;            // 4 waves
constexpr int kGramU = 4;                  // float4 loads in flight per lane per iteration

using f32x4acc = __attribute__((ext_vector_type(4))) float;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}

// Sum over the 8 (PACK 2) or 16 (PACK 1) consecutive lanes that hold one
// coordinate of all particles; every lane of the group gets the sum.
template <int PACK>
__device__ __forceinline__ float group_sum(float x) {
  float s = x + dpp_mov<0xB1>(x);      // quad_perm [1,0,3,2]
  s += dpp_mov<0x4E>(s);               // quad_perm [2,3,0,1]
  s += dpp_mov<0x141>(s);              // row_half_mirror: 8 lanes
  if (PACK == 1) s += dpp_mov<0x140>(s);   // row_mirror: 16 lanes
  return s;
}

// One tile = kGramU float4 columns per lane.  Full tiles take the branch-free path; the ragged
// last tile masks by index (never by multiplication: the row padding may hold NaNs).
// NT: non-temporal loads (the Gram pass alone then streams at 6.5 instead of 5.9 TB/s, profiles/r02_probes_and_variants_before.txt --
// but see BDE_GRAM_NT above); the single-launch path re-reads the particles from L2 and always loads them normally.
template <int W4, bool NT = false>
__device__ __forceinline__ void gram_load_tile(f32x4 (&v)[kGramU], const float* __restrict__ rowp, bool valid,
                                               int64_t t, int64_t tile4, int c4, int64_t n4, int64_t D) {
  const int64_t base4 = t * tile4 + c4;
  if ((t + 1) * tile4 * 4 <= D) {                    // wave-uniform: every column full
#pragma unroll
    for (int u = 0; u < kGramU; ++u) v[u] = NT ? ld4_nt(rowp + 4 * (base4 + u * W4)) : ld4(rowp + 4 * (base4 + u * W4));
  } else {
#pragma unroll
    for (int u = 0; u < kGramU; ++u) {
      const int64_t col = base4 + u * W4;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (valid && col < n4) {
        x = ld4(rowp + 4 * col);                     // in bounds: ld >= roundup4(D)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * col + j >= D) x[j] = 0.f;
      }
      v[u] = x;
    }
  }
}

// PACK = 2: M <= 8, the 16 tile rows are (particle, d-chunk 0/1); PACK = 1: M <= 16.
// Lanes of padded particle rows (prow >= M) read row 0 and are zeroed after the load, so the
// hot loop has no divergent branches.  The next tile's loads are issued before this tile's
// DPP/MFMA work (register double buffering).
//
// Row map (PACK = 1 only): tile rows 0..7 are particles rowA .. rowA+nA-1, tile rows 8..15 are particles
// rowB .. rowB+nB-1.  M <= 16 uses (0, min(M,8), 8, M-8); the generic path (M > 16) launches one such
// tile per PAIR of 8-particle groups, each into its own slice of ws (tile_slot).
struct GramRows {
  int rowA, nA, rowB, nB, tile_slot;
  int nt_split = 0;      // per mille of the tile walk loaded non-temporally (the rest stays cacheable)
};

// The tail of the Gram walk that is kept cacheable for the combine pass: what fits the Infinity Cache (256 MiB,
// minus headroom for the pass's own partials and whatever else is live).
static inline int gram_nt_split(int M, int64_t D, double keep = 240e6) {
  const double total = 4.0 * static_cast<double>(M) * static_cast<double>(D);
  if (keep <= 0.0) return 1000;                      // nothing kept cacheable: the whole walk non-temporal
  if (total <= keep) return 0;
  return static_cast<int>(1000.0 * (1.0 - keep / total));
}


// ---------------------------------------------------------------- statistics --
struct StatParams {
  float l2_reg, kernel_grad_scale, dataset_size, sign, h_override, log_m1;
  int mode;
};

// From the reduced (centred) Gram matrix gmat [MP, MP] (fp64, in LDS) to d2, the torch.quantile median, the
// bandwidth, K and the coefficient matrices of the combine pass.  Every thread of the workgroup must call it
// (it contains barriers); blockDim.x >= M * M.  Results go to `kstat` (global, may be null) and/or to the LDS
// arrays lds_cg / lds_cp ([j * M + i], may be null).
__device__ __forceinline__ void svgd_stats_core(const double* gmat, int M, int MP, const StatParams sp,
                                                float* __restrict__ kstat, float* lds_cg, float* lds_cp) {
  __shared__ float d2f[256];
  __shared__ float sorted[256];
  __shared__ float kmat[256];
  __shared__ float rowsum[16];
  __shared__ float hs[2];
  const int tid = threadIdx.x;
  const int n = M * M;
  if (tid < n) {
    const int i = tid / M, j = tid % M;
    double d = gmat[i * MP + i] + gmat[j * MP + j] - 2.0 * gmat[i * MP + j];   // svgd.py:15
    if (d < 0.0 || i == j) d = 0.0;
    d2f[tid] = static_cast<float>(d);
  }
  __syncthreads();
  // rank sort of the M*M distances (diagonal zeros included, svgd.py:18)
  if (tid < n) {
    const float v = d2f[tid];
    int rank = 0;
    for (int u = 0; u < n; ++u) {
      const float o = d2f[u];
      rank += (o < v || (o == v && u < tid)) ? 1 : 0;
    }
    sorted[rank] = v;
  }
  __syncthreads();
  if (tid == 0) {
    // torch.quantile(d2, 0.5), 'linear' interpolation, fp32 like the reference
    const float pos = 0.5f * static_cast<float>(n - 1);
    const float lo = floorf(pos);
    const float wgt = pos - lo;
    const float a = sorted[static_cast<int>(lo)], b = sorted[static_cast<int>(ceilf(pos))];
    const float med = (fabsf(wgt) < 0.5f) ? a + wgt * (b - a) : b - (b - a) * (1.0f - wgt);   // at::lerp
    float h = __builtin_sqrtf((0.5f * med) / sp.log_m1) + 1e-8f;                                 // svgd.py:18
    if (sp.h_override > 0.f) h = sp.h_override;
    hs[0] = h;
    hs[1] = med;
  }
  __syncthreads();
  const float h = hs[0];
  if (tid < n) kmat[tid] = expf(-d2f[tid] / (2.0f * (h * h)));   // svgd.py:21
  __syncthreads();
  if (tid < M) {
    float s = 0.f;
    for (int j = 0; j < M; ++j) s += kmat[tid * M + j];
    rowsum[tid] = s;
  }
  __syncthreads();

  const int oK = 0, oD2 = n, oRow = 2 * n, oMisc = 2 * n + M, oCG = oMisc + 4, oCP = oCG + n;
  const double h2 = static_cast<double>(h) * static_cast<double>(h);
  const double s_rep = static_cast<double>(sp.kernel_grad_scale) / (static_cast<double>(sp.dataset_size) * h2);
  if (tid < n) {
    const int i = tid / M, j = tid % M;
    const double kij = kmat[tid];
    const double rep = ((i == j) ? static_cast<double>(rowsum[i]) : 0.0) - kij;   // rowsum_i [i==j] - K_ij
    double cg, cp;
    if (sp.mode == 0) {
      // phi_i = sum_j -K_ij (G_j + l2/2 P_j) + s_rep * (rowsum_i P_i - K_ij P_j)   (svgd.py:86-89)
      cg = static_cast<double>(sp.sign) * (-kij);
      cp = static_cast<double>(sp.sign) * (-kij * (0.5 * static_cast<double>(sp.l2_reg)) + s_rep * rep);
    } else {
      cg = 0.0;
      cp = rep / h2;                                                              // svgd.py:23,31
    }
    if (kstat) {
      kstat[oK + tid] = kmat[tid];
      kstat[oD2 + tid] = d2f[tid];
      kstat[oCG + j * M + i] = static_cast<float>(cg);
      kstat[oCP + j * M + i] = static_cast<float>(cp);
    }
    if (lds_cg) {
      lds_cg[j * M + i] = static_cast<float>(cg);
      lds_cp[j * M + i] = static_cast<float>(cp);
    }
  }
  if (kstat) {
    if (tid < M) kstat[oRow + tid] = rowsum[tid];
    if (tid == 0) {
      kstat[oMisc + 0] = h;
      kstat[oMisc + 1] = hs[1];
      kstat[oMisc + 2] = static_cast<float>(s_rep);
      kstat[oMisc + 3] = static_cast<float>(M);
    }
  }
  if (lds_cg) __syncthreads();
}

}  // namespace bde
