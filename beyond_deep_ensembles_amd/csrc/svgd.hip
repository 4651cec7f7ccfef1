// SVGD posterior update over M flattened particles.
//
// Reference: src/algos/svgd.py:14-32 (rbf) and :83-103 (step's no_grad block).
// The reference gathers M*n_tensors per-tensor states into two [M, D]
// matrices, then runs cdist**2 -> quantile -> exp -> rowsum*P - K@P -> K@(-G)
// as ~10 separate ATen passes.  Here the particles and gradients LIVE in flat
// [M, ld] buffers and the whole update is
//
//   gram    (reads P once,  4*M*D B): mean-centred Gram partials on the f32
//           MFMA (v_mfma_f32_16x16x4_f32).  Centring matters: in the real
//           configs the particles share a pretrained backbone, so ||x||^2 >>
//           d^2 and the plain Gram trick loses 3 digits; centred by the
//           per-coordinate particle mean (DPP butterfly across the lanes that
//           hold the M particles) the error is ~1e-7.
//   kstats  (one workgroup): fixed-order fp64 reduction of the partials ->
//           d2 -> torch.quantile median -> h -> K -> coefficient matrices.
//   combine (reads P, G, writes out, 12*M*D B): out = CG@G + CP@P with the
//           2*M*M coefficients in SGPRs; out may alias G, so -phi lands
//           directly in the gradient rows the base optimizer consumes.
//
// 16*M*D bytes and 6*M*M*D flop per step: ~3 flop/B at M = 8, far below the
// fp32 ridge (~20 flop/B) -> HBM-bound; the MFMA is used for the Gram
// contraction because it leaves the VALU free for the centring, not because
// the kernel is matrix-bound.
#include <atomic>
#include "svgd_gram.hpp"

namespace bde {

template <int PACK>
__global__ __launch_bounds__(kGramBlock) void svgd_gram_kernel(const float* __restrict__ P, int M, int64_t D,
                                                              int64_t ld, float* __restrict__ ws, GramRows rows) {
  constexpr int W4 = (PACK == 2) ? 8 : 4;           // float4 columns one wave-load covers
  constexpr int MP = (PACK == 2) ? 8 : 16;          // padded particle count
  __shared__ float tile[kGramBlock / 64][16][17];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int c4 = (PACK == 2) ? ((r16 >> 3) * 4 + kq) : kq;
  int prow;
  bool valid;
  float inv_m;
  if (PACK == 2) {
    prow = r16 & 7;
    valid = prow < M;
    inv_m = 1.0f / static_cast<float>(M);
  } else {
    const bool hi = r16 >= 8;
    const int r = r16 & 7;
    valid = r < (hi ? rows.nB : rows.nA);
    prow = (hi ? rows.rowB : rows.rowA) + r;
    inv_m = 1.0f / static_cast<float>(rows.nA + rows.nB);
  }
  const float* rowp = P + static_cast<int64_t>(valid ? prow : 0) * ld;

  const int64_t n4 = (D + 3) >> 2;                   // float4 columns (last may be partial)
  const int64_t tile4 = static_cast<int64_t>(kGramU) * W4;
  const int64_t n_tiles = (n4 + tile4 - 1) / tile4;
  const int64_t waves_total = static_cast<int64_t>(gridDim.x) * (kGramBlock / 64);

  f32x4acc acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  f32x4 cur[kGramU], nxt[kGramU];
  int64_t t = static_cast<int64_t>(blockIdx.x) * (kGramBlock / 64) + wave;
  // Tiles before rows.nt_split (per mille of the walk) are loaded non-temporally, the rest normally: only the tail
  // of the walk -- as much as the 256 MB Infinity Cache holds -- can still be resident when the combine pass re-reads
  // the particles, and the streaming head of the walk is 14 % faster with nt loads.  Measured at M = 8, D = 23.9 M:
  // step 0.534-0.538 ms with the head (2/3) nt vs 0.544 all plain, 0.579 all nt, 0.548-0.553 with 80 % nt
  // (profiles/r02_gram_split_ab.txt).
  const int64_t nt_until = n_tiles * rows.nt_split / 1000;
  auto load_tile = [&](f32x4 (&v)[kGramU], int64_t tt) {
    if (tt < nt_until)
      gram_load_tile<W4, true>(v, rowp, valid, BDE_GRAM_TILE(tt), tile4, c4, n4, D);
    else
      gram_load_tile<W4, BDE_GRAM_NT>(v, rowp, valid, BDE_GRAM_TILE(tt), tile4, c4, n4, D);
  };
  if (t < n_tiles) load_tile(cur, t);
  for (; t < n_tiles; t += waves_total) {
    const int64_t tn = t + waves_total;
    if (tn < n_tiles) load_tile(nxt, tn);
#pragma unroll
    for (int u = 0; u < kGramU; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = valid ? cur[u][j] : 0.f;
        const float s = group_sum<PACK>(x);
        const float q = valid ? (x - s * inv_m) : 0.f;
        if ((j & 1) == 0)
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(q, q, acc0, 0, 0, 0);
        else
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q, q, acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < kGramU; ++u) cur[u] = nxt[u];
  }

  // C[i][j]: lane holds column j = lane & 15, rows i = 4 * (lane >> 4) + r
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[wave][4 * kq + r][r16] = acc0[r] + acc1[r];
  __syncthreads();
  if (threadIdx.x < MP * MP) {
    const int pi = threadIdx.x / MP, pj = threadIdx.x % MP;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kGramBlock / 64; ++w) {
      s += tile[w][pi][pj];
      if (PACK == 2) s += tile[w][pi + 8][pj + 8];
    }
    const int64_t slot = (PACK == 1) ? static_cast<int64_t>(rows.tile_slot) * kGramMaxBlocks * 256 : 0;
    ws[kWsHeaderFloats + slot + static_cast<int64_t>(blockIdx.x) * (MP * MP) + threadIdx.x] = s;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ws[0] = static_cast<float>(gridDim.x);
    ws[1] = static_cast<float>(MP);
  }
}

// ---------------------------------------------------------------- kstats --
constexpr int kStatsBlock = 1024;

// Fixed-order fp64 reduction of the per-workgroup partial Gram tiles of `ws` into gmat [MP * MP] (LDS).
__device__ __forceinline__ void reduce_gram_partials(const float* __restrict__ ws, double* red, double* gmat, int& MP_out) {
  const int nb = static_cast<int>(ws[0]);
  const int MP = static_cast<int>(ws[1]);
  const int mp2 = MP * MP;
  const float* part = ws + kWsHeaderFloats;
  const int tid = threadIdx.x;
  const int nslices = kStatsBlock / mp2;
  {
    // 8 independent accumulators keep 8 loads in flight (a single dependent chain made this
    // latency-bound: 45 us for 2048 partials); the summation order is still fixed.
    const int e = tid % mp2, slice = tid / mp2;
    double s8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int b = slice;
    for (; b + 7 * nslices < nb; b += 8 * nslices) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s8[u] += static_cast<double>(part[static_cast<int64_t>(b + u * nslices) * mp2 + e]);
    }
    for (int u = 0; b < nb; b += nslices, ++u) s8[u] += static_cast<double>(part[static_cast<int64_t>(b) * mp2 + e]);
    red[tid] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  }
  __syncthreads();
  if (tid < mp2) {
    double s = 0.0;
    for (int sl = 0; sl < nslices; ++sl) s += red[sl * mp2 + tid];
    gmat[tid] = s;
  }
  __syncthreads();
  MP_out = MP;
}

__global__ __launch_bounds__(kStatsBlock) void svgd_kstats_kernel(const float* __restrict__ ws, int M, StatParams sp,
                                                                 float* __restrict__ kstat) {
  __shared__ double red[kStatsBlock];
  __shared__ double gmat[256];
  int MP;
  reduce_gram_partials(ws, red, gmat, MP);
  svgd_stats_core(gmat, M, MP, sp, kstat, nullptr, nullptr);
}

// Dimension-sharded multi-GPU update: every rank reduces the Gram partials of ITS column slice to an fp64
// [MP, MP] matrix (gram_finish), the ranks exchange those (MP*MP doubles each), and every rank sums them in
// rank order and evaluates the statistics (kstats_gmat) -- identical bits on all ranks.
__global__ __launch_bounds__(kStatsBlock) void svgd_gram_finish_kernel(const float* __restrict__ ws,
                                                                      double* __restrict__ gmat_out) {
  __shared__ double red[kStatsBlock];
  __shared__ double gmat[256];
  int MP;
  reduce_gram_partials(ws, red, gmat, MP);
  if (threadIdx.x < 256) gmat_out[threadIdx.x] = (static_cast<int>(threadIdx.x) < MP * MP) ? gmat[threadIdx.x] : 0.0;
  if (threadIdx.x == 0) gmat_out[256] = static_cast<double>(MP);
}

__global__ __launch_bounds__(256) void svgd_kstats_gmat_kernel(const double* __restrict__ gmats, int n_mats,
                                                              int64_t mat_stride, int M, StatParams sp,
                                                              float* __restrict__ kstat) {
  __shared__ double gmat[256];
  const int MP = static_cast<int>(gmats[256]);
  if (static_cast<int>(threadIdx.x) < MP * MP) {
    double s = 0.0;
    for (int r = 0; r < n_mats; ++r) s += gmats[r * mat_stride + threadIdx.x];
    gmat[threadIdx.x] = s;
  }
  __syncthreads();
  svgd_stats_core(gmat, M, MP, sp, kstat, nullptr, nullptr);
}

// --------------------------------------------------------------- combine --
template <int M, bool HAS_G>
__global__ __launch_bounds__(kBlock) void svgd_combine_kernel(const float* __restrict__ P, const float* G, float* out,
                                                             int64_t D, int64_t ld, int64_t ldg,
                                                             const float* __restrict__ cgT,
                                                             const float* __restrict__ cpT) {
  const int64_t n4 = D >> 2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i4 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i4 < n4; i4 += stride) {
    f32x4 acc[M];
#pragma unroll
    for (int i = 0; i < M; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < M; ++j) {
      // streamed once: non-temporal loads/stores (+4 % measured)
      const f32x4 p = ld4_nt(P + j * ld + 4 * i4);
      if (HAS_G) {
        const f32x4 g = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(G + j * ldg + 4 * i4));
#pragma unroll
        for (int i = 0; i < M; ++i) {
          const float a = cgT[j * M + i];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[i][c] = __builtin_fmaf(a, g[c], acc[i][c]);
        }
      }
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const float b = cpT[j * M + i];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = __builtin_fmaf(b, p[c], acc[i][c]);
      }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) __builtin_nontemporal_store(acc[i], reinterpret_cast<f32x4*>(out + i * ld + 4 * i4));
  }
  if (blockIdx.x == 0) {
    const int64_t e = (n4 << 2) + threadIdx.x;
    if (e < D) {
      float acc[M];
#pragma unroll
      for (int i = 0; i < M; ++i) acc[i] = 0.f;
#pragma unroll
      for (int j = 0; j < M; ++j) {
        const float p = P[j * ld + e];
        const float g = HAS_G ? G[j * ldg + e] : 0.f;
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

// The same combine with the gradients read from the tensors autograd produced (svgd_shared.hpp: segments / chunks).
// One chunk = <= 256 float4 columns of one parameter tensor = one column per thread; chunk descriptor and the M
// gradient pointers of its segment are wave-uniform (scalar loads).
template <int M>
__global__ __launch_bounds__(kBlock) void svgd_combine_seg_kernel(const float* __restrict__ P,
                                                                 const float* const* __restrict__ seg_ptrs,
                                                                 const SegChunk* __restrict__ chunks, int n_chunks,
                                                                 float* out, int64_t ld,
                                                                 const float* __restrict__ cgT,
                                                                 const float* __restrict__ cpT) {
  SegChunk ch = {}, ch_next = {};
  if (static_cast<int>(blockIdx.x) < n_chunks) ch = chunks[blockIdx.x];
  for (int q = blockIdx.x; q < n_chunks; q += gridDim.x, ch = ch_next) {
    if (q + static_cast<int>(gridDim.x) < n_chunks) ch_next = chunks[q + gridDim.x];   // next descriptor in flight early
    const int valid = ch.nflt - 4 * static_cast<int>(threadIdx.x);
    if (valid <= 0) continue;
    const float* const* gp = seg_ptrs + static_cast<int64_t>(ch.seg) * M;
    const int64_t i4 = ch.c4 + threadIdx.x, l4 = ch.loc4 + threadIdx.x;
    f32x4 acc[M];
#pragma unroll
    for (int i = 0; i < M; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const f32x4 p = ld4_nt(P + j * ld + 4 * i4);
      const f32x4 g = seg_load(gp[j] + 4 * l4, valid);
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const float a = cgT[j * M + i];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = __builtin_fmaf(a, g[c], acc[i][c]);
      }
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const float b = cpT[j * M + i];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = __builtin_fmaf(b, p[c], acc[i][c]);
      }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) __builtin_nontemporal_store(acc[i], reinterpret_cast<f32x4*>(out + i * ld + 4 * i4));
  }
}

// Pack segmented gradients into flat rows: G[row, chunk columns] = the chunk of that row's gradient tensor.
__global__ __launch_bounds__(kBlock) void svgd_gather_seg_kernel(const float* const* __restrict__ seg_ptrs,
                                                                const SegChunk* __restrict__ chunks, int n_chunks,
                                                                float* G, int M, int row0, int n_rows, int64_t ld) {
  for (int q = blockIdx.x; q < n_chunks; q += gridDim.x) {
    const SegChunk ch = chunks[q];
    const int valid = ch.nflt - 4 * static_cast<int>(threadIdx.x);
    const float* const* gp = seg_ptrs + static_cast<int64_t>(ch.seg) * M;
    for (int j = row0; j < row0 + n_rows; ++j) {
      const float* src = gp[j] + 4 * ch.loc4;
      float* dst = G + j * ld + 4 * ch.c4;
      if (src == dst || valid <= 0) continue;             // the first test is wave-uniform: the piece already lives in G
      st4_nt(dst + 4 * threadIdx.x, seg_load(src + 4 * threadIdx.x, valid));
    }
  }
}

template <int M>
static int launch_combine(const float* P, const float* G, float* out, int64_t D, int64_t ld, int64_t ldg,
                          const float* kstat, hipStream_t s) {
  const int n = M * M;
  const float* cg = kstat + 2 * n + M + 4;
  const float* cp = cg + n;
  const int grid = stream_grid((D + 3) / 4);
  if (G)
    hipLaunchKernelGGL((svgd_combine_kernel<M, true>), dim3(grid), dim3(kBlock), 0, s, P, G, out, D, ld, ldg, cg, cp);
  else
    hipLaunchKernelGGL((svgd_combine_kernel<M, false>), dim3(grid), dim3(kBlock), 0, s, P, G, out, D, ld, ldg, cg, cp);
  return to_err(hipGetLastError());
}

// ------------------------------------------- fused shared-state optimizers --
// One thread owns a float4 column of all M particles and walks the particles
// in order, carrying the SHARED optimizer state in registers.
__global__ __launch_bounds__(kBlock) void svgd_apply_sgd_kernel(float* __restrict__ P, const float* __restrict__ grad,
                                                               float* __restrict__ buf, int M, int64_t D, int64_t ld,
                                                               SgdParams k) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < D; e += stride) {
    float b = (k.momentum != 0.f && !k.first) ? buf[e] : 0.f;
    for (int i = 0; i < M; ++i) P[i * ld + e] = sgd_apply(P[i * ld + e], grad[i * ld + e], b, k, i == 0);
    if (k.momentum != 0.f) buf[e] = b;
  }
}

__global__ __launch_bounds__(kBlock) void svgd_apply_adam_kernel(float* __restrict__ P, const float* __restrict__ grad,
                                                                float* __restrict__ exp_avg,
                                                                float* __restrict__ exp_avg_sq, int M, int64_t D,
                                                                int64_t ld, AdamParams k, AdamSteps st) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < D; e += stride) {
    float m = exp_avg[e], v = exp_avg_sq[e];
    for (int i = 0; i < M; ++i)
      P[i * ld + e] = adam_apply(P[i * ld + e], grad[i * ld + e], m, v, k, st.step_size[i], st.bc2_sqrt[i]);
    exp_avg[e] = m;
    exp_avg_sq[e] = v;
  }
}

// ===================== generic path: 16 < M <= 64 (blocked, several passes) =====================
// Not a tuned path (the reference's configs use 5 particles); it exists so that any particle_count the
// reference accepts up to 64 works.  Particles are split into groups of 8; one 16-row Gram tile per PAIR
// of groups (centred by the mean of the rows in the tile, which is all the distances need), reduced to a
// d2 [M, M] matrix; statistics as in the fast path; combine in chunks of 16 output rows.

// One workgroup per pair of groups: fp64 fixed-order reduction of that pair's partial tiles -> d2 entries.
__global__ __launch_bounds__(kStatsBlock) void svgd_pairs_to_d2_kernel(const float* __restrict__ ws, int M,
                                                                      float* __restrict__ d2mat) {
  __shared__ double red[kStatsBlock];
  __shared__ double gmat[256];
  // pair index -> (ga <= gb)
  const int ng = (M + 7) / 8;
  int pair = blockIdx.x, ga = 0;
  while (pair >= ng - ga) { pair -= ng - ga; ++ga; }
  const int gb = ga + pair;
  const int nb = static_cast<int>(ws[0]);
  const float* part = ws + kWsHeaderFloats + static_cast<int64_t>(blockIdx.x) * kGramMaxBlocks * 256;
  const int tid = threadIdx.x;
  {
    const int e = tid & 255, slice = tid >> 8;          // 4 slices
    double s = 0.0;
    for (int b = slice; b < nb; b += 4) s += static_cast<double>(part[static_cast<int64_t>(b) * 256 + e]);
    red[tid] = s;
  }
  __syncthreads();
  if (tid < 256) gmat[tid] = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
  __syncthreads();
  if (tid < 256) {
    const int ti = tid >> 4, tj = tid & 15;
    const int i = (ti < 8 ? ga * 8 + ti : gb * 8 + ti - 8), j = (tj < 8 ? ga * 8 + tj : gb * 8 + tj - 8);
    const bool vi = i < M && (ti < 8 || ga != gb), vj = j < M && (tj < 8 || ga != gb);
    // the diagonal blocks come from the (g, g) pairs, the cross blocks from the (ga < gb) pairs
    const bool mine = (ga == gb) ? (ti < 8 && tj < 8) : ((ti < 8) != (tj < 8));
    if (vi && vj && mine) {
      double d = gmat[ti * 16 + ti] + gmat[tj * 16 + tj] - 2.0 * gmat[ti * 16 + tj];
      if (d < 0.0 || i == j) d = 0.0;
      d2mat[i * M + j] = static_cast<float>(d);
    }
  }
}

__global__ __launch_bounds__(kStatsBlock) void svgd_kstats_generic_kernel(const float* __restrict__ d2mat, int M,
                                                                         float l2_reg, float kernel_grad_scale,
                                                                         float dataset_size, float sign,
                                                                         float h_override, float log_m1, int mode,
                                                                         float* __restrict__ kstat) {
  __shared__ float d2f[BDE_MAX_PARTICLES * BDE_MAX_PARTICLES];
  __shared__ float sorted[BDE_MAX_PARTICLES * BDE_MAX_PARTICLES];
  __shared__ float kmat[BDE_MAX_PARTICLES * BDE_MAX_PARTICLES];
  __shared__ float rowsum[BDE_MAX_PARTICLES];
  __shared__ float hs[2];
  const int tid = threadIdx.x, n = M * M;
  for (int e = tid; e < n; e += kStatsBlock) d2f[e] = d2mat[e];
  __syncthreads();
  for (int e = tid; e < n; e += kStatsBlock) {
    const float v = d2f[e];
    int rank = 0;
    for (int u = 0; u < n; ++u) {
      const float o = d2f[u];
      rank += (o < v || (o == v && u < e)) ? 1 : 0;
    }
    sorted[rank] = v;
  }
  __syncthreads();
  if (tid == 0) {
    const float pos = 0.5f * static_cast<float>(n - 1);
    const float lo = floorf(pos);
    const float wgt = pos - lo;
    const float a = sorted[static_cast<int>(lo)], b = sorted[static_cast<int>(ceilf(pos))];
    const float med = (fabsf(wgt) < 0.5f) ? a + wgt * (b - a) : b - (b - a) * (1.0f - wgt);
    float h = __builtin_sqrtf((0.5f * med) / log_m1) + 1e-8f;
    if (h_override > 0.f) h = h_override;
    hs[0] = h;
    hs[1] = med;
  }
  __syncthreads();
  const float h = hs[0];
  for (int e = tid; e < n; e += kStatsBlock) kmat[e] = expf(-d2f[e] / (2.0f * (h * h)));
  __syncthreads();
  if (tid < M) {
    float s = 0.f;
    for (int j = 0; j < M; ++j) s += kmat[tid * M + j];
    rowsum[tid] = s;
  }
  __syncthreads();
  const int oK = 0, oD2 = n, oRow = 2 * n, oMisc = 2 * n + M, oCG = oMisc + 4, oCP = oCG + n;
  const double h2 = static_cast<double>(h) * static_cast<double>(h);
  const double s_rep = static_cast<double>(kernel_grad_scale) / (static_cast<double>(dataset_size) * h2);
  for (int e = tid; e < n; e += kStatsBlock) {
    const int i = e / M, j = e % M;
    const double kij = kmat[e];
    const double rep = ((i == j) ? static_cast<double>(rowsum[i]) : 0.0) - kij;
    double cg, cp;
    if (mode == 0) {
      cg = static_cast<double>(sign) * (-kij);
      cp = static_cast<double>(sign) * (-kij * (0.5 * static_cast<double>(l2_reg)) + s_rep * rep);
    } else {
      cg = 0.0;
      cp = rep / h2;
    }
    kstat[oK + e] = kmat[e];
    kstat[oD2 + e] = d2f[e];
    kstat[oCG + j * M + i] = static_cast<float>(cg);
    kstat[oCP + j * M + i] = static_cast<float>(cp);
  }
  if (tid < M) kstat[oRow + tid] = rowsum[tid];
  if (tid == 0) {
    kstat[oMisc + 0] = h;
    kstat[oMisc + 1] = hs[1];
    kstat[oMisc + 2] = static_cast<float>(s_rep);
    kstat[oMisc + 3] = static_cast<float>(M);
  }
}

// 16 output rows [i0, i0 + mi) per launch, all M input rows streamed through.
template <bool HAS_G>
__global__ __launch_bounds__(kBlock) void svgd_combine_generic_kernel(const float* __restrict__ P,
                                                                     const float* __restrict__ G,
                                                                     float* __restrict__ out, int M, int i0, int mi,
                                                                     int64_t D, int64_t ld,
                                                                     const float* __restrict__ cgT,
                                                                     const float* __restrict__ cpT) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  const int64_t n4 = D >> 2;
  for (int64_t i4 = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i4 < n4 + 1; i4 += stride) {
    const bool tail = i4 == n4;                       // the D % 4 leftover coordinates: one thread, scalar
    if (tail && (D & 3) == 0) continue;
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < M; ++j) {
      f32x4 p = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
      if (!tail) {
        p = ld4_nt(P + j * ld + 4 * i4);
        if (HAS_G) g = ld4_nt(G + j * ld + 4 * i4);
      } else {
        for (int c = 0; c < static_cast<int>(D & 3); ++c) {
          p[c] = P[j * ld + 4 * i4 + c];
          if (HAS_G) g[c] = G[j * ld + 4 * i4 + c];
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i < mi) {                                 // wave-uniform
          const float a = HAS_G ? cgT[j * M + i0 + i] : 0.f, b = cpT[j * M + i0 + i];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[i][c] = __builtin_fmaf(b, p[c], HAS_G ? __builtin_fmaf(a, g[c], acc[i][c]) : acc[i][c]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i < mi) {
        if (!tail) {
          st4_nt(out + (i0 + i) * ld + 4 * i4, acc[i]);
        } else {
          for (int c = 0; c < static_cast<int>(D & 3); ++c) out[(i0 + i) * ld + 4 * i4 + c] = acc[i][c];
        }
      }
    }
  }
}

}  // namespace bde

using namespace bde;

extern "C" size_t bde_svgd_ws_bytes(int M) {
  if (M < 1 || M > BDE_MAX_PARTICLES) return 0;
  if (M > BDE_FAST_PARTICLES) return sizeof(float) * (svgd_generic_d2_offset(M) + static_cast<size_t>(M) * M);
  const size_t mp = (M <= 8) ? 8 : 16;
  return sizeof(float) * (kWsHeaderFloats + static_cast<size_t>(kGramMaxBlocks) * mp * mp);
}

extern "C" size_t bde_svgd_kstat_floats(int M) {
  if (M < 1 || M > BDE_MAX_PARTICLES) return 0;
  return static_cast<size_t>(4 * M * M + M + 4);
}

// Bytes of the Gram walk's tail that stay cacheable for the combine pass (default 240 MB of the 256 MB Infinity Cache);
// a tuning hook (tools/gram_split_ab.py: the A/B at M = 5, 8, 16), process-wide, < 0 restores the default.
static std::atomic<int64_t> g_gram_keep_bytes{240000000};
extern "C" int bde_svgd_set_gram_keep_bytes(int64_t bytes) {
  g_gram_keep_bytes.store(bytes < 0 ? 240000000 : bytes, std::memory_order_relaxed);
  return 0;
}
static inline int gram_split_now(int M, int64_t D) {
  return gram_nt_split(M, D, static_cast<double>(g_gram_keep_bytes.load(std::memory_order_relaxed)));
}

extern "C" int bde_svgd_gram(const float* P, int M, int64_t D, int64_t ld, void* ws, void* stream) {
  if (!svgd_args_ok(P, M, D, ld) || !ws || !aligned16(ws)) return BDE_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t n4 = (D + 3) / 4;
  float* wsf = static_cast<float*>(ws);
  if (M <= 8) {
    const int64_t tiles = (n4 + kGramU * 8 - 1) / (kGramU * 8);
    const int grid = static_cast<int>(std::min<int64_t>((tiles + 3) / 4, kGramMaxBlocks));
    hipLaunchKernelGGL(svgd_gram_kernel<2>, dim3(grid), dim3(kGramBlock), 0, s, P, M, D, ld, wsf,
                       GramRows{0, 0, 0, 0, 0, gram_split_now(M, D)});
    return to_err(hipGetLastError());
  }
  const int64_t tiles = (n4 + kGramU * 4 - 1) / (kGramU * 4);
  const int grid = static_cast<int>(std::min<int64_t>((tiles + 3) / 4, kGramMaxBlocks));
  if (M <= BDE_FAST_PARTICLES) {
    hipLaunchKernelGGL(svgd_gram_kernel<1>, dim3(grid), dim3(kGramBlock), 0, s, P, M, D, ld, wsf,
                       GramRows{0, 8, 8, M - 8, 0, gram_split_now(M, D)});
    return to_err(hipGetLastError());
  }
  // generic: one 16-row tile per pair of 8-particle groups, then the pairs' tiles -> d2 [M, M]
  const int ng = svgd_groups(M);
  int slot = 0;
  for (int ga = 0; ga < ng; ++ga) {
    for (int gb = ga; gb < ng; ++gb, ++slot) {
      const int nA = std::min(8, M - ga * 8), nB = (gb == ga) ? 0 : std::min(8, M - gb * 8);
      hipLaunchKernelGGL(svgd_gram_kernel<1>, dim3(grid), dim3(kGramBlock), 0, s, P, M, D, ld, wsf,
                         GramRows{ga * 8, nA, gb * 8, nB, slot});
    }
  }
  hipLaunchKernelGGL(svgd_pairs_to_d2_kernel, dim3(svgd_pairs(M)), dim3(kStatsBlock), 0, s, wsf, M,
                     wsf + svgd_generic_d2_offset(M));
  return to_err(hipGetLastError());
}

extern "C" int bde_svgd_kstats(const void* ws, int M, float l2_reg, float kernel_grad_scale, float dataset_size,
                               float sign, float h_override, int mode, float* kstat, void* stream) {
  if (!ws || !kstat || M < 1 || M > BDE_MAX_PARTICLES || (mode != 0 && mode != 1)) return BDE_ERR_INVALID;
  const float log_m1 = static_cast<float>(std::log(static_cast<double>(M) + 1.0));   // np.log(M + 1), svgd.py:18
  if (M > BDE_FAST_PARTICLES) {
    hipLaunchKernelGGL(svgd_kstats_generic_kernel, dim3(1), dim3(kStatsBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float*>(ws) + svgd_generic_d2_offset(M), M, l2_reg, kernel_grad_scale,
                       dataset_size, sign, h_override, log_m1, mode, kstat);
    return to_err(hipGetLastError());
  }
  hipLaunchKernelGGL(svgd_kstats_kernel, dim3(1), dim3(kStatsBlock), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float*>(ws), M,
                     StatParams{l2_reg, kernel_grad_scale, dataset_size, sign, h_override, log_m1, mode}, kstat);
  return to_err(hipGetLastError());
}

extern "C" int bde_svgd_gram_finish(const void* ws, int M, double* gmat_out, void* stream) {
  if (!ws || !gmat_out || M < 1 || M > BDE_FAST_PARTICLES) return BDE_ERR_INVALID;
  hipLaunchKernelGGL(svgd_gram_finish_kernel, dim3(1), dim3(kStatsBlock), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float*>(ws), gmat_out);
  return to_err(hipGetLastError());
}

extern "C" int bde_svgd_kstats_gmat(const double* gmats, int n_mats, int64_t mat_stride, int M, float l2_reg,
                                    float kernel_grad_scale, float dataset_size, float sign, float h_override, int mode,
                                    float* kstat, void* stream) {
  if (!gmats || n_mats < 1 || mat_stride < BDE_GMAT_DOUBLES || !kstat || M < 1 || M > BDE_FAST_PARTICLES ||
      (mode != 0 && mode != 1))
    return BDE_ERR_INVALID;
  const float log_m1 = static_cast<float>(std::log(static_cast<double>(M) + 1.0));
  hipLaunchKernelGGL(svgd_kstats_gmat_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), gmats, n_mats,
                     mat_stride, M, StatParams{l2_reg, kernel_grad_scale, dataset_size, sign, h_override, log_m1, mode},
                     kstat);
  return to_err(hipGetLastError());
}

extern "C" int bde_svgd_combine(const float* P, const float* G, float* out, int M, int64_t D, int64_t ld, int64_t ldg,
                                const float* kstat, void* stream) {
  if (ldg == 0) ldg = ld;
  if (!svgd_args_ok(P, M, D, ld) || !out || !kstat || !aligned16(out) || (G && !aligned16(G)) || out == P)
    return BDE_ERR_INVALID;
  if (G && (ldg < D || (ldg & 3) || (out == G && ldg != ld))) return BDE_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (M > BDE_FAST_PARTICLES) {
    if (out == G || ldg != ld) return BDE_ERR_INVALID;  // chunks of output rows re-read all of G
    const int n = M * M;
    const float* cg = kstat + 2 * n + M + 4;
    const float* cp = cg + n;
    const int grid = stream_grid((D + 3) / 4 + 1);
    for (int i0 = 0; i0 < M; i0 += 16) {
      const int mi = std::min(16, M - i0);
      if (G)
        hipLaunchKernelGGL(svgd_combine_generic_kernel<true>, dim3(grid), dim3(kBlock), 0, s, P, G, out, M, i0, mi, D,
                           ld, cg, cp);
      else
        hipLaunchKernelGGL(svgd_combine_generic_kernel<false>, dim3(grid), dim3(kBlock), 0, s, P, G, out, M, i0, mi, D,
                           ld, cg, cp);
    }
    return to_err(hipGetLastError());
  }
  switch (M) {
#define BDE_CASE(m) \
  case m:           \
    return launch_combine<m>(P, G, out, D, ld, ldg, kstat, s);
    BDE_CASE(1) BDE_CASE(2) BDE_CASE(3) BDE_CASE(4) BDE_CASE(5) BDE_CASE(6) BDE_CASE(7) BDE_CASE(8)
    BDE_CASE(9) BDE_CASE(10) BDE_CASE(11) BDE_CASE(12) BDE_CASE(13) BDE_CASE(14) BDE_CASE(15) BDE_CASE(16)
#undef BDE_CASE
  }
  return BDE_ERR_INVALID;
}

extern "C" int bde_svgd_combine_seg(const float* P, const void* const* seg_ptrs, const bde_seg_chunk* chunks,
                                    int64_t n_chunks, float* out, int M, int64_t D, int64_t ld, const float* kstat,
                                    void* stream) {
  if (!svgd_args_ok(P, M, D, ld) || M > BDE_FAST_PARTICLES || !seg_args_ok(seg_ptrs, chunks, n_chunks, D) || !out ||
      !kstat || !aligned16(out) || out == P)
    return BDE_ERR_INVALID;
  const int n = M * M;
  const float* cg = kstat + 2 * n + M + 4;
  const float* cp = cg + n;
  const int grid = static_cast<int>(std::min<int64_t>(n_chunks, kMaxStreamBlocks));
  const float* const* sp = reinterpret_cast<const float* const*>(seg_ptrs);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (M) {
#define BDE_CASE(m)                                                                                                    \
  case m:                                                                                                              \
    hipLaunchKernelGGL((svgd_combine_seg_kernel<m>), dim3(grid), dim3(kBlock), 0, s, P, sp, chunks,                      \
                       static_cast<int>(n_chunks), out, ld, cg, cp);                                                    \
    break;
    BDE_CASE(1) BDE_CASE(2) BDE_CASE(3) BDE_CASE(4) BDE_CASE(5) BDE_CASE(6) BDE_CASE(7) BDE_CASE(8)
    BDE_CASE(9) BDE_CASE(10) BDE_CASE(11) BDE_CASE(12) BDE_CASE(13) BDE_CASE(14) BDE_CASE(15) BDE_CASE(16)
#undef BDE_CASE
    default:
      return BDE_ERR_INVALID;
  }
  return to_err(hipGetLastError());
}

extern "C" int bde_svgd_gather_seg(const void* const* seg_ptrs, const bde_seg_chunk* chunks, int64_t n_chunks, float* G,
                                   int M, int row0, int n_rows, int64_t ld, void* stream) {
  if (!seg_ptrs || !chunks || n_chunks < 1 || n_chunks > (int64_t{1} << 30) || !G || !aligned16(G) || M < 1 ||
      M > BDE_MAX_PARTICLES || row0 < 0 || n_rows < 1 || row0 + n_rows > M || (ld & 3))
    return BDE_ERR_INVALID;
  const int grid = static_cast<int>(std::min<int64_t>(n_chunks, kMaxStreamBlocks));
  hipLaunchKernelGGL(svgd_gather_seg_kernel, dim3(grid), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float* const*>(seg_ptrs), chunks, static_cast<int>(n_chunks), G, M, row0,
                     n_rows, ld);
  return to_err(hipGetLastError());
}

namespace bde {
struct ScalarPtrs {
  const float* v[64];
};
// every lane fetches one scalar (all loads in flight at once), lane 0 adds them in index order; svgd.py:105
// `total_loss / particle_count` in the same launch, rounded as torch's GPU kernel for `tensor / python_number` rounds it
// (the sum TIMES fl(1 / divisor) -- not an IEEE division: the two differ by one ulp for e.g. the reference's 5 particles,
// iwildcam.yaml:218); divisor 1 leaves the sum as it is
__global__ __launch_bounds__(64) void sum_scalars_kernel(ScalarPtrs p, int n, float divisor, float* __restrict__ out) {
  __shared__ float vals[64];
  const int i = threadIdx.x;
  vals[i] = i < n ? *p.v[i] : 0.f;
  __syncthreads();
  if (i == 0) {
    float s = vals[0];
    for (int j = 1; j < n; ++j) s += vals[j];
    out[0] = divisor == 1.f ? s : s * (1.f / divisor);
  }
}
}  // namespace bde

extern "C" int bde_mean_scalars(const float* const* scalars, int n, float divisor, float* out, void* stream) {
  if (!scalars || !out || n < 1 || n > 64 || !(divisor > 0.f)) return BDE_ERR_INVALID;
  bde::ScalarPtrs p;
  for (int i = 0; i < 64; ++i) p.v[i] = i < n ? scalars[i] : nullptr;
  for (int i = 0; i < n; ++i)
    if (!p.v[i]) return BDE_ERR_INVALID;
  hipLaunchKernelGGL(bde::sum_scalars_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), p, n, divisor, out);
  return bde::to_err(hipGetLastError());
}

extern "C" int bde_sum_scalars(const float* const* scalars, int n, float* out, void* stream) {
  return bde_mean_scalars(scalars, n, 1.f, out, stream);
}

extern "C" int bde_svgd_step(const float* P, const float* G, float* out, int M, int64_t D, int64_t ld, float l2_reg,
                             float kernel_grad_scale, float dataset_size, float sign, void* ws, float* kstat,
                             void* stream) {
  // the three streaming stages at EVERY size: the small-model kernel (bde_svgd_step_small*) is the caller's explicit choice,
  // made by the host side only for sources whose parity tests have been green on a device (device_verified.py)
  if (!G) return BDE_ERR_INVALID;
  int rc = bde_svgd_gram(P, M, D, ld, ws, stream);
  if (rc) return rc;
  rc = bde_svgd_kstats(ws, M, l2_reg, kernel_grad_scale, dataset_size, sign, 0.f, 0, kstat, stream);
  if (rc) return rc;
  return bde_svgd_combine(P, G, out, M, D, ld, ld, kstat, stream);
}

extern "C" int bde_svgd_apply_sgd(float* P, const float* grad, float* momentum_buf, int M, int64_t D, int64_t ld,
                                  double lr, double momentum, double dampening, double weight_decay, int nesterov,
                                  int first, void* stream) {
  if (!svgd_args_ok(P, M, D, ld) || !grad || (momentum != 0.0 && !momentum_buf)) return BDE_ERR_INVALID;
  const SgdParams k{static_cast<float>(lr), static_cast<float>(momentum), static_cast<float>(1.0 - dampening),
                    static_cast<float>(weight_decay), nesterov, first};
  const int grid = stream_grid(D);
  hipLaunchKernelGGL(svgd_apply_sgd_kernel, dim3(grid), dim3(kBlock), 0, static_cast<hipStream_t>(stream), P, grad,
                     momentum_buf, M, D, ld, k);
  return to_err(hipGetLastError());
}

extern "C" int bde_svgd_apply_adam(float* P, const float* grad, float* exp_avg, float* exp_avg_sq, int M, int64_t D,
                                   int64_t ld, double lr, double beta1, double beta2, double eps, double weight_decay,
                                   int64_t step0, void* stream) {
  if (!svgd_args_ok(P, M, D, ld) || !grad || !exp_avg || !exp_avg_sq || step0 < 0) return BDE_ERR_INVALID;
  const AdamParams k{static_cast<float>(beta1), static_cast<float>(beta2), static_cast<float>(1.0 - beta1),
                     static_cast<float>(1.0 - beta2), static_cast<float>(eps), static_cast<float>(weight_decay)};
  const int grid = stream_grid(D);
  hipLaunchKernelGGL(svgd_apply_adam_kernel, dim3(grid), dim3(kBlock), 0, static_cast<hipStream_t>(stream), P, grad,
                     exp_avg, exp_avg_sq, M, D, ld, k, make_adam_steps(lr, beta1, beta2, step0));
  return to_err(hipGetLastError());
}

// bde_init(): load this translation unit's code object on the current device now (HIP otherwise uploads it at the
// first launch of one of its kernels).  Internal to the library (not exported).
extern "C" __attribute__((visibility("hidden"))) int bde_internal_load_svgd(void) {
  hipFuncAttributes attr;
  return bde::to_err(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&bde::svgd_kstats_kernel)));
}
