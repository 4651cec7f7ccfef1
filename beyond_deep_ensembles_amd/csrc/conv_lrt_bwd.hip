// Backward of the local-reparameterisation CONVOLUTION layer (BBBConv2d, bbb_layers.py:146-154), weight side:
//
//   dW_mu [o, c, r, q] = sum_{n, ho, wo} g   [n, o, ho, wo] * x          [n, c, ho s - p + r, wo s - p + q]
//   dS2   [o, c, r, q] = sum_{n, ho, wo} gvar[n, o, ho, wo] * clamp(x^2) [n, c, ...]              (0 in the padding)
//   dW_rho = dS2 * [softplus(rho)^2 >= 1e-4] * 2 softplus(rho) sigmoid(rho)
//
// (g = gradient of the layer output, gvar = g eps / (2 sqrt(var)) from the first pass of the backward, conv_lrt_gvar_kernel
// below, which also delivers the bias gradients) -- what autograd computes with two weight-gradient convolutions plus the
// element-wise chain in the reference.  One implicit GEMM with two accumulators: rows = output channels, columns =
// (c, r, q), reduction over the output PIXELS.  A workgroup owns a [MF rows x CT * MF columns] block of both gradient
// matrices (one instantiation per CT) and a share of the (image group, row band) items -- shares sized to ONE resident set of
// workgroups; per item it stages the input patch of the block's channels once as pairs (x, clamp(x^2)), the block's rows of
// g and gvar as pairs, and a pixel -> patch-offset table (flat over the lanes, 8-16 loads in flight: conv_common.hpp); its four
// waves take the k-steps (2 or 4 pixels each) round robin with two alternating operand sets, are summed through LDS as a
// fixed two-round tree, and the block goes to a partials buffer [share][2][O][C KH KW]; the finish pass (16 waves per 64
// elements, eight shares in flight per wave) adds the shares in a fixed order (bit-reproducible) and applies the chain
// rule for rho.
#include "conv_common.hpp"
#include <array>
#include <map>
#include <mutex>
#include <vector>

namespace bde {

using f32x16w = __attribute__((ext_vector_type(16))) float;
using f32x4w = __attribute__((ext_vector_type(4))) float;

struct WgGeo {
  int N, C, H, W, O, KH, KW, sh, sw, ph, pw, Ho, Wo;
};
constexpr int kWgTablePad = 48;   // zero entries behind the pixel table (the product loop looks up to 12 k-steps' worth ahead)

struct WgTile {
  int NI, TH, bands, PS, CT, colgroups, PH, PWP, cmax, GP, npix;
  float rcp_pwp;     // fl(1 / PWP) for the flat patch staging (conv_common.hpp)
};

template <int MF> struct MfmaW;
template <> struct MfmaW<32> {
  using Acc = f32x16w;
  static constexpr int KS = 2, REGS = 16;
  __device__ __forceinline__ static Acc run(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
  __device__ __forceinline__ static int row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
};
template <> struct MfmaW<16> {
  using Acc = f32x4w;
  static constexpr int KS = 4, REGS = 4;
  __device__ __forceinline__ static Acc run(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  __device__ __forceinline__ static int row(int r, int h) { return 4 * h + r; }
};

// LDS: xq [NI][cmax][PH][PWP] pairs (x, clamp(x^2)) | gq [MF][GP] pairs (g, g_var) | pixtab [npix + 48] (byte offsets)   (reused for the wave reduction)
template <int MF, int CT>
__global__ __launch_bounds__(256, 2) void conv_lrt_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                const float* __restrict__ gvar, float* __restrict__ part,
                                                                WgGeo geo, WgTile t) {
  using M = MfmaW<MF>;
  using Acc = typename M::Acc;
  constexpr int KS = M::KS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int row_elems = t.PH * t.PWP;
  const int khw = geo.KH * geo.KW, ktot = geo.C * khw;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane / MF, idx = lane % MF;
  // work index -> (share of the items, block of the gradient matrices), block fastest: the blocks of one share stage the same
  // patches and g rows and run on one XCD back to back (xcd_work_index)
  const int work = xcd_work_index(static_cast<int>(blockIdx.y * gridDim.x + blockIdx.x), static_cast<int>(gridDim.x * gridDim.y));
  const int share = work / static_cast<int>(gridDim.y), yb = work % static_cast<int>(gridDim.y);
  const int otile = yb / t.colgroups, cg = yb % t.colgroups;
  const int o0 = otile * MF, col0 = cg * CT * MF;                // (t.CT == CT: the host instantiates the planner's choice)
  const int col_end = min(ktot, col0 + CT * MF);
  const int c_lo = col0 / khw, c_hi = (col_end - 1) / khw, cc = c_hi - c_lo + 1;
  const int img_floats = cc * row_elems;
  const int patch_floats = t.NI * t.cmax * row_elems;
  f32x2* xq = reinterpret_cast<f32x2*>(lds);                       // pairs (x, clamp(x^2)): one b64 read serves both products
  f32x2* gq = reinterpret_cast<f32x2*>(lds + 2 * patch_floats);    // pairs (g, g_var) [MF][GP]
  int* pixtab = reinterpret_cast<int*>(lds + 2 * patch_floats + 2 * MF * t.GP);

  int kofs[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = col0 + ct * MF + idx;
    kofs[ct] = 0;
    if (col < ktot) {                                            // a column past the matrix multiplies patch element 0: never stored
      const int c = col / khw - c_lo, rq = col % khw;
      kofs[ct] = 8 * (c * row_elems + (rq / geo.KW) * t.PWP + (rq % geo.KW));    // bytes of pairs, like the pixel table's entries
    }
  }
  Acc accm[CT], accv[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) accm[ct] = accv[ct] = Acc{};

  const int bpi = t.TH * geo.Wo;                               // band pixels per image
  const int64_t howo = static_cast<int64_t>(geo.Ho) * geo.Wo;
  const int items = ((geo.N + t.NI - 1) / t.NI) * t.bands;
  for (int item = share; item < items; item += t.PS) {
    const int img0 = (item / t.bands) * t.NI, band = item % t.bands;
    const int ho0 = band * t.TH;
    const int th = min(t.TH, geo.Ho - ho0);
    const int hi0 = ho0 * geo.sh - geo.ph;
    __syncthreads();
    // the input patch of the block's channels as pairs (x, clamp(x^2)): flat over the lanes, eight loads in flight per lane
    // (conv_common.hpp)
    constexpr int kDepth = CT * M::REGS <= 48 ? 16 : 8;           // sixteen items in flight where the accumulators leave the registers
    conv_stage_patch<0, false, kDepth>(x, x, xq, wave, lane, t.NI * cc, cc, row_elems, t.PWP, t.rcp_pwp, img_floats, img0, geo.N, geo.C, c_lo,
                               geo.H, geo.W, hi0, geo.pw, 1, 1);
    // the block's rows of g and gvar ((row, image) strips of contiguous band pixels) as pairs: flat items, sixteen loads in flight
    conv_stage_rows<kDepth>(g, gvar, gq, wave, lane, MF * t.NI, t.NI, bpi, th * geo.Wo, t.GP, img0, geo.N, o0, geo.O, howo,
                    static_cast<int64_t>(ho0) * geo.Wo);
    const int padn = t.npix - t.NI * bpi;                      // the padding behind the last image's strip
    for (int e = threadIdx.x; e < MF * padn; e += 256) {
      const int o = e / padn, pp = t.NI * bpi + e % padn;
      gq[o * t.GP + pp] = f32x2{0.f, 0.f};
    }
    for (int pp = threadIdx.x; pp < t.npix + kWgTablePad; pp += 256) {   // (zero entries behind the table: read ahead, never used)
      const int img = pp / bpi, p = pp % bpi;
      const int hl = p / geo.Wo, wo = p % geo.Wo;
      pixtab[pp] = (pp < t.npix && img < t.NI && hl < th) ? 8 * (img * img_floats + hl * geo.sh * t.PWP + wo * geo.sw) : 0;
    }
    __syncthreads();
    const int ksteps = t.npix / KS;
    // Two operand sets, used alternately (no register copies): the operands of this wave's NEXT k-step are requested before
    // the products of the current one are issued -- the LDS round trip hides behind 2 CT MFMAs instead of stalling in front
    // of them.  One ds_read_b64 per operand pair: 2 + CT reads (+ the pixel's patch offset) per 2 CT products.
    // The pixel -> patch-offset entry of a step is read one step BEFORE its operands are requested (po_next): an operand request
    // never waits for the LDS round trip of its own address.
    f32x2 a0, a1, b0[CT], b1[CT];
    auto fetch = [&](int ks_, int po, f32x2& a, f32x2(&b)[CT]) {
      a = gq[idx * t.GP + ks_ * KS + h];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) b[ct] = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(xq) + (po + kofs[ct]));
    };
    auto products = [&](const f32x2& a, const f32x2(&b)[CT]) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        accm[ct] = M::run(a.x, b[ct].x, accm[ct]);
        accv[ct] = M::run(a.y, b[ct].y, accv[ct]);
      }
    };
    int ks = wave;                                             // this wave's steps: wave, wave + 4, ...
    int po_next = 0;
    if (ks < ksteps) {
      fetch(ks, pixtab[ks * KS + h], a0, b0);
      po_next = pixtab[(ks + 4) * KS + h];
    }
    for (; ks + 4 < ksteps; ks += 8) {
      fetch(ks + 4, po_next, a1, b1);
      po_next = pixtab[(ks + 8) * KS + h];
      products(a0, b0);
      if (ks + 8 < ksteps) {
        fetch(ks + 8, po_next, a0, b0);
        po_next = pixtab[(ks + 12) * KS + h];
      }
      products(a1, b1);
    }
    if (ks < ksteps) products(a0, b0);                         // an odd number of steps: the last set fetched is still pending
  }

  // ---- the four waves' blocks summed through LDS as a fixed two-round tree, (w0 + w1) + (w2 + w3): round 1 waves 1 and 3
  //      hand their blocks to waves 0 and 2, round 2 wave 2 hands its sum to wave 0 (two blocks of CT tiles in LDS at a time:
  //      the planner reserves them).  Fixed order: bit-reproducible.
  float* red = lds;
  constexpr int per_wave = CT * 2 * M::REGS * 64;
#pragma unroll 1
  for (int round = 0; round < 2; ++round) {
    const bool writer = round == 0 ? (wave & 1) == 1 : wave == 2;
    const bool reader = round == 0 ? (wave & 1) == 0 : wave == 0;
    float* slot = red + (round == 0 ? (wave >> 1) : 0) * per_wave;     // round 1: pair (0, 1) -> slot 0, pair (2, 3) -> slot 1
    __syncthreads();                                             // the operand reads (round 2: the readers of round 1) are done
    if (writer) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int r = 0; r < M::REGS; ++r) {
          slot[((ct * 2 + 0) * M::REGS + r) * 64 + lane] = accm[ct][r];
          slot[((ct * 2 + 1) * M::REGS + r) * 64 + lane] = accv[ct][r];
        }
      }
    }
    __syncthreads();
    if (reader) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int r = 0; r < M::REGS; ++r) {
          accm[ct][r] += slot[((ct * 2 + 0) * M::REGS + r) * 64 + lane];
          accv[ct][r] += slot[((ct * 2 + 1) * M::REGS + r) * 64 + lane];
        }
      }
    }
  }
  if (wave != 0) return;
  // part [share][2][O][ktot]
  float* pm = part + static_cast<int64_t>(share) * 2 * geo.O * ktot;
  float* pv = pm + static_cast<int64_t>(geo.O) * ktot;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = col0 + ct * MF + idx;
    if (col < ktot) {
#pragma unroll
      for (int r = 0; r < M::REGS; ++r) {
        const int o = o0 + M::row(r, h);
        if (o < geo.O) {
          pm[static_cast<int64_t>(o) * ktot + col] = accm[ct][r];
          pv[static_cast<int64_t>(o) * ktot + col] = accv[ct][r];
        }
      }
    }
  }
}

// The per-share partial blocks summed in a FIXED order, then the chain rule of sigma^2 = clamp(softplus(rho)^2, 1e-4) for the rho
// gradient.  A workgroup = 16 waves x 64 consecutive elements: wave w adds the shares w, w + 16, w + 32, ... of its 64 elements
// in that order with eight shares (sixteen 4-byte loads, 256 contiguous bytes per wave each) in flight, the sixteen wave sums
// are added in wave order through LDS.  (Round 4: one thread per element walking ALL shares, two loads and a full memory round
// trip per share -- up to 768 dependent round trips, by the latency count several hundred microseconds for a 5 us job.)
constexpr int kFinWaves = 16;
__global__ __launch_bounds__(kFinWaves * 64) void conv_lrt_wgrad_finish_kernel(const float* __restrict__ part, int shares, int64_t n,
                                                                            const float* __restrict__ w_rho,
                                                                            float* __restrict__ g_wmu, float* __restrict__ g_wrho) {
  __shared__ float red[2][kFinWaves][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 64 + lane;
  const bool in = i < n;
  const int64_t ii = in ? i : 0;                               // (a lane past the end reads element 0 and drops it)
  float sm = 0.f, sv = 0.f;
  for (int s0 = wave; s0 < shares; s0 += 8 * kFinWaves) {
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + u * kFinWaves;
      const int64_t sc = s < shares ? s : 0;                   // wave-uniform; a share past the end: share 0, dropped below
      a[u] = part[(sc * 2 + 0) * n + ii];
      b[u] = part[(sc * 2 + 1) * n + ii];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (s0 + u * kFinWaves < shares) {
        sm += a[u];
        sv += b[u];
      }
    }
  }
  red[0][wave][lane] = sm;
  red[1][wave][lane] = sv;
  __syncthreads();
  if (wave == 0 && in) {
    float m = red[0][0][lane], v = red[1][0][lane];
#pragma unroll
    for (int w = 1; w < kFinWaves; ++w) {
      m += red[0][w][lane];
      v += red[1][w][lane];
    }
    g_wmu[i] = m;
    const SoftplusSigmoid ss = softplus_sigmoid(w_rho[i]);
    const float s2 = ss.sp * ss.sp;
    g_wrho[i] = s2 >= 1e-4f ? v * (2.0f * ss.sp * ss.sg) : 0.f;
  }
}

// The first pass of the layer's backward: g_var = g eps / (2 sqrt(var)) over the layer output [N, O, HW] (bbb_layers.py:148-154
// through autograd; the noise is the forward's: supplied, or the Philox stream of the flat output's float4 group e >> 2 -- the
// numbering of bde_local_reparam_bwd, which this pass replaces for the convolution layer) AND, riding on the same pass, the
// channel sums the bias gradients need: sum g and sum g_var per output channel (two torch reductions over the whole
// gradient + a kernel for the rho chain rule in round 4: two more passes over 2 x N O HW floats and three launches).
// grid (O, NCH): workgroup (o, ch) walks the planes (n, o) of the images n = ch, ch + NCH, ... in order; thread sums in fp32
// in a fixed order, workgroup sums in fp64 in a fixed order (block_sum) -> part [2][NCH][O] doubles; the finish pass adds the
// NCH partials of a channel in order and applies d softplus(b_rho)^2 / d b_rho (the bias variance is not clamped: line 147).
template <bool RNG>
__global__ __launch_bounds__(kBlock) void conv_lrt_gvar_kernel(const float* __restrict__ g, const float* __restrict__ var,
                                                              const float* __restrict__ eps, uint64_t seed, uint64_t stream_id,
                                                              float* __restrict__ gvar, double* __restrict__ part, int N, int O,
                                                              int64_t HW, int NCH, int vec) {
  __shared__ double smem[kBlock / 64];
  const int o = blockIdx.x, ch = blockIdx.y;
  float sm = 0.f, sv = 0.f;
  for (int n = ch; n < N; n += NCH) {
    const int64_t base = (static_cast<int64_t>(n) * O + o) * HW;
    if (vec) {                                                   // HW % 4 == 0 and 16-byte aligned tensors: float4 groups stay inside a plane
      for (int64_t i = threadIdx.x; i < (HW >> 2); i += kBlock) {
        const int64_t e = base + 4 * i;
        const f32x4 go = ld4_nt(g + e), v = ld4_nt(var + e);
        const f32x4 z = RNG ? philox_normal4(seed, stream_id, static_cast<uint64_t>(e >> 2), kDomainDiag) : ld4_nt(eps + e);
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          r[j] = (go[j] * z[j]) / (2.0f * __builtin_sqrtf(v[j]));
          sm += go[j];
          sv += r[j];
        }
        st4_nt(gvar + e, r);
      }
    } else {
      for (int64_t i = threadIdx.x; i < HW; i += kBlock) {
        const int64_t e = base + i;
        float z;
        if (RNG) {
          const f32x4 zz = philox_normal4(seed, stream_id, static_cast<uint64_t>(e >> 2), kDomainDiag);
          const int c = static_cast<int>(e & 3);
          z = c == 0 ? zz.x : c == 1 ? zz.y : c == 2 ? zz.z : zz.w;
        } else {
          z = eps[e];
        }
        const float go = g[e];
        const float r = (go * z) / (2.0f * __builtin_sqrtf(var[e]));
        gvar[e] = r;
        sm += go;
        sv += r;
      }
    }
  }
  if (part) {
    const double tm = block_sum(static_cast<double>(sm), smem);
    const double tv = block_sum(static_cast<double>(sv), smem);
    if (threadIdx.x == 0) {
      part[(0 * static_cast<int64_t>(NCH) + ch) * O + o] = tm;
      part[(1 * static_cast<int64_t>(NCH) + ch) * O + o] = tv;
    }
  }
}

__global__ __launch_bounds__(kBlock) void conv_lrt_bias_finish_kernel(const double* __restrict__ part, int NCH, int O,
                                                                     const float* __restrict__ b_rho,
                                                                     float* __restrict__ g_bmu, float* __restrict__ g_brho) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= O) return;
  double tm = 0.0, tv = 0.0;
  for (int ch = 0; ch < NCH; ++ch) {                            // (NCH <= 64 values of 8 bytes per channel: one round of loads)
    tm += part[(0 * static_cast<int64_t>(NCH) + ch) * O + o];
    tv += part[(1 * static_cast<int64_t>(NCH) + ch) * O + o];
  }
  g_bmu[o] = static_cast<float>(tm);
  const SoftplusSigmoid ss = softplus_sigmoid(b_rho[o]);
  g_brho[o] = static_cast<float>(tv) * (2.0f * ss.sp * ss.sg);
}

}  // namespace bde

using namespace bde;

namespace {

struct WgPlan {
  WgTile t;
  int mf, otiles;
  dim3 grid;
  size_t lds;
};

// Candidate tilings: CT column tiles of the [O, C*KH*KW] gradient per workgroup (as many as the accumulators hold, fewer when
// the input patches of the channels they span do not fit the LDS even for one output row: wide 1x1 layers on large images;
// the tiles spread evenly over the column groups: 9 tiles with at most 4 per group run as 3 + 3 + 3, not 4 + 4 + 1), a band
// of TH output rows (halving from the whole image) of NI images per item, PS shares of the items.  Shares: ONE resident set
// of workgroups (256 CUs x 2 per CU = 512 slots; round 4 asked for 768, i.e. a second, half empty wave of workgroups and
// half as many more partial blocks to write and re-read), every share the same number of items (+- 1).  The planner takes
// the best score among the LARGEST feasible CT unless a tiling is pinned (bde_conv_lrt_wgrad_set_tiling).
struct WgCand {
  WgTile t;
  size_t lds;
  double score;
  int ct_cap;
};

static void wgrad_candidates(const WgGeo& g, std::vector<WgCand>& out) {
  const int mf = g.O <= 16 ? 16 : 32;
  const int ct_max = mf == 16 ? 9 : 4;
  const int ks = mf == 32 ? 2 : 4;
  const int regs = mf == 32 ? 16 : 4;
  const int khw = g.KH * g.KW, ktot = g.C * khw;
  const int otiles = (g.O + mf - 1) / mf;
  const int coltiles = (ktot + mf - 1) / mf;
  int last_ct = -1;
  for (int ct_cap = std::min(ct_max, coltiles); ct_cap >= 1; --ct_cap) {
    const int colgroups = (coltiles + ct_cap - 1) / ct_cap;
    const int ct = (coltiles + colgroups - 1) / colgroups;
    if (ct == last_ct) continue;                                // the same spread as a larger cap
    last_ct = ct;
    const int cmax = std::min(g.C, (ct * mf + khw - 2) / khw + 1);
    for (int th = g.Ho; th >= 1; th = (th > 1 ? (th + 1) / 2 : 0)) {
      const int bands = (g.Ho + th - 1) / th;
      for (int ni = 8; ni >= 1; ni /= 2) {
        if (ni > 1 && th != g.Ho) continue;
        const int ph = (th - 1) * g.sh + g.KH, pwp = (g.Wo - 1) * g.sw + g.KW;
        int npix = ni * th * g.Wo;
        npix = (npix + ks * 4 - 1) / (ks * 4) * (ks * 4);
        int gp = npix;
        if (mf == 32) gp |= 1; else gp = (gp + 31) / 32 * 32 + 2;
        const size_t lds = sizeof(float) * (2ull * ni * cmax * ph * pwp + 2ull * mf * gp + npix + kWgTablePad);
        const size_t red = sizeof(float) * 2ull * ct * 2 * regs * 64;
        if (std::max(lds, red) > 64 * 1024) continue;
        const int items = ((g.N + ni - 1) / ni) * bands;
        const int blocks = otiles * colgroups;
        const int ps_cap = std::max(1, std::min(items, 512 / blocks));
        const int per_share = (items + ps_cap - 1) / ps_cap;
        const int ps = (items + per_share - 1) / per_share;
        const double fill = std::min(1.0, static_cast<double>(ps) * blocks / 512.0);
        const double halo = static_cast<double>(th) / ph;
        const double big = std::min(1.0, static_cast<double>(npix) / 256.0);      // enough k-steps per staging
        const double score = fill * (0.5 + 0.5 * halo) * (0.5 + 0.5 * big);
        out.push_back(WgCand{WgTile{ni, th, bands, ps, ct, colgroups, ph, pwp, cmax, gp, npix, 1.0f / static_cast<float>(pwp)}, std::max(lds, red), score, ct_cap});
      }
    }
  }
}

using WgKey = std::array<int, 11>;
static WgKey wg_key(const WgGeo& g) { return WgKey{g.N, g.C, g.H, g.W, g.O, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw}; }
static std::mutex& wg_pin_mutex() {
  static std::mutex m;
  return m;
}
static std::map<WgKey, std::array<int, 4>>& wg_pins() {               // (CT, TH, NI, PS)
  static std::map<WgKey, std::array<int, 4>> m;
  return m;
}

static bool plan_wgrad(const WgGeo& g, WgPlan& p, int* chosen_index = nullptr) {
  std::vector<WgCand> cands;
  wgrad_candidates(g, cands);
  if (cands.empty()) return false;
  int best = -1;
  {
    std::lock_guard<std::mutex> lock(wg_pin_mutex());
    const auto it = wg_pins().find(wg_key(g));
    if (it != wg_pins().end())
      for (size_t i = 0; i < cands.size(); ++i) {
        const WgTile& t = cands[i].t;
        if (t.CT == it->second[0] && t.TH == it->second[1] && t.NI == it->second[2] && t.PS == it->second[3]) best = static_cast<int>(i);
      }
  }
  if (best < 0) {
    // the largest CT that has a tiling at all (fewer column groups = fewer re-stagings of g / g_var), the best score within it
    const int ct_first = cands[0].t.CT;
    best = 0;
    for (size_t i = 1; i < cands.size() && cands[i].t.CT == ct_first; ++i)
      if (cands[i].score > cands[best].score) best = static_cast<int>(i);
  }
  const int mf = g.O <= 16 ? 16 : 32;
  p.t = cands[best].t;
  p.mf = mf;
  p.otiles = (g.O + mf - 1) / mf;
  p.grid = dim3(static_cast<unsigned>(p.t.PS), static_cast<unsigned>(p.otiles * p.t.colgroups));
  p.lds = cands[best].lds;
  if (chosen_index) *chosen_index = best;
  return true;
}

static bool geo_ok(int N, int C, int H, int W, int O, int KH, int KW, int sh, int sw, int ph, int pw, WgGeo& g) {
  if (N < 1 || C < 1 || H < 1 || W < 1 || O < 1 || KH < 1 || KW < 1 || KH > 7 || KW > 7 || sh < 1 || sw < 1 || ph < 0 || pw < 0)
    return false;
  const int Ho = (H + 2 * ph - KH) / sh + 1, Wo = (W + 2 * pw - KW) / sw + 1;
  if (Ho < 1 || Wo < 1) return false;
  g = WgGeo{N, C, H, W, O, KH, KW, sh, sw, ph, pw, Ho, Wo};
  return true;
}

}  // namespace

// The tiling of the weight-gradient kernel (tests / tools).  out[16]: MF, CT_MAX, NI, TH, bands, PS, CT, colgroups, PH, PWP,
// cmax, GP, npix, grid.y, LDS bytes, otiles.
extern "C" int bde_conv_lrt_bwd_weight_plan(int N, int C, int H, int W, int O, int KH, int KW, int sh, int sw, int ph, int pw,
                                            int* out) {
  WgGeo g;
  WgPlan p;
  if (!out || !geo_ok(N, C, H, W, O, KH, KW, sh, sw, ph, pw, g) || !plan_wgrad(g, p)) return BDE_ERR_INVALID;
  const int v[16] = {p.mf, p.mf == 16 ? 9 : 4, p.t.NI, p.t.TH, p.t.bands, p.t.PS, p.t.CT, p.t.colgroups, p.t.PH, p.t.PWP,
                     p.t.cmax, p.t.GP, p.t.npix, static_cast<int>(p.grid.y), static_cast<int>(p.lds), p.otiles};
  for (int i = 0; i < 16; ++i) out[i] = v[i];
  return 0;
}

// Tuning hooks of the weight-gradient pass (tools/conv_autotune.py): the candidate tilings of a LAYER geometry
// (layer[11] = N, C, H, W, O, KH, KW, sh, sw, ph, pw): out[i][5] = CT, TH, NI, PS, LDS bytes; *chosen = the planner's index.
extern "C" int bde_conv_lrt_wgrad_candidates(const int* layer, int* out, int max, int* chosen) {
  WgGeo g;
  if (!layer || (max > 0 && !out) ||
      !geo_ok(layer[0], layer[1], layer[2], layer[3], layer[4], layer[5], layer[6], layer[7], layer[8], layer[9], layer[10], g))
    return BDE_ERR_INVALID;
  std::vector<WgCand> cands;
  wgrad_candidates(g, cands);
  for (int i = 0; i < static_cast<int>(cands.size()) && i < max; ++i) {
    const WgTile& t = cands[i].t;
    const int v[5] = {t.CT, t.TH, t.NI, t.PS, static_cast<int>(cands[i].lds)};
    for (int j = 0; j < 5; ++j) out[5 * i + j] = v[j];
  }
  if (chosen) {
    WgPlan p;
    int idx = -1;
    *chosen = plan_wgrad(g, p, &idx) ? idx : -1;
  }
  return static_cast<int>(cands.size());
}

// Pin (CT, TH, NI, PS) for a layer geometry (process-wide); ct = 0 removes the pin; a tiling that is not a candidate is refused.
// The caller's partials buffer must be sized AFTER pinning (bde_conv_lrt_bwd_weight_ws_bytes follows the pin).
extern "C" int bde_conv_lrt_wgrad_set_tiling(const int* layer, int ct, int th, int ni, int ps) {
  WgGeo g;
  if (!layer || !geo_ok(layer[0], layer[1], layer[2], layer[3], layer[4], layer[5], layer[6], layer[7], layer[8], layer[9], layer[10], g))
    return BDE_ERR_INVALID;
  std::lock_guard<std::mutex> lock(wg_pin_mutex());
  if (ct == 0) {
    wg_pins().erase(wg_key(g));
    return 0;
  }
  std::vector<WgCand> cands;
  wgrad_candidates(g, cands);
  for (const WgCand& c : cands)
    if (c.t.CT == ct && c.t.TH == th && c.t.NI == ni && c.t.PS == ps) {
      wg_pins()[wg_key(g)] = {ct, th, ni, ps};
      return 0;
    }
  return BDE_ERR_INVALID;
}

static int gvar_chunks(int N, int O) {
  // about 1024 workgroups (4 per CU), at most 64 partials per channel, never more than N
  return std::max(1, std::min(std::min(N, 64), (1024 + O - 1) / O));
}

extern "C" size_t bde_conv_lrt_gvar_ws_bytes(int N, int O) {
  if (N < 1 || O < 1) return 0;
  return sizeof(double) * 2 * static_cast<size_t>(gvar_chunks(N, O)) * O;
}

// g_var of BBBConv2d's backward + (b_rho != NULL) the bias gradients g_bmu / g_brho, one pass over the output gradient.
extern "C" int bde_conv_lrt_gvar_bias(const float* g, const float* var, const float* eps, uint64_t seed, uint64_t stream_id,
                                      float* gvar, const float* b_rho, float* g_bmu, float* g_brho, void* ws, int N, int O,
                                      int64_t HW, void* stream) {
  if (!g || !var || !gvar || N < 1 || O < 1 || HW < 1) return BDE_ERR_INVALID;
  if (b_rho && (!g_bmu || !g_brho || !ws || (reinterpret_cast<uintptr_t>(ws) & 7))) return BDE_ERR_INVALID;
  const int nch = gvar_chunks(N, O);
  const int vec = (HW & 3) == 0 && aligned16(g) && aligned16(var) && aligned16(gvar) && (!eps || aligned16(eps));
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* part = b_rho ? static_cast<double*>(ws) : nullptr;
  const dim3 grid(static_cast<unsigned>(O), static_cast<unsigned>(nch));
  if (eps)
    hipLaunchKernelGGL(conv_lrt_gvar_kernel<false>, grid, dim3(kBlock), 0, s, g, var, eps, seed, stream_id, gvar, part, N, O, HW, nch, vec);
  else
    hipLaunchKernelGGL(conv_lrt_gvar_kernel<true>, grid, dim3(kBlock), 0, s, g, var, eps, seed, stream_id, gvar, part, N, O, HW, nch, vec);
  if (b_rho)
    hipLaunchKernelGGL(conv_lrt_bias_finish_kernel, dim3((O + kBlock - 1) / kBlock), dim3(kBlock), 0, s, part, nch, O, b_rho, g_bmu,
                       g_brho);
  return to_err(hipGetLastError());
}

// bytes of the partials buffer bde_conv_lrt_bwd_weight needs (0: unsupported geometry)
extern "C" size_t bde_conv_lrt_bwd_weight_ws_bytes(int N, int C, int H, int W, int O, int KH, int KW, int sh, int sw, int ph,
                                                   int pw) {
  WgGeo g;
  WgPlan p;
  if (!geo_ok(N, C, H, W, O, KH, KW, sh, sw, ph, pw, g) || !plan_wgrad(g, p)) return 0;
  return sizeof(float) * static_cast<size_t>(p.t.PS) * 2 * O * C * KH * KW;
}

extern "C" int bde_conv_lrt_bwd_weight(const float* x, const float* g, const float* gvar, const float* w_rho, void* ws,
                                       size_t ws_bytes, float* g_wmu, float* g_wrho, int N, int C, int H, int W, int O, int KH,
                                       int KW, int sh, int sw, int ph, int pw, void* stream) {
  WgGeo geo;
  WgPlan p;
  if (!x || !g || !gvar || !w_rho || !ws || !g_wmu || !g_wrho || !geo_ok(N, C, H, W, O, KH, KW, sh, sw, ph, pw, geo) ||
      !plan_wgrad(geo, p))
    return BDE_ERR_INVALID;
  // the plan of THIS launch (a tiling may have been pinned since the caller sized `ws`) must fit the caller's buffer
  if (ws_bytes < sizeof(float) * static_cast<size_t>(p.t.PS) * 2 * O * C * KH * KW) return BDE_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(ws);
  // one instantiation per (tile size, column tiles per workgroup): the product loop carries no per-tile branch
  bool launched = false;
#define BDE_WGRAD_CASE(MF_, CT_)                                                                                      \
  if (!launched && p.mf == MF_ && p.t.CT == CT_) {                                                                    \
    hipLaunchKernelGGL((conv_lrt_wgrad_kernel<MF_, CT_>), p.grid, dim3(256), p.lds, s, x, g, gvar, part, geo, p.t);    \
    launched = true;                                                                                                  \
  }
  BDE_WGRAD_CASE(16, 1) BDE_WGRAD_CASE(16, 2) BDE_WGRAD_CASE(16, 3) BDE_WGRAD_CASE(16, 4) BDE_WGRAD_CASE(16, 5)
  BDE_WGRAD_CASE(16, 6) BDE_WGRAD_CASE(16, 7) BDE_WGRAD_CASE(16, 8) BDE_WGRAD_CASE(16, 9)
  BDE_WGRAD_CASE(32, 1) BDE_WGRAD_CASE(32, 2) BDE_WGRAD_CASE(32, 3) BDE_WGRAD_CASE(32, 4)
#undef BDE_WGRAD_CASE
  if (!launched) return BDE_ERR_INVALID;
  const int64_t n = static_cast<int64_t>(O) * C * KH * KW;
  hipLaunchKernelGGL(conv_lrt_wgrad_finish_kernel, dim3(static_cast<unsigned>((n + 63) / 64)), dim3(kFinWaves * 64), 0, s, part,
                     p.t.PS, n, w_rho, g_wmu, g_wrho);
  return to_err(hipGetLastError());
}

extern "C" __attribute__((visibility("hidden"))) int bde_internal_load_conv_lrt_bwd(void) {
  hipFuncAttributes attr;
  return bde::to_err(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&bde::conv_lrt_wgrad_kernel<32, 4>)));
}
