// Local-reparameterisation forward of a mean-field CONVOLUTION layer (BBBConv2d, bbb_layers.py:146-154) as ONE kernel:
//
//   activation_mean = conv2d(x,                 W_mu,                       b_mu)         (line 146)
//   activation_var  = conv2d(clamp(x^2, 1e-4),  clamp(softplus(W_rho)^2),   softplus(b_rho)^2)   (line 147)
//   output          = activation_mean + sqrt(activation_var) * eps                          (lines 148-154)
//
// The reference runs two cuDNN/MIOpen convolutions over the same input windows plus ~8 element-wise launches.  Here
// both products are one implicit GEMM with TWO accumulators per output tile: the input patch of a band of output rows
// is staged ONCE into LDS as x and as clamp(x^2) (zero padding applied AFTER the clamp, as F.conv2d pads the clamped
// tensor), the weight tile as W_mu and sigma^2 (sigma^2 comes from the caller: bde_var_operand_fwd mode 1, once per
// weight version), and every k-step issues a pair of f32 MFMAs -- (W_mu, x) and (sigma^2, clamp(x^2)) -- whose B
// operands sit at the same LDS offset of the two patch images.  The epilogue adds the bias terms, draws eps (Philox,
// the stream of bde_local_reparam_fwd: element e of the NCHW output uses normal (e & 3) of group e >> 2) or reads
// it, and writes the output and the total variance (the backward needs sqrt(var)).
//
// GEMM view: rows = output channels (MF = 32 per tile on v_mfma_f32_32x32x2_f32, 16 on v_mfma_f32_16x16x4_f32 for
// layers with <= 16 channels), columns = MF consecutive output pixels of one image (flattened ho * Wo + wo: the
// accumulator of a lane is one pixel x several channels, so a store instruction writes MF consecutive floats per
// channel), k = (c, r, q) over a chunk of CC input channels.  A workgroup = 4 waves = WP pixel-tile groups x WK k-splits
// (small images have too few pixel tiles to fill the chip: the waves then split the reduction and sum their
// accumulators through LDS in wave order -- fixed order, bit-reproducible).
#include "bde_common.hpp"

namespace bde {

using f32x16c = __attribute__((ext_vector_type(16))) float;
using f32x4c = __attribute__((ext_vector_type(4))) float;

struct ConvGeo {
  int N, C, H, W, O, KH, KW, sh, sw, ph, pw, Ho, Wo;
  int dh, dw;      // dilation of the INPUT image (1 in the forward; the layer's stride in the input-gradient pass)
};
struct ConvTile {
  int NI, TH, bands, CC, PH, PWP, WP, WK, PT, tiles_per_img, kcpad_max;
};

template <int MF> struct Mfma;
template <> struct Mfma<32> {
  using Acc = f32x16c;
  static constexpr int KS = 2, REGS = 16;
  __device__ __forceinline__ static Acc run(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
  // accumulator register r of a lane in half h: output row (channel)
  __device__ __forceinline__ static int row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
};
template <> struct Mfma<16> {
  using Acc = f32x4c;
  static constexpr int KS = 4, REGS = 4;
  __device__ __forceinline__ static Acc run(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  __device__ __forceinline__ static int row(int r, int h) { return 4 * h + r; }
};

// LDS: xs [NI][CC][PH][PWP] | x2s (same) | wm [kcpad][MF] | ws [kcpad][MF] | kofs [kcpad] (int)
//
// MODE 0: the forward above.  MODE 1: the INPUT gradient of the same layer as the same implicit GEMM,
//   dx = convT(g, W_mu) + 2 x [x^2 >= 1e-4] * convT(gvar, sigma^2)
// (g = gradient of the layer output, gvar = g eps / (2 sqrt(var)) from bde_local_reparam_bwd): the "input" images are g
// and gvar [N, O', Ho', Wo'] (two tensors instead of x and clamp(x^2)), dilated by the layer's stride (zeros between
// the samples: a stride-s layer spends s^2 times the products here) and padded by K - 1 - p, the "weights" are
// W^T flipped -- gathered from the [O', C', KH, KW] tensors while staging --, the "output" has the layer's C' input
// channels and H' x W' pixels at stride 1, and the epilogue applies the clamp's derivative with x.  ConvGeo then
// describes THAT convolution: C = O', (H, W) = (Ho', Wo') before dilation (dh, dw), O = C', (Ho, Wo) = (H', W').
template <int MF, int PT_MAX, bool RNG, int MODE>
__global__ __launch_bounds__(256, 2) void conv_lrt_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ x_second, const float* __restrict__ wmu,
    const float* __restrict__ ws2, const float* __restrict__ bmu, const float* __restrict__ bvar,
    const float* __restrict__ eps, uint64_t seed, uint64_t stream_id, float* __restrict__ out,
    float* __restrict__ var_out, ConvGeo g, ConvTile t) {
  using M = Mfma<MF>;
  using Acc = typename M::Acc;
  constexpr int KS = M::KS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int img_floats = t.CC * t.PH * t.PWP;
  const int patch_floats = t.NI * img_floats;
  float* xs = lds;
  float* x2s = lds + patch_floats;
  float* wm = lds + 2 * patch_floats;
  float* wsv = wm + t.kcpad_max * MF;
  int* kofs = reinterpret_cast<int*>(wsv + t.kcpad_max * MF);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane / MF, idx = lane % MF;
  const int wp = wave % t.WP, wk = wave / t.WP;
  const int img0 = (blockIdx.x / t.bands) * t.NI, band = blockIdx.x % t.bands;
  const int o0 = blockIdx.y * MF;
  const int ho0 = band * t.TH;
  const int th = min(t.TH, g.Ho - ho0);
  const int band_pixels = th * g.Wo;
  const int khw = g.KH * g.KW;
  const int ktot = g.C * khw;

  int pixoff[PT_MAX], ppix[PT_MAX], pimg[PT_MAX];
  bool pok[PT_MAX];
#pragma unroll
  for (int i = 0; i < PT_MAX; ++i) {
    const int tile = wp + t.WP * i;
    const int img = tile / t.tiles_per_img, p = (tile % t.tiles_per_img) * MF + idx;
    pok[i] = i < t.PT && img < t.NI && img0 + img < g.N && p < band_pixels;
    const int hl = pok[i] ? p / g.Wo : 0, wo = pok[i] ? p % g.Wo : 0;
    pixoff[i] = (pok[i] ? img : 0) * img_floats + hl * g.sh * t.PWP + wo * g.sw;
    ppix[i] = p;
    pimg[i] = img;
  }
  Acc accm[PT_MAX], accv[PT_MAX];
#pragma unroll
  for (int i = 0; i < PT_MAX; ++i) accm[i] = accv[i] = Acc{};

  const int hi0 = ho0 * g.sh - g.ph;                       // (dilated) input row of patch row 0
  for (int c0 = 0; c0 < g.C; c0 += t.CC) {
    const int cc = min(t.CC, g.C - c0);
    const int kc = cc * khw;
    const int kcpad = (kc + KS * t.WK - 1) / (KS * t.WK) * (KS * t.WK);
    __syncthreads();                                       // the previous chunk's operand reads are done
    // ---- stage the input patch: x and clamp(x^2) (zero outside the image: padding is applied after the clamp)
    const int row_elems = t.PH * t.PWP;
    for (int e = threadIdx.x; e < t.NI * cc * row_elems; e += 256) {
      const int img = e / (cc * row_elems), rem = e % (cc * row_elems);
      const int c = rem / row_elems, rr = rem % row_elems;
      const int py = rr / t.PWP, px = rr % t.PWP;
      int hi = hi0 + py, wi = px - g.pw;
      float v = 0.f, v2 = 0.f;
      bool inside = img0 + img < g.N && hi >= 0 && wi >= 0;
      if (MODE == 1) {                                     // dilated input: only multiples of the dilation carry a sample
        inside = inside && hi % g.dh == 0 && wi % g.dw == 0;
        hi /= g.dh;
        wi /= g.dw;
      }
      if (inside && hi < g.H && wi < g.W) {
        const int64_t src = ((static_cast<int64_t>(img0 + img) * g.C + c0 + c) * g.H + hi) * g.W + wi;
        v = x[src];
        v2 = MODE == 0 ? fmaxf(v * v, 1e-4f) : x_second[src];
      }
      const int dst = img * img_floats + c * row_elems + rr;
      xs[dst] = v;
      x2s[dst] = v2;
    }
    // ---- stage the weight tile k-major ([k][MF]: conflict-free A-operand reads); rows past O and k past the chunk are zero
    for (int e = threadIdx.x; e < kcpad * MF; e += 256) {
      const int o = e % MF, k = e / MF;
      float a = 0.f, b = 0.f;
      if (o0 + o < g.O && k < kc) {
        int64_t src;
        if (MODE == 0) {
          src = static_cast<int64_t>(o0 + o) * ktot + c0 * khw + k;
        } else {                                           // W^T flipped: row = the layer's input channel, k = (o', r, q)
          const int oc = c0 + k / khw, rq = k % khw;
          src = (static_cast<int64_t>(oc) * g.O + o0 + o) * khw + (khw - 1 - rq);
        }
        a = wmu[src];
        b = ws2[src];
      }
      wm[e] = a;
      wsv[e] = b;
    }
    for (int k = threadIdx.x; k < kcpad; k += 256) {
      int off = 0;
      if (k < kc) {
        const int c = k / khw, rq = k % khw;
        off = c * row_elems + (rq / g.KW) * t.PWP + (rq % g.KW);
      }
      kofs[k] = off;
    }
    __syncthreads();
    const int ksteps = kcpad / KS;
    const int ks0 = wk * (ksteps / t.WK), ks1 = ks0 + ksteps / t.WK;
    for (int ks = ks0; ks < ks1; ++ks) {
      const int kk = ks * KS + h;
      const int ko = kofs[kk];
      const float am = wm[kk * MF + idx], as = wsv[kk * MF + idx];
#pragma unroll
      for (int i = 0; i < PT_MAX; ++i) {
        if (i < t.PT) {                                    // wave-uniform
          const float b = xs[pixoff[i] + ko], b2 = x2s[pixoff[i] + ko];
          accm[i] = M::run(am, b, accm[i]);
          accv[i] = M::run(as, b2, accv[i]);
        }
      }
    }
  }

  // ---- k-split: waves wk > 0 hand their accumulators to wave wk = 0 of the same pixel-tile group through LDS
  if (t.WK > 1) {
    __syncthreads();
    float* red = lds;                                      // [wave][PT_MAX][2][REGS][64]
    const int per_wave = PT_MAX * 2 * M::REGS * 64;
    if (wk > 0) {
#pragma unroll
      for (int i = 0; i < PT_MAX; ++i)
#pragma unroll
        for (int r = 0; r < M::REGS; ++r) {
          red[wave * per_wave + ((i * 2 + 0) * M::REGS + r) * 64 + lane] = accm[i][r];
          red[wave * per_wave + ((i * 2 + 1) * M::REGS + r) * 64 + lane] = accv[i][r];
        }
    }
    __syncthreads();
    if (wk == 0) {
      for (int s = 1; s < t.WK; ++s) {                     // fixed order
        const int src = (wp + s * t.WP) * per_wave;
#pragma unroll
        for (int i = 0; i < PT_MAX; ++i)
#pragma unroll
          for (int r = 0; r < M::REGS; ++r) {
            accm[i][r] += red[src + ((i * 2 + 0) * M::REGS + r) * 64 + lane];
            accv[i][r] += red[src + ((i * 2 + 1) * M::REGS + r) * 64 + lane];
          }
      }
    }
  }
  if (wk != 0) return;

  // ---- epilogue: bias terms, noise, output + total variance
  const int64_t howo = static_cast<int64_t>(g.Ho) * g.Wo;
#pragma unroll
  for (int i = 0; i < PT_MAX; ++i) {
    if (!pok[i]) continue;
    const int64_t base = static_cast<int64_t>(img0 + pimg[i]) * g.O * howo + static_cast<int64_t>(ho0) * g.Wo + ppix[i];
#pragma unroll
    for (int r = 0; r < M::REGS; ++r) {
      const int o = o0 + M::row(r, h);
      if (o < g.O) {
        const int64_t e = base + o * howo;
        if (MODE == 1) {                                   // bmu = the layer's input x: d clamp(x^2, 1e-4) / dx = 2 x [x^2 >= 1e-4]
          const float xv = bmu[e];
          out[e] = accm[i][r] + (xv * xv >= 1e-4f ? 2.0f * xv * accv[i][r] : 0.f);
          continue;
        }
        const float mean = accm[i][r] + (bmu ? bmu[o] : 0.f);
        const float var = accv[i][r] + (bvar ? bvar[o] : 0.f);
        float z;
        if (RNG) {
          const f32x4 zz = philox_normal4(seed, stream_id, static_cast<uint64_t>(e >> 2), kDomainDiag);
          const int c = static_cast<int>(e & 3);
          z = c == 0 ? zz.x : c == 1 ? zz.y : c == 2 ? zz.z : zz.w;
        } else {
          z = eps[e];
        }
        out[e] = mean + __builtin_sqrtf(var) * z;
        var_out[e] = var;
      }
    }
  }
}

}  // namespace bde

using namespace bde;

namespace {

struct FwdPlan {
  ConvTile t;
  int mf;
  dim3 grid;
  size_t lds;
};

// Tile choice: enough workgroups to fill 256 CUs (>= 2 per CU where the layer allows), LDS <= 64 KB per workgroup
// (two resident), <= PT_MAX pixel tiles per wave.
static bool plan_fwd(const ConvGeo& g, FwdPlan& p) {
  const int mf = g.O <= 16 ? 16 : 32;
  const int pt_max = mf == 16 ? 8 : 4;
  const int ks = mf == 32 ? 2 : 4;
  const int khw = g.KH * g.KW;
  const int64_t howo = static_cast<int64_t>(g.Ho) * g.Wo;
  ConvTile best{};
  bool found = false;
  double best_score = -1.0;
  const int otiles = (g.O + mf - 1) / mf;
  for (int wk = 1; wk <= 4; wk *= 2) {
    const int wpn = 4 / wk;
    for (int th = 1; th <= g.Ho; ++th) {
      if (th != g.Ho && th != 1 && th != 2 && th != 4 && th != 8 && th != 16 && th != 32) continue;
      const int bands = (g.Ho + th - 1) / th;
      const int tiles_per_img = static_cast<int>((static_cast<int64_t>(th) * g.Wo + mf - 1) / mf);
      for (int ni = 1; ni <= 8; ni *= 2) {
        if (ni > 1 && th != g.Ho) continue;               // several images per workgroup only for whole (small) images
        const int tiles = ni * tiles_per_img;
        const int pt = (tiles + wpn - 1) / wpn;
        if (pt > pt_max) continue;
        const int ph = (th - 1) * g.sh + g.KH, pwp = (g.Wo - 1) * g.sw + g.KW;
        for (int cc = g.C; cc >= 1; cc = (cc > 8 ? cc / 2 : cc - 1)) {
          const int kc = cc * khw;
          const int kcpad = (kc + ks * wk - 1) / (ks * wk) * (ks * wk);
          const size_t lds = sizeof(float) * (2ull * ni * cc * ph * pwp + 2ull * kcpad * mf + kcpad);
          const size_t red = wk > 1 ? sizeof(float) * 4ull * pt_max * 2 * (mf == 32 ? 16 : 4) * 64 : 0;
          if (std::max(lds, red) > 64 * 1024) continue;
          const int64_t wgs = static_cast<int64_t>((g.N + ni - 1) / ni) * bands * otiles;
          // score: chip fill first (up to 4 workgroups per CU), then fewer chunks / less halo, then no k-split
          const double fill = std::min(1.0, static_cast<double>(wgs) / 512.0);
          const double util = static_cast<double>(tiles) / (pt * wpn);               // idle waves / partial tiles
          const double halo = static_cast<double>(th) / ph;
          const double chunks = 1.0 / ((g.C + cc - 1) / cc);
          const double score = fill * util * (0.6 + 0.4 * halo) * (0.8 + 0.2 * chunks) * (wk == 1 ? 1.0 : 0.9);
          if (score > best_score) {
            best_score = score;
            best = ConvTile{ni, th, bands, cc, ph, pwp, wpn, wk, pt, tiles_per_img, kcpad};
            found = true;
          }
          break;                                            // the largest chunk that fits is the one to take
        }
      }
    }
  }
  (void)howo;
  if (!found) return false;
  p.t = best;
  p.mf = mf;
  p.grid = dim3(static_cast<unsigned>(((g.N + best.NI - 1) / best.NI) * best.bands), static_cast<unsigned>(otiles));
  const size_t lds = sizeof(float) * (2ull * best.NI * best.CC * best.PH * best.PWP + 2ull * best.kcpad_max * mf + best.kcpad_max);
  const size_t red = best.WK > 1 ? sizeof(float) * 4ull * pt_max * 2 * (mf == 32 ? 16 : 4) * 64 : 0;
  p.lds = std::max(lds, red);
  return true;
}

}  // namespace

static bool layer_geo(int N, int C, int H, int W, int O, int KH, int KW, int sh, int sw, int ph, int pw, ConvGeo& g) {
  if (N < 1 || C < 1 || H < 1 || W < 1 || O < 1 || KH < 1 || KW < 1 || KH > 7 || KW > 7 || sh < 1 || sw < 1 || ph < 0 || pw < 0)
    return false;
  const int Ho = (H + 2 * ph - KH) / sh + 1, Wo = (W + 2 * pw - KW) / sw + 1;
  if (Ho < 1 || Wo < 1) return false;
  g = ConvGeo{N, C, H, W, O, KH, KW, sh, sw, ph, pw, Ho, Wo, 1, 1};
  return true;
}

// the input-gradient pass as a convolution: channels O -> C, g dilated by the stride, padding K - 1 - p, stride 1
static bool data_grad_geo(const ConvGeo& l, ConvGeo& g) {
  if (l.ph > l.KH - 1 || l.pw > l.KW - 1) return false;
  g = ConvGeo{l.N, l.O, l.Ho, l.Wo, l.C, l.KH, l.KW, 1, 1, l.KH - 1 - l.ph, l.KW - 1 - l.pw, l.H, l.W, l.sh, l.sw};
  return true;
}

extern "C" int bde_conv_lrt_supported(int N, int C, int H, int W, int O, int KH, int KW, int sh, int sw, int ph, int pw) {
  ConvGeo g, d;
  FwdPlan p;
  return layer_geo(N, C, H, W, O, KH, KW, sh, sw, ph, pw, g) && plan_fwd(g, p) && data_grad_geo(g, d) && plan_fwd(d, p) ? 1 : 0;
}

extern "C" int bde_conv_lrt_fwd(const float* x, const float* w_mu, const float* w_s2, const float* b_mu, const float* b_var,
                                const float* eps, uint64_t seed, uint64_t stream_id, float* out, float* var_out, int N,
                                int C, int H, int W, int O, int KH, int KW, int sh, int sw, int ph, int pw, void* stream) {
  ConvGeo g;
  FwdPlan p;
  if (!x || !w_mu || !w_s2 || !out || !var_out || !layer_geo(N, C, H, W, O, KH, KW, sh, sw, ph, pw, g) || !plan_fwd(g, p))
    return BDE_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* none = nullptr;
#define BDE_CONV_LAUNCH(MF_, PT_, RNG_)                                                                                  \
  hipLaunchKernelGGL((conv_lrt_fwd_kernel<MF_, PT_, RNG_, 0>), p.grid, dim3(256), p.lds, s, x, none, w_mu, w_s2, b_mu, b_var, eps, \
                     seed, stream_id, out, var_out, g, p.t)
  if (p.mf == 16) { if (eps) BDE_CONV_LAUNCH(16, 8, false); else BDE_CONV_LAUNCH(16, 8, true); }
  else { if (eps) BDE_CONV_LAUNCH(32, 4, false); else BDE_CONV_LAUNCH(32, 4, true); }
#undef BDE_CONV_LAUNCH
  return to_err(hipGetLastError());
}

extern "C" int bde_conv_lrt_bwd_data(const float* g_out, const float* g_var, const float* w_mu, const float* w_s2, const float* x,
                                     float* g_x, int N, int C, int H, int W, int O, int KH, int KW, int sh, int sw, int ph,
                                     int pw, void* stream) {
  ConvGeo l, g;
  FwdPlan p;
  if (!g_out || !g_var || !w_mu || !w_s2 || !x || !g_x || !layer_geo(N, C, H, W, O, KH, KW, sh, sw, ph, pw, l) ||
      !data_grad_geo(l, g) || !plan_fwd(g, p))
    return BDE_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* none = nullptr;
  float* no_out = nullptr;
  if (p.mf == 16)
    hipLaunchKernelGGL((conv_lrt_fwd_kernel<16, 8, false, 1>), p.grid, dim3(256), p.lds, s, g_out, g_var, w_mu, w_s2, x, none, none,
                       uint64_t{0}, uint64_t{0}, g_x, no_out, g, p.t);
  else
    hipLaunchKernelGGL((conv_lrt_fwd_kernel<32, 4, false, 1>), p.grid, dim3(256), p.lds, s, g_out, g_var, w_mu, w_s2, x, none, none,
                       uint64_t{0}, uint64_t{0}, g_x, no_out, g, p.t);
  return to_err(hipGetLastError());
}

extern "C" __attribute__((visibility("hidden"))) int bde_internal_load_conv_lrt(void) {
  hipFuncAttributes attr;
  return bde::to_err(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&bde::conv_lrt_fwd_kernel<32, 4, true, 0>)));
}
