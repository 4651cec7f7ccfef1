// Local-reparameterisation CONVOLUTION layer (BBBConv2d, bbb_layers.py:146-154): forward and input gradient, each ONE
// kernel.
//
//   activation_mean = conv2d(x,                 W_mu,                       b_mu)         (line 146)
//   activation_var  = conv2d(clamp(x^2, 1e-4),  clamp(softplus(W_rho)^2),   softplus(b_rho)^2)   (line 147)
//   output          = activation_mean + sqrt(activation_var) * eps                          (lines 148-154)
//
// The reference runs two cuDNN/MIOpen convolutions over the same input windows plus ~8 element-wise launches.  Here
// both products are one implicit GEMM with TWO accumulators per output tile: the input patch of a band of output rows
// is staged ONCE into LDS as PAIRS (x, clamp(x^2)) (zero padding applied AFTER the clamp, as F.conv2d pads the clamped
// tensor), the weight tile as pairs (W_mu, sigma^2), and every k-step issues a pair of f32 MFMAs -- (W_mu, x) and
// (sigma^2, clamp(x^2)) -- whose operands arrive with ONE ds_read_b64 each.  The epilogue transposes each accumulator tile
// through LDS so that a lane owns 4 consecutive pixels of one channel, adds the bias terms, draws eps (Philox, the stream of
// bde_local_reparam_fwd: float4 group e >> 2 of the flat NCHW output) or reads it, and writes output and total variance
// (the backward needs sqrt(var)) as 16-byte stores.
//
// Round 5 (the kernel has still not run on an MI355X; what guided these was the latency count and the ISA): staging flat over
// the lanes with eight loads in flight and no branch between them (conv_common.hpp), the weight tile requested ahead of the
// patch where the registers allow, two operand sets used alternately (no register copies), the tap -> offset entry read a
// step ahead, no global load of the epilogue at its point of use (bias terms through LDS), XCD-aware work assignment,
// every candidate tiling enumerable and pinnable (tools/conv_autotune.py).
//
// GEMM view: rows = output channels (MF = 32 per tile on v_mfma_f32_32x32x2_f32, 16 on v_mfma_f32_16x16x4_f32 for
// layers with <= 16 channels), columns = MF consecutive output pixels of one image (flattened ho * Wo + wo),
// k = (c, r, q) over a chunk of CC input channels.  A workgroup = 4 waves = WP pixel-tile groups x WK k-splits (small
// images have too few pixel tiles to fill the chip: the waves then split the reduction and sum their accumulators
// through LDS in wave order -- fixed order, bit-reproducible).
//
// Weights come pre-arranged by bde_conv_lrt_prep (once per weight version; it also evaluates sigma^2): k-major
// [C KH KW][O padded to 32] for the forward, transposed and flipped [O KH KW][C padded to 32] for the input gradient,
// so that staging a weight tile is a run of aligned 16-byte copies.
//
// MODE 1: the INPUT gradient as the same implicit GEMM,
//   dx = convT(g, W_mu) + 2 x [x^2 >= 1e-4] * convT(gvar, sigma^2)
// (g = gradient of the layer output, gvar = g eps / (2 sqrt(var)) from bde_conv_lrt_gvar_bias): the "input" images are
// g and gvar [N, O', Ho', Wo'] (two tensors instead of x and clamp(x^2)), dilated by the layer's stride (zeros between
// the samples: a stride-s layer spends s^2 times the products here) and padded by K - 1 - p, the "output" has the
// layer's C' input channels and H' x W' pixels at stride 1, and the epilogue applies the clamp's derivative with x.
// ConvGeo then describes THAT convolution: C = O', (H, W) = (Ho', Wo') before dilation (dh, dw), O = C',
// (Ho, Wo) = (H', W').
#include "conv_common.hpp"
#include <array>
#include <map>
#include <mutex>
#include <vector>

namespace bde {

using f32x16c = __attribute__((ext_vector_type(16))) float;
using f32x4c = __attribute__((ext_vector_type(4))) float;

struct ConvGeo {
  int N, C, H, W, O, KH, KW, sh, sw, ph, pw, Ho, Wo;
  int dh, dw;      // dilation of the INPUT image (1 in the forward; the layer's stride in the dilated input-gradient pass)
  int rp;          // row pitch of the pre-arranged weights: output channels padded to 32
  // where output pixel (i, j) of this launch lives in the output planes [OH][OW]: (oh0 + osh i, ow0 + osw j).  Ho x Wo planes
  // written densely (osh = osw = 1, oh0 = ow0 = 0, OH = Ho, OW = Wo) everywhere except the per-phase launches of the
  // input gradient of a strided layer, which write every sh-th row / sw-th column of g_x.
  int osh = 1, osw = 1, oh0 = 0, ow0 = 0, OH = 0, OW = 0;
};
constexpr int kConvTablePad = 16;   // zero entries behind the tap table (the product loop looks up to 3 k-steps = 12 entries ahead)

struct ConvTile {
  int NI, TH, bands, CC, PH, PWP, WP, WK, tiles_per_img, kcpad_max;
  float rcp_pwp;     // fl(1 / PWP): the staging pass splits a flat patch index into (row, column) with it (conv_common.hpp)
  int bias_off;      // float offset of the 2 x MF bias terms in LDS (behind everything the main loop and the epilogue use)
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

// LDS: xq [NI][CC][PH][PWP] pairs (x, clamp(x^2)) | wq [kcpad][MF] pairs (W_mu, sigma^2) | kofs [kcpad + 16] (int, byte offsets); the epilogue
// reuses the front of it: per wave 2 x [MF][MF + 4] floats.  Pairs: both products of a k-step read their operands at the
// same element, so one ds_read_b64 (one address computation) serves both.
template <int MF, int PT, bool RNG, int MODE>
__global__ __launch_bounds__(256, 2) void conv_lrt_kernel(
    const float* __restrict__ x, const float* __restrict__ x_second, const float* __restrict__ wt_mu,
    const float* __restrict__ wt_s2, const float* __restrict__ bmu, const float* __restrict__ bvar,
    const float* __restrict__ eps, uint64_t seed, uint64_t stream_id, float* __restrict__ out,
    float* __restrict__ var_out, ConvGeo g, ConvTile t) {
  using M = Mfma<MF>;
  using Acc = typename M::Acc;
  constexpr int KS = M::KS;
  constexpr int RS = MF + 4;                               // row pitch of the epilogue's transposition tile
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int row_elems = t.PH * t.PWP;
  const int img_floats = t.CC * row_elems;
  const int patch_floats = (t.NI * img_floats + 3) & ~3;    // the weight tiles behind the two patch images are written as float4
  f32x2* xq = reinterpret_cast<f32x2*>(lds);
  f32x2* wq = reinterpret_cast<f32x2*>(lds + 2 * patch_floats);
  int* kofs = reinterpret_cast<int*>(lds + 2 * patch_floats + 2 * t.kcpad_max * MF);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane / MF, idx = lane % MF;
  const int wp = wave % t.WP, wk = wave / t.WP;
  // work index -> (image group x band, channel tile), channel tile fastest: the channel tiles of one band read the same patch and
  // run on one XCD back to back (xcd_work_index)
  const int work = xcd_work_index(static_cast<int>(blockIdx.y * gridDim.x + blockIdx.x), static_cast<int>(gridDim.x * gridDim.y));
  const int bx = work / static_cast<int>(gridDim.y), by = work % static_cast<int>(gridDim.y);
  const int img0 = (bx / t.bands) * t.NI, band = bx % t.bands;
  const int o0 = by * MF;
  const int ho0 = band * t.TH;
  const int th = min(t.TH, g.Ho - ho0);
  const int band_pixels = th * g.Wo;
  const int khw = g.KH * g.KW;

  // a wave's PT tiles are consecutive: tile = wp * PT + i
  int pixoff[PT];                                          // BYTE offsets of the pairs (the table's entries too: one add per request)
#pragma unroll
  for (int i = 0; i < PT; ++i) {
    const int tile = wp * PT + i;
    const int img = tile / t.tiles_per_img, p = (tile % t.tiles_per_img) * MF + idx;
    const bool ok = img < t.NI && img0 + img < g.N && p < band_pixels;
    const int hl = ok ? p / g.Wo : 0, wo = ok ? p % g.Wo : 0;
    pixoff[i] = 8 * ((ok ? img : 0) * img_floats + hl * g.sh * t.PWP + wo * g.sw);
  }
  Acc accm[PT], accv[PT];
#pragma unroll
  for (int i = 0; i < PT; ++i) accm[i] = accv[i] = Acc{};

  // patch offset (in bytes of pairs) of tap k = (c, r, q), chunk-relative: the same for every chunk.  kConvTablePad entries past
  // the chunk are zero: the product loop reads its table entry up to three steps ahead without clamping the index
  for (int k = threadIdx.x; k < t.kcpad_max + kConvTablePad; k += 256) {
    const int c = k / khw, rq = k % khw;
    kofs[k] = (k < t.kcpad_max && c < t.CC) ? 8 * (c * row_elems + (rq / g.KW) * t.PWP + (rq % g.KW)) : 0;
  }

  // the tile's bias terms (forward: mean bias, bias variance) go to LDS once, behind everything else: the epilogue reads them
  // with an LDS latency instead of a global one per pass (they become visible at the first barrier of the chunk loop)
  float* bias_lds = lds + t.bias_off;
  if (MODE == 0 && threadIdx.x < 2 * MF) {
    const int c = threadIdx.x % MF, which = threadIdx.x / MF;
    const float* src = which ? bvar : bmu;
    bias_lds[threadIdx.x] = (src && o0 + c < g.O) ? src[o0 + c] : 0.f;
  }
  const int hi0 = ho0 * g.sh - g.ph;                       // (dilated) input row of patch row 0
  for (int c0 = 0; c0 < g.C; c0 += t.CC) {
    const int cc = min(t.CC, g.C - c0);
    const int kc = cc * khw;
    const int kcpad = (kc + KS * t.WK - 1) / (KS * t.WK) * (KS * t.WK);
    if (c0 > 0) __syncthreads();                           // the previous chunk's operand reads are done
    // ---- the weight tile, k-major [k][MF] pairs (conflict-free A-operand reads): aligned float4 copies of the pre-arranged
    //      [k][rp] matrices, interleaved on the way; k past the chunk is zero.  Several copies in flight per thread (one copy at a
    //      time put a full memory round trip behind every 32 bytes).  With one or two pixel tiles per wave the registers are
    //      there to REQUEST the first eight copies of a thread before the patch is staged and to store them after it: the
    //      tile's round trip then overlaps the patch's instead of following it (at the 32- and 64-channel CIFAR layers the
    //      weight tile was two of the three round trips of a chunk, and a chunk's products are shorter than one).
    constexpr int Q = MF / 4;
    constexpr int WB = PT <= 2 ? 8 : 0;                      // copies requested ahead of the patch (registers permitting)
    const int64_t k_base = static_cast<int64_t>(c0) * khw;
    float* wflat = reinterpret_cast<float*>(wq);
    const int n_e = kcpad * Q;
    auto w_src = [&](int e) {                                // (a copy past the tile / the chunk: row 0 stands in, dropped at the store)
      const int k = e / Q, o4 = e % Q;
      return (k_base + ((e < n_e && k < kc) ? k : 0)) * g.rp + o0 + 4 * o4;
    };
    auto w_store = [&](int e, const f32x4& a, const f32x4& b) {
      if (e < n_e) {
        const int k = e / Q, o4 = e % Q;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 av = k < kc ? a : z, bv = k < kc ? b : z;
        st4(wflat + 2 * (k * MF + 4 * o4), f32x4{av.x, bv.x, av.y, bv.y});
        st4(wflat + 2 * (k * MF + 4 * o4) + 4, f32x4{av.z, bv.z, av.w, bv.w});
      }
    };
    f32x4 wa[WB > 0 ? WB : 1], wb[WB > 0 ? WB : 1];
#pragma unroll
    for (int u = 0; u < WB; ++u) {
      const int64_t src = w_src(static_cast<int>(threadIdx.x) + u * 256);
      wa[u] = ld4(wt_mu + src);
      wb[u] = ld4(wt_s2 + src);
    }
    // ---- the input patch of the chunk: x and clamp(x^2) (zero outside the image: padding is applied after the clamp);
    //      MODE 1: g and gvar, dilated.  Flat over the lanes, eight loads in flight per lane (conv_common.hpp)
    if (MODE == 1 && (g.dh != 1 || g.dw != 1))               // (uniform for the launch: the dilated input-gradient pass)
      conv_stage_patch<1, true>(x, x_second, xq, wave, lane, t.NI * cc, cc, row_elems, t.PWP, t.rcp_pwp, img_floats, img0, g.N,
                                g.C, c0, g.H, g.W, hi0, g.pw, g.dh, g.dw);
    else
      conv_stage_patch<MODE, false>(x, MODE == 1 ? x_second : x, xq, wave, lane, t.NI * cc, cc, row_elems, t.PWP, t.rcp_pwp,
                                    img_floats, img0, g.N, g.C, c0, g.H, g.W, hi0, g.pw, 1, 1);
#pragma unroll
    for (int u = 0; u < WB; ++u) w_store(static_cast<int>(threadIdx.x) + u * 256, wa[u], wb[u]);
    for (int e0 = static_cast<int>(threadIdx.x) + WB * 256; e0 < n_e; e0 += 4 * 256) {     // the rest, four copies in flight
      f32x4 a[4], b4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t src = w_src(e0 + u * 256);
        a[u] = ld4(wt_mu + src);
        b4[u] = ld4(wt_s2 + src);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) w_store(e0 + u * 256, a[u], b4[u]);
    }
    __syncthreads();
    const int ksteps = kcpad / KS;
    const int ks0 = wk * (ksteps / t.WK), ks1 = ks0 + ksteps / t.WK;
    // Two operand sets, used alternately: the operands of step ks + 1 are requested before the products of step ks are issued,
    // and no register is copied (round 4 copied the prefetched set into the "current" one: 2 + 2 PT moves per step)
    // The tap -> patch-offset entry of a step is read one step BEFORE its operands are requested (ko_next): an operand
    // request then never waits for an LDS round trip of its own address -- with one pixel tile per wave a k-step is only two
    // MFMAs long, and the dependent table read was as long as the step.
    f32x2 a0, a1, b0[PT], b1[PT];
    auto fetch = [&](int kk_, int ko, f32x2& a, f32x2(&b)[PT]) {
      a = wq[kk_ * MF + idx];
#pragma unroll
      for (int i = 0; i < PT; ++i) b[i] = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(xq) + (pixoff[i] + ko));
    };
    auto products = [&](const f32x2& a, const f32x2(&b)[PT]) {
#pragma unroll
      for (int i = 0; i < PT; ++i) {
        accm[i] = M::run(a.x, b[i].x, accm[i]);
        accv[i] = M::run(a.y, b[i].y, accv[i]);
      }
    };
    int ks = ks0, kk = ks0 * KS + h;
    int ko_next = 0;
    if (ks < ks1) {
      fetch(kk, kofs[kk], a0, b0);
      ko_next = kofs[kk + KS];
    }
    for (; ks + 1 < ks1; ks += 2, kk += 2 * KS) {
      fetch(kk + KS, ko_next, a1, b1);
      ko_next = kofs[kk + 2 * KS];
      products(a0, b0);
      if (ks + 2 < ks1) {
        fetch(kk + 2 * KS, ko_next, a0, b0);
        ko_next = kofs[kk + 3 * KS];
      }
      products(a1, b1);
    }
    if (ks < ks1) products(a0, b0);                         // an odd number of steps: the last set fetched is still pending
  }

  // ---- k-split: waves wk > 0 hand their accumulators to wave wk = 0 of the same pixel-tile group through LDS
  __syncthreads();
  if (t.WK > 1) {
    float* red = lds;                                      // [wave][PT][2][REGS][64]
    constexpr int per_wave = PT * 2 * M::REGS * 64;
    if (wk > 0) {
#pragma unroll
      for (int i = 0; i < PT; ++i)
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
        for (int i = 0; i < PT; ++i)
#pragma unroll
          for (int r = 0; r < M::REGS; ++r) {
            accm[i][r] += red[src + ((i * 2 + 0) * M::REGS + r) * 64 + lane];
            accv[i][r] += red[src + ((i * 2 + 1) * M::REGS + r) * 64 + lane];
          }
      }
    }
    __syncthreads();
  }
  if (wk != 0) return;

  // ---- epilogue.  Each tile goes through a per-wave LDS tile [MF channels][MF pixels] so that a lane ends up with 4
  // consecutive pixels of one channel: one Philox call per float4, 16-byte loads / stores.  Needs the float4 groups of
  // the flat output aligned with the tiles (Ho Wo and the band size multiples of 4); otherwise element by element.
  const int64_t howo = static_cast<int64_t>(g.Ho) * g.Wo;
  const uintptr_t ptr_bits = reinterpret_cast<uintptr_t>(out) | (MODE == 1 ? reinterpret_cast<uintptr_t>(bmu)
                                                                            : reinterpret_cast<uintptr_t>(var_out) |
                                                                                  (RNG ? 0 : reinterpret_cast<uintptr_t>(eps)));
  const bool dense = g.osh == 1 && g.osw == 1 && g.OH == g.Ho && g.OW == g.Wo;
  const bool vec = dense && (howo & 3) == 0 && ((t.TH * g.Wo) & 3) == 0 && (ptr_bits & 15) == 0;
  const int64_t plane = static_cast<int64_t>(g.OH) * g.OW;
  float* tm = lds + wave * 2 * MF * RS;
  float* tv = tm + MF * RS;
  constexpr int Q = MF / 4;                                // float4 groups per tile row
  constexpr int NPASS = MF * Q / 64;                       // float4 passes of a wave over a tile (4 at MF = 32, 1 at MF = 16)
  // No global load of the epilogue sits where it is consumed: the bias terms come from LDS (above), the tile's x / noise operands
  // are requested a batch ahead (before the tile's LDS transposition) -- a load issued at its use costs a memory round trip per
  // pass / per accumulator register: up to 16 dependent round trips per tile, as long as the tile's products.
  // a rolled loop over the wave's tiles (its body with the Philox code is too large to replicate PT times: the compiler
  // then gives up unrolling and puts the accumulators into scratch): tile i's accumulators are picked by uniform branches
#pragma unroll 1
  for (int i = 0; i < PT; ++i) {
    const int tile = wp * PT + i;
    const int img = tile / t.tiles_per_img, p0 = (tile % t.tiles_per_img) * MF;
    if (img >= t.NI || img0 + img >= g.N || p0 >= band_pixels) continue;          // wave-uniform
    const int64_t base = static_cast<int64_t>(img0 + img) * g.O * howo + static_cast<int64_t>(ho0) * g.Wo + p0;
    Acc cm = accm[0], cv = accv[0];
#pragma unroll
    for (int j = 1; j < PT; ++j) {
      if (j == i) {
        cm = accm[j];
        cv = accv[j];
      }
    }
    if (vec) {
      // the tile's global operands (x for the clamp's derivative, or the supplied noise) are requested PB passes at a time, the
      // first batch before the transposition (in flight during it); PB = 1 where the accumulators fill the register file
      constexpr bool kLoads = MODE == 1 || !RNG;
      constexpr int PB = (PT * M::REGS >= 64 && NPASS > 1) ? 1 : NPASS;
      f32x4 gop[PB];
      auto request = [&](int pass0) {
#pragma unroll
        for (int u = 0; u < PB; ++u) {
          const int ch = (pass0 + u) * (64 / Q) + lane / Q, p4 = lane % Q;
          const bool live = o0 + ch < g.O && p0 + 4 * p4 < band_pixels;
          const int64_t e = base + (o0 + ch) * howo + 4 * p4;
          if (kLoads) gop[u] = ld4((MODE == 1 ? bmu : eps) + (live ? e : int64_t{0}));   // (element 0 stands in, dropped)
        }
      };
      request(0);
#pragma unroll
      for (int r = 0; r < M::REGS; ++r) {
        tm[M::row(r, h) * RS + idx] = cm[r];
        tv[M::row(r, h) * RS + idx] = cv[r];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes have landed
#pragma unroll
      for (int pass0 = 0; pass0 < NPASS; pass0 += PB) {
        if (pass0 > 0) request(pass0);
#pragma unroll
        for (int u = 0; u < PB; ++u) {
          const int pass = pass0 + u;
          const int ch = pass * (64 / Q) + lane / Q, p4 = lane % Q;
          if (o0 + ch < g.O && p0 + 4 * p4 < band_pixels) {
            const f32x4 m4 = ld4(tm + ch * RS + 4 * p4), v4 = ld4(tv + ch * RS + 4 * p4);
            const int64_t e = base + (o0 + ch) * howo + 4 * p4;
            if (MODE == 1) {                               // bmu = the layer's input x: d clamp(x^2, 1e-4) / dx = 2 x [x^2 >= 1e-4]
              const f32x4 xv = gop[u];
              f32x4 d;
#pragma unroll
              for (int c = 0; c < 4; ++c) d[c] = m4[c] + (xv[c] * xv[c] >= 1e-4f ? 2.0f * xv[c] * v4[c] : 0.f);
              st4(out + e, d);
            } else {
              const float bm = bias_lds[ch], bv = bias_lds[MF + ch];
              const f32x4 z = RNG ? philox_normal4(seed, stream_id, static_cast<uint64_t>(e >> 2), kDomainDiag) : gop[u];
              f32x4 res, var;
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                var[c] = v4[c] + bv;
                res[c] = (m4[c] + bm) + __builtin_sqrtf(var[c]) * z[c];
              }
              st4(out + e, res);
              if (var_out) st4(var_out + e, var);        // (NULL: a forward nobody will differentiate)
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next tile overwrites the LDS tile
    } else {
      const int p = p0 + idx;
      if (p < band_pixels) {
        // element (image, channel o, pixel p of the band) in the output planes: dense, or the phase's rows / columns of g_x
        const int prow = ho0 + p / g.Wo, pcol = p % g.Wo;
        const int64_t pix = static_cast<int64_t>(g.oh0 + g.osh * prow) * g.OW + g.ow0 + g.osw * pcol;
        const int64_t img_base = static_cast<int64_t>(img0 + img) * g.O * plane;
        // four or eight accumulator registers (channels) at a time: their loads first (x, or the supplied noise), then the
        // arithmetic and the stores
        constexpr int GR = M::REGS < 8 ? M::REGS : (PT * M::REGS >= 64 ? 4 : 8);     // 4 where the accumulators fill the register file
#pragma unroll
        for (int r0 = 0; r0 < M::REGS; r0 += GR) {
          float l1[GR];
#pragma unroll
          for (int u = 0; u < GR; ++u) {
            const int o = o0 + M::row(r0 + u, h);
            const int64_t e = o < g.O ? img_base + o * plane + pix : int64_t{0};
            l1[u] = 0.f;
            if (MODE == 1) l1[u] = bmu[e];
            else if (!RNG) l1[u] = eps[e];
          }
#pragma unroll
          for (int u = 0; u < GR; ++u) {
            const int r = r0 + u;
            const int o = o0 + M::row(r, h);
            if (o < g.O) {
              const int64_t e = img_base + o * plane + pix;
              if (MODE == 1) {
                const float xv = l1[u];
                out[e] = cm[r] + (xv * xv >= 1e-4f ? 2.0f * xv * cv[r] : 0.f);
              } else {
                const float mean = cm[r] + bias_lds[M::row(r, h)];
                const float var = cv[r] + bias_lds[MF + M::row(r, h)];
                float z;
                if (RNG) {
                  const f32x4 zz = philox_normal4(seed, stream_id, static_cast<uint64_t>(e >> 2), kDomainDiag);
                  const int c = static_cast<int>(e & 3);
                  z = c == 0 ? zz.x : c == 1 ? zz.y : c == 2 ? zz.z : zz.w;
                } else {
                  z = l1[u];
                }
                out[e] = mean + __builtin_sqrtf(var) * z;
                if (var_out) var_out[e] = var;
              }
            }
          }
        }
      }
    }
  }
}

// Once per weight version: sigma^2 = clamp(softplus(rho)^2, 1e-4), its rho-derivative, and both weight matrices in the
// layouts the kernels stage from.  wbuf (zero-initialised by the caller once; the padding is never written):
//   [0]  WT_mu [ktot][op]      [1]  WT_s2 [ktot][op]           k = (c, r, q), op = O padded to 32    (forward)
//   [2]  WB_mu [O khw][cp]     [3]  WB_s2 [O khw][cp]          k' = (o, flipped tap), cp = C padded to 32 (input gradient)
//   [4]  DS2   [O][ktot]       [sigma^2 >= 1e-4] 2 sigma sigmoid(rho)                                 (weight gradient)
//   [5]  BVAR  [op]            softplus(b_rho)^2, NOT clamped (bbb_layers.py:147); zero without a bias  ([4] rounded up to 4 floats)
//   [6]  WP_mu, [7] WP_s2      the input-gradient matrices of a STRIDED layer, one block per phase (a, b) of the output pixel
//        grid, a-major: [O][R_a][Q_b][cp] with the taps r = rho_a + sh r' (rho_a = (a + ph) mod sh) flipped within the phase
//        (PhaseGeo below) -- written by bde_conv_lrt_prep_strided only
struct PhaseAxis {
  int rho, taps, delta, count;     // first tap, number of taps, (a + p - rho) / s, number of output rows / columns of the phase
};
__host__ __device__ static inline PhaseAxis phase_axis(int a, int K, int s, int p, int extent) {
  PhaseAxis x;
  x.rho = (a + p) % s;
  x.taps = x.rho < K ? (K - x.rho + s - 1) / s : 0;
  x.delta = (a + p - x.rho) / s;
  x.count = a < extent ? (extent - a + s - 1) / s : 0;
  return x;
}

__global__ __launch_bounds__(kBlock) void conv_lrt_prep_kernel(const float* __restrict__ w_mu, const float* __restrict__ w_rho,
                                                               const float* __restrict__ b_rho, int O, int C, int KH, int KW,
                                                               int op, int cp, int sh, int sw, int ph, int pw,
                                                               float* __restrict__ wbuf) {
  const int khw = KH * KW;
  const int ktot = C * khw;
  const int64_t n = static_cast<int64_t>(O) * ktot;
  float* wt_mu = wbuf;
  float* wt_s2 = wt_mu + static_cast<int64_t>(ktot) * op;
  float* wb_mu = wt_s2 + static_cast<int64_t>(ktot) * op;
  float* wb_s2 = wb_mu + static_cast<int64_t>(O) * khw * cp;
  float* ds2 = wb_s2 + static_cast<int64_t>(O) * khw * cp;
  float* bvar = ds2 + ((n + 3) & ~int64_t{3});                // (rounded: the phase matrices behind it are read as float4)
  float* wp_mu = bvar + op;
  float* wp_s2 = wp_mu + static_cast<int64_t>(O) * khw * cp;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (i < O) {
      const float sb = b_rho ? softplus(b_rho[i]) : 0.f;
      bvar[i] = sb * sb;
    }
    const int o = static_cast<int>(i / ktot), k = static_cast<int>(i % ktot);
    const int c = k / khw, rq = k % khw;
    const float mu = w_mu[i];
    const SoftplusSigmoid ss = softplus_sigmoid(w_rho[i]);
    const float s2 = ss.sp * ss.sp;
    const float s2c = fmaxf(s2, 1e-4f);
    wt_mu[static_cast<int64_t>(k) * op + o] = mu;
    wt_s2[static_cast<int64_t>(k) * op + o] = s2c;
    const int64_t kb = static_cast<int64_t>(o) * khw + (khw - 1 - rq);
    wb_mu[kb * cp + c] = mu;
    wb_s2[kb * cp + c] = s2c;
    ds2[i] = s2 >= 1e-4f ? 2.0f * ss.sp * ss.sg : 0.f;
    if (sh > 1 || sw > 1) {
      // tap (r, q) belongs to the phase (a, b) with rho_a = r mod sh, rho_b = q mod sw; the blocks of the earlier phases hold
      // O cp (taps of those phases) floats
      const int r = rq / KW, q = rq % KW;
      const int a = ((r % sh) - (ph % sh) + sh) % sh, b = ((q % sw) - (pw % sw) + sw) % sw;
      int64_t before = 0;
      for (int aa = 0; aa < sh; ++aa)
        for (int bb = 0; bb < sw; ++bb) {
          if (aa * sw + bb >= a * sw + b) continue;
          before += static_cast<int64_t>(phase_axis(aa, KH, sh, ph, 1 << 20).taps) * phase_axis(bb, KW, sw, pw, 1 << 20).taps;
        }
      const PhaseAxis pa = phase_axis(a, KH, sh, ph, 1 << 20), pb = phase_axis(b, KW, sw, pw, 1 << 20);
      const int rf = pa.taps - 1 - r / sh, qf = pb.taps - 1 - q / sw;                    // flipped within the phase
      const int64_t dst = (before * O + (static_cast<int64_t>(o) * pa.taps + rf) * pb.taps + qf) * cp + c;
      wp_mu[dst] = mu;
      wp_s2[dst] = s2c;
    }
  }
}

}  // namespace bde

using namespace bde;

namespace {

struct FwdPlan {
  ConvTile t;
  int mf, pt;
  dim3 grid;
  size_t lds;
};

static inline int pad32(int v) { return (v + 31) / 32 * 32; }

// ---- tilings.  fwd_candidates enumerates every tiling the kernel can run a launch geometry with: 4 waves = WP pixel-tile
// groups x WK k-splits, a band of TH output rows (whole images or power-of-two bands) of NI images (several only for whole
// small images), 1 / 2 / 4 / 8 pixel tiles per wave, the input channels in chunks of CC (the largest chunk that fits
// 64 KB of LDS -- two workgroups per CU -- and half of it), each with a score:
// chip fill first (up to 2 workgroups per CU), then idle waves / partial tiles, halo, chunks, k-split; re-staging the
// weight tile per workgroup costs more the fewer products a workgroup does per chunk.  The planner takes the best score
// unless a tiling has been PINNED for the launch geometry (bde_conv_lrt_set_tiling: tools/conv_autotune.py times every
// candidate on the device and records the winners; the score's weights are hand-set and were never compared with one).
struct FwdCand {
  ConvTile t;
  int pt;
  size_t lds;
  double score;
};

static void fwd_candidates(const ConvGeo& g, std::vector<FwdCand>& out) {
  const int mf = g.O <= 16 ? 16 : 32;
  const int pt_max = mf == 16 ? 8 : 4;
  const int ks = mf == 32 ? 2 : 4;
  const int regs = mf == 32 ? 16 : 4;
  const int khw = g.KH * g.KW;
  const int otiles = (g.O + mf - 1) / mf;
  const size_t epi = sizeof(float) * 4ull * 2 * mf * (mf + 4);
  for (int wk = 1; wk <= 4; wk *= 2) {
    const int wpn = 4 / wk;
    for (int th = 1; th <= g.Ho; ++th) {
      if (th != g.Ho && (th & (th - 1)) != 0) continue;   // whole images or power-of-two bands
      const int bands = (g.Ho + th - 1) / th;
      const int tiles_per_img = static_cast<int>((static_cast<int64_t>(th) * g.Wo + mf - 1) / mf);
      for (int ni = 1; ni <= 8; ni *= 2) {
        if (ni > 1 && th != g.Ho) continue;               // several images per workgroup only for whole (small) images
        const int tiles = ni * tiles_per_img;
        int pt = 1;
        while (pt * wpn < tiles) pt *= 2;
        if (pt > pt_max) continue;
        const int ph = (th - 1) * g.sh + g.KH, pwp = (g.Wo - 1) * g.sw + g.KW;
        int emitted = 0;
        for (int cc = g.C; cc >= 1 && emitted < 2; cc = (cc > 8 ? cc / 2 : cc - 1)) {
          const int kc = cc * khw;
          const int kcpad = (kc + ks * wk - 1) / (ks * wk) * (ks * wk);
          const size_t lds = sizeof(float) * (2ull * ((static_cast<size_t>(ni) * cc * ph * pwp + 3) & ~size_t{3}) + 2ull * kcpad * mf + kcpad + kConvTablePad);
          const size_t red = wk > 1 ? sizeof(float) * 4ull * pt * 2 * regs * 64 : 0;
          const size_t before_bias = std::max(std::max(lds, red), epi);
          const size_t need = before_bias + sizeof(float) * 2 * mf;       // + the tile's bias terms (mean, variance)
          if (need > 64 * 1024) continue;
          const int64_t wgs = static_cast<int64_t>((g.N + ni - 1) / ni) * bands * otiles;
          const double fill = std::min(1.0, static_cast<double>(wgs) / 512.0);
          const double util = static_cast<double>(tiles) / (pt * wpn);
          const double halo = static_cast<double>(th) / ph;
          const double chunks = 1.0 / ((g.C + cc - 1) / cc);
          const double wreuse = std::min(1.0, static_cast<double>(tiles) / 8.0);
          double score = (0.25 + 0.75 * fill) * util * (0.6 + 0.4 * halo) * (0.8 + 0.2 * chunks) * (0.5 + 0.5 * wreuse) *
                         (wk == 1 ? 1.0 : 0.9);
          if (emitted == 1) score *= 0.999;               // the half chunk only ever wins when it is pinned
          out.push_back(FwdCand{ConvTile{ni, th, bands, cc, ph, pwp, wpn, wk, tiles_per_img, kcpad, 1.0f / static_cast<float>(pwp),
                                       static_cast<int>(before_bias / sizeof(float))},
                              pt, need, score});
          ++emitted;                                        // the largest chunk that fits, then the next smaller one
        }
      }
    }
  }
}

// launch geometry -> pinned (WK, TH, NI, CC)
using GeoKey = std::array<int, 15>;
static GeoKey geo_key(const ConvGeo& g) {
  return GeoKey{g.N, g.C, g.H, g.W, g.O, g.KH, g.KW, g.sh, g.sw, g.ph, g.pw, g.dh, g.dw, g.Ho, g.Wo};
}
static std::mutex& pin_mutex() {
  static std::mutex m;
  return m;
}
static std::map<GeoKey, std::array<int, 4>>& pins() {
  static std::map<GeoKey, std::array<int, 4>> m;
  return m;
}

static bool plan_fwd(const ConvGeo& g, FwdPlan& p, int* chosen_index = nullptr) {
  std::vector<FwdCand> cands;
  fwd_candidates(g, cands);
  if (cands.empty()) return false;
  int best = -1;
  {
    std::lock_guard<std::mutex> lock(pin_mutex());
    const auto it = pins().find(geo_key(g));
    if (it != pins().end()) {
      for (size_t i = 0; i < cands.size(); ++i) {
        const ConvTile& t = cands[i].t;
        if (t.WK == it->second[0] && t.TH == it->second[1] && t.NI == it->second[2] && t.CC == it->second[3]) best = static_cast<int>(i);
      }
    }
  }
  if (best < 0) {
    best = 0;
    for (size_t i = 1; i < cands.size(); ++i)
      if (cands[i].score > cands[best].score) best = static_cast<int>(i);
  }
  const FwdCand& c = cands[best];
  const int mf = g.O <= 16 ? 16 : 32;
  p.t = c.t;
  p.mf = mf;
  p.pt = c.pt;
  p.grid = dim3(static_cast<unsigned>(((g.N + c.t.NI - 1) / c.t.NI) * c.t.bands), static_cast<unsigned>((g.O + mf - 1) / mf));
  p.lds = c.lds;
  if (chosen_index) *chosen_index = best;
  return true;
}

static bool layer_geo(int N, int C, int H, int W, int O, int KH, int KW, int sh, int sw, int ph, int pw, ConvGeo& g) {
  if (N < 1 || C < 1 || H < 1 || W < 1 || O < 1 || KH < 1 || KW < 1 || KH > 7 || KW > 7 || sh < 1 || sw < 1 || ph < 0 || pw < 0)
    return false;
  const int Ho = (H + 2 * ph - KH) / sh + 1, Wo = (W + 2 * pw - KW) / sw + 1;
  if (Ho < 1 || Wo < 1) return false;
  g = ConvGeo{N, C, H, W, O, KH, KW, sh, sw, ph, pw, Ho, Wo, 1, 1, pad32(O)};
  g.OH = Ho;
  g.OW = Wo;
  return true;
}

// the input-gradient pass as a convolution: channels O -> C, g dilated by the stride, padding K - 1 - p, stride 1
static bool data_grad_geo(const ConvGeo& l, ConvGeo& g) {
  if (l.ph > l.KH - 1 || l.pw > l.KW - 1) return false;
  g = ConvGeo{l.N, l.O, l.Ho, l.Wo, l.C, l.KH, l.KW, 1, 1, l.KH - 1 - l.ph, l.KW - 1 - l.pw, l.H, l.W, l.sh, l.sw, pad32(l.C)};
  g.OH = l.H;
  g.OW = l.W;
  return true;
}

// Phase (a, b) of the input gradient of a strided layer: the output pixels (a + sh i, b + sw j) of g_x only see the taps
// r = rho_a + sh r', q = rho_b + sw q', and for them the pass is a stride-1 convolution of the UNDILATED g with that
// sub-kernel (flipped), g row i + delta_a - r'  ->  top padding taps_a - 1 - delta_a (rows past the image read as zero).
// The dilated formulation (data_grad_geo) multiplies sh sw times as many products, all but one in sh sw of them by zero.
// false: the phase has no output pixels or no taps (its pixels of g_x are zero).
static bool phase_geo(const ConvGeo& l, int a, int b, ConvGeo& g, int64_t& wp_offset) {
  const PhaseAxis pa = phase_axis(a, l.KH, l.sh, l.ph, l.H), pb = phase_axis(b, l.KW, l.sw, l.pw, l.W);
  int64_t before = 0;
  for (int aa = 0; aa < l.sh; ++aa)
    for (int bb = 0; bb < l.sw; ++bb)
      if (aa * l.sw + bb < a * l.sw + b)
        before += static_cast<int64_t>(phase_axis(aa, l.KH, l.sh, l.ph, l.H).taps) * phase_axis(bb, l.KW, l.sw, l.pw, l.W).taps;
  wp_offset = before * l.O * pad32(l.C);
  if (pa.taps == 0 || pb.taps == 0 || pa.count == 0 || pb.count == 0) return false;
  g = ConvGeo{l.N, l.O, l.Ho, l.Wo, l.C, pa.taps, pb.taps, 1, 1, pa.taps - 1 - pa.delta, pb.taps - 1 - pb.delta, pa.count, pb.count,
              1, 1, pad32(l.C)};
  g.osh = l.sh;
  g.osw = l.sw;
  g.oh0 = a;
  g.ow0 = b;
  g.OH = l.H;
  g.OW = l.W;
  return true;
}

struct WBuf {
  const float *wt_mu, *wt_s2, *wb_mu, *wb_s2, *ds2, *bvar, *wp_mu, *wp_s2;
};
static WBuf wbuf_parts(const float* wbuf, int O, int C, int khw) {
  const int64_t ktot = static_cast<int64_t>(C) * khw, op = pad32(O), cp = pad32(C);
  WBuf b;
  b.wt_mu = wbuf;
  b.wt_s2 = b.wt_mu + ktot * op;
  b.wb_mu = b.wt_s2 + ktot * op;
  b.wb_s2 = b.wb_mu + static_cast<int64_t>(O) * khw * cp;
  b.ds2 = b.wb_s2 + static_cast<int64_t>(O) * khw * cp;
  b.bvar = b.ds2 + ((static_cast<int64_t>(O) * ktot + 3) & ~int64_t{3});
  b.wp_mu = b.bvar + op;
  b.wp_s2 = b.wp_mu + static_cast<int64_t>(O) * khw * cp;
  return b;
}

template <int MODE, bool RNG>
static void launch_conv(const FwdPlan& p, hipStream_t s, const float* a, const float* a2, const float* wm, const float* ws,
                        const float* b1, const float* b2, const float* eps, uint64_t seed, uint64_t stream_id, float* out,
                        float* var_out, const ConvGeo& g) {
#define BDE_CONV_CASE(MF_, PT_)                                                                                         \
  if (p.mf == MF_ && p.pt == PT_) {                                                                                    \
    hipLaunchKernelGGL((conv_lrt_kernel<MF_, PT_, RNG, MODE>), p.grid, dim3(256), p.lds, s, a, a2, wm, ws, b1, b2, eps, seed, \
                       stream_id, out, var_out, g, p.t);                                                               \
    return;                                                                                                            \
  }
  BDE_CONV_CASE(16, 1) BDE_CONV_CASE(16, 2) BDE_CONV_CASE(16, 4) BDE_CONV_CASE(16, 8)
  BDE_CONV_CASE(32, 1) BDE_CONV_CASE(32, 2) BDE_CONV_CASE(32, 4)
#undef BDE_CONV_CASE
}

}  // namespace

extern "C" int bde_conv_lrt_supported(int N, int C, int H, int W, int O, int KH, int KW, int sh, int sw, int ph, int pw) {
  ConvGeo g, d;
  FwdPlan p;
  return layer_geo(N, C, H, W, O, KH, KW, sh, sw, ph, pw, g) && plan_fwd(g, p) && data_grad_geo(g, d) && plan_fwd(d, p) &&
                 bde_conv_lrt_bwd_weight_ws_bytes(N, C, H, W, O, KH, KW, sh, sw, ph, pw) != 0   // ... and for the weight gradient
             ? 1
             : 0;
}

// The tiling the kernels would run with (tests / tools: tests/conv_emulator.py replays the kernels' index arithmetic on the
// CPU with exactly these numbers).  which = 0: forward, 1: input gradient.  out[16]: MF, PT, NI, TH, bands, CC, PH, PWP, WP,
// WK, tiles_per_img, kcpad_max, grid.x, grid.y, LDS bytes, 0.  Returns 0, or BDE_ERR_INVALID for an unsupported geometry.
extern "C" int bde_conv_lrt_plan(int which, int N, int C, int H, int W, int O, int KH, int KW, int sh, int sw, int ph, int pw,
                                 int* out) {
  ConvGeo l, g;
  FwdPlan p;
  if (!out || !layer_geo(N, C, H, W, O, KH, KW, sh, sw, ph, pw, l)) return BDE_ERR_INVALID;
  g = l;
  if (which == 1 && !data_grad_geo(l, g)) return BDE_ERR_INVALID;
  if (!plan_fwd(g, p)) return BDE_ERR_INVALID;
  const int v[16] = {p.mf, p.pt, p.t.NI, p.t.TH, p.t.bands, p.t.CC, p.t.PH, p.t.PWP, p.t.WP, p.t.WK, p.t.tiles_per_img,
                     p.t.kcpad_max, static_cast<int>(p.grid.x), static_cast<int>(p.grid.y), static_cast<int>(p.lds), 0};
  for (int i = 0; i < 16; ++i) out[i] = v[i];
  return 0;
}

// ---- tuning hooks (tools/conv_autotune.py, tests): the launch geometries of a pass, the candidate tilings of a launch
// geometry, pinning one.  A launch geometry is what ONE launch of conv_lrt_kernel convolves:
// geo[15] = N, C, H, W, O, KH, KW, sh, sw, ph, pw, dh, dw, Ho, Wo (dh, dw: dilation of the input image, > 1 only in the dilated
// input-gradient pass; Ho, Wo: the output extent the launch writes -- the layer's input extent in the gradient passes, which
// can exceed what the padded input alone would give).
static bool geo_from_key(const int* k, ConvGeo& g) {
  if (!k) return false;
  for (int i = 0; i < 15; ++i)
    if (k[i] < (i == 9 || i == 10 ? -64 : 1)) return false;    // (a phase's top / left padding can be negative: its first taps lie past the edge)
  if (k[5] > 7 || k[6] > 7) return false;
  g = ConvGeo{k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], k[8], k[9], k[10], k[13], k[14], k[11], k[12], pad32(k[4])};
  g.OH = k[13];
  g.OW = k[14];
  return true;
}

extern "C" int bde_conv_lrt_pass_geos(int which, const int* layer, int* out, int max) {
  ConvGeo l, g;
  if (!layer || !out || max < 1 ||
      !layer_geo(layer[0], layer[1], layer[2], layer[3], layer[4], layer[5], layer[6], layer[7], layer[8], layer[9], layer[10], l))
    return BDE_ERR_INVALID;
  int n = 0;
  auto emit = [&](const ConvGeo& q) {
    if (n < max) {
      const GeoKey k = geo_key(q);
      for (int i = 0; i < 15; ++i) out[15 * n + i] = k[i];
    }
    ++n;
  };
  if (which == 0) {
    emit(l);
  } else if (which == 1) {
    if (!data_grad_geo(l, g)) return BDE_ERR_INVALID;
    emit(g);
  } else if (which == 2) {
    if (l.ph > l.KH - 1 || l.pw > l.KW - 1) return BDE_ERR_INVALID;
    for (int a = 0; a < l.sh; ++a)
      for (int b = 0; b < l.sw; ++b) {
        int64_t off;
        if (phase_geo(l, a, b, g, off)) emit(g);
      }
  } else {
    return BDE_ERR_INVALID;
  }
  return n;
}

// out[i][6] = WK, TH, NI, CC, PT, LDS bytes; *chosen = the index the planner runs (pinned or best score).  Returns the number
// of candidates (possibly > max: only the first max are written), 0 when the geometry has no tiling.
extern "C" int bde_conv_lrt_candidates(const int* geo, int* out, int max, int* chosen) {
  ConvGeo g;
  if (!geo_from_key(geo, g) || (max > 0 && !out)) return BDE_ERR_INVALID;
  std::vector<FwdCand> cands;
  fwd_candidates(g, cands);
  for (int i = 0; i < static_cast<int>(cands.size()) && i < max; ++i) {
    const FwdCand& c = cands[i];
    const int v[6] = {c.t.WK, c.t.TH, c.t.NI, c.t.CC, c.pt, static_cast<int>(c.lds)};
    for (int j = 0; j < 6; ++j) out[6 * i + j] = v[j];
  }
  if (chosen) {
    FwdPlan p;
    int idx = -1;
    *chosen = plan_fwd(g, p, &idx) ? idx : -1;
  }
  return static_cast<int>(cands.size());
}

// Pin the tiling (WK, TH, NI, CC) of a launch geometry for this process; wk = 0 removes the pin.  BDE_ERR_INVALID when the tiling
// is not one of the geometry's candidates (nothing is pinned then).
extern "C" int bde_conv_lrt_set_tiling(const int* geo, int wk, int th, int ni, int cc) {
  ConvGeo g;
  if (!geo_from_key(geo, g)) return BDE_ERR_INVALID;
  std::lock_guard<std::mutex> lock(pin_mutex());
  if (wk == 0) {
    pins().erase(geo_key(g));
    return 0;
  }
  std::vector<FwdCand> cands;
  fwd_candidates(g, cands);
  for (const FwdCand& c : cands)
    if (c.t.WK == wk && c.t.TH == th && c.t.NI == ni && c.t.CC == cc) {
      pins()[geo_key(g)] = {wk, th, ni, cc};
      return 0;
    }
  return BDE_ERR_INVALID;
}

extern "C" size_t bde_conv_lrt_prep_floats(int O, int C, int KH, int KW) {
  if (O < 1 || C < 1 || KH < 1 || KW < 1) return 0;
  const size_t khw = static_cast<size_t>(KH) * KW, ktot = khw * C;
  return 2 * ktot * pad32(O) + 4 * static_cast<size_t>(O) * khw * pad32(C) + ((static_cast<size_t>(O) * ktot + 3) & ~size_t{3}) + pad32(O);
}

extern "C" int bde_conv_lrt_prep_strided(const float* w_mu, const float* w_rho, const float* b_rho, int O, int C, int KH, int KW,
                                         int sh, int sw, int ph, int pw, float* wbuf, void* stream) {
  if (!w_mu || !w_rho || !wbuf || !aligned16(wbuf) || bde_conv_lrt_prep_floats(O, C, KH, KW) == 0 || sh < 1 || sw < 1 || ph < 0 ||
      pw < 0)
    return BDE_ERR_INVALID;
  const int64_t n = static_cast<int64_t>(O) * C * KH * KW;                    // n >= O: the bias entries ride along
  hipLaunchKernelGGL(conv_lrt_prep_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), w_mu, w_rho,
                     b_rho, O, C, KH, KW, pad32(O), pad32(C), sh, sw, ph, pw, wbuf);
  return to_err(hipGetLastError());
}

extern "C" int bde_conv_lrt_prep(const float* w_mu, const float* w_rho, const float* b_rho, int O, int C, int KH, int KW,
                                 float* wbuf, void* stream) {
  return bde_conv_lrt_prep_strided(w_mu, w_rho, b_rho, O, C, KH, KW, 1, 1, 0, 0, wbuf, stream);
}

extern "C" int bde_conv_lrt_fwd(const float* x, const float* wbuf, const float* b_mu, int has_bias_var, const float* eps,
                                uint64_t seed, uint64_t stream_id, float* out, float* var_out, int N, int C, int H, int W,
                                int O, int KH, int KW, int sh, int sw, int ph, int pw, void* stream) {
  ConvGeo g;
  FwdPlan p;
  if (!x || !wbuf || !out || !layer_geo(N, C, H, W, O, KH, KW, sh, sw, ph, pw, g) || !plan_fwd(g, p))     // var_out may be NULL
    return BDE_ERR_INVALID;
  if (!aligned16(wbuf)) return BDE_ERR_INVALID;           // out / var_out / eps: 16-byte alignment only selects the float4 epilogue
  const WBuf w = wbuf_parts(wbuf, O, C, KH * KW);
  const float* b_var = has_bias_var ? w.bvar : nullptr;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (eps) launch_conv<0, false>(p, s, x, nullptr, w.wt_mu, w.wt_s2, b_mu, b_var, eps, seed, stream_id, out, var_out, g);
  else launch_conv<0, true>(p, s, x, nullptr, w.wt_mu, w.wt_s2, b_mu, b_var, eps, seed, stream_id, out, var_out, g);
  return to_err(hipGetLastError());
}

extern "C" int bde_conv_lrt_bwd_data(const float* g_out, const float* g_var, const float* wbuf, const float* x, float* g_x,
                                     int N, int C, int H, int W, int O, int KH, int KW, int sh, int sw, int ph, int pw,
                                     void* stream) {
  ConvGeo l, g;
  FwdPlan p;
  if (!g_out || !g_var || !wbuf || !x || !g_x || !layer_geo(N, C, H, W, O, KH, KW, sh, sw, ph, pw, l) || !data_grad_geo(l, g) ||
      !plan_fwd(g, p))
    return BDE_ERR_INVALID;
  if (!aligned16(wbuf)) return BDE_ERR_INVALID;
  const WBuf w = wbuf_parts(wbuf, O, C, KH * KW);
  launch_conv<1, false>(p, static_cast<hipStream_t>(stream), g_out, g_var, w.wb_mu, w.wb_s2, x, nullptr, nullptr, 0, 0, g_x, nullptr, g);
  return to_err(hipGetLastError());
}

extern "C" int bde_conv_lrt_bwd_data_phases(const float* g_out, const float* g_var, const float* wbuf, const float* x, float* g_x,
                                            int N, int C, int H, int W, int O, int KH, int KW, int sh, int sw, int ph, int pw,
                                            void* stream) {
  ConvGeo l, g;
  if (!g_out || !g_var || !wbuf || !x || !g_x || !layer_geo(N, C, H, W, O, KH, KW, sh, sw, ph, pw, l) || !aligned16(wbuf))
    return BDE_ERR_INVALID;
  if (sh == 1 && sw == 1) return bde_conv_lrt_bwd_data(g_out, g_var, wbuf, x, g_x, N, C, H, W, O, KH, KW, sh, sw, ph, pw, stream);
  // every phase needs a tiling; otherwise the dilated pass does the whole job
  FwdPlan plans[49];
  ConvGeo geos[49];
  int64_t offs[49];
  bool live[49];
  bool zero_fill = false;
  for (int a = 0; a < sh; ++a)
    for (int b = 0; b < sw; ++b) {
      const int q = a * sw + b;
      if (q >= 49) return bde_conv_lrt_bwd_data(g_out, g_var, wbuf, x, g_x, N, C, H, W, O, KH, KW, sh, sw, ph, pw, stream);
      live[q] = phase_geo(l, a, b, geos[q], offs[q]);
      if (!live[q]) {
        const PhaseAxis pa = phase_axis(a, KH, sh, ph, H), pb = phase_axis(b, KW, sw, pw, W);
        zero_fill = zero_fill || (pa.count > 0 && pb.count > 0);        // pixels without a tap: exactly zero
      } else if (!plan_fwd(geos[q], plans[q])) {
        return bde_conv_lrt_bwd_data(g_out, g_var, wbuf, x, g_x, N, C, H, W, O, KH, KW, sh, sw, ph, pw, stream);
      }
    }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (zero_fill) {
    const hipError_t e = hipMemsetAsync(g_x, 0, sizeof(float) * static_cast<size_t>(N) * C * H * W, s);
    if (e != hipSuccess) return to_err(e);
  }
  const WBuf w = wbuf_parts(wbuf, O, C, KH * KW);
  for (int q = 0; q < sh * sw; ++q)
    if (live[q])
      launch_conv<1, false>(plans[q], s, g_out, g_var, w.wp_mu + offs[q], w.wp_s2 + offs[q], x, nullptr, nullptr, 0, 0, g_x, nullptr,
                            geos[q]);
  return to_err(hipGetLastError());
}

extern "C" __attribute__((visibility("hidden"))) int bde_internal_load_conv_lrt(void) {
  hipFuncAttributes attr;
  return bde::to_err(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&bde::conv_lrt_kernel<32, 4, true, 0>)));
}
