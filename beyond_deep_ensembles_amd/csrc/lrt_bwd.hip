// Backward of the fused local-reparameterisation linear layer (forward: lrt.hip).
//
// Reference: the autograd graph of BBBLinear.forward, sampling="activations" (src/algos/bbb_layers.py:61-80):
//   mean = x W_mu^T + b_mu,  var = clamp(x^2, 1e-4) clamp(sigma_W^2, 1e-4)^T + clamp(sigma_b^2, 1e-4),
//   out  = mean + sqrt(var) * eps,  sigma = softplus(rho)
// which autograd runs as ~25 ATen launches re-reading sigma, sigma^2, its clamp mask, x^2 and its mask.  With
// g = d loss / d out and gvar = g * eps / (2 sqrt(var)):
//   g_x     = g W_mu + (gvar clamp(sigma_W^2)) * 2 x * [x^2 >= 1e-4]
//   g_Wmu   = g^T x
//   g_Wrho  = (gvar^T clamp(x^2)) * [sigma_W^2 >= 1e-4] * 2 sigma_W sigmoid(rho_W)
//   g_bmu   = sum_b g,   g_brho = (sum_b gvar) * [sigma_b^2 >= 1e-4] * 2 sigma_b sigmoid(rho_b)
// Two launches for layers up to 2^20 weights (prep + lrt_bwd_fused_kernel: one pass over the weights for all three
// matrix gradients), otherwise three (four when the reduction over O is split):
//   lrt_bwd_prep_kernel   gvar [B, O] once (noise supplied or regenerated from the forward's Philox stream) and the
//                         two bias gradients (column sums in a fixed order)
//   lrt_bwd_w_kernel      one wave per 32 x 32 tile of [O, I]: both weight gradients as two accumulator tiles on
//                         v_mfma_f32_32x32x2_f32 over K = B, the rho chain rule applied in the C layout (rows of
//                         W_rho read and rows of the gradients written as 128-byte segments)
//   lrt_bwd_x_kernel      one wave per (32 input columns, O-slice): streams rows of W_mu / W_rho (128-byte segments
//                         per row), sigma^2 on the fly, for up to 4 batch tiles; O-slice partials + fixed-order finish
//                         only when the layer is large enough to need the split
// HBM traffic 20*O*I (W_rho, W_mu + W_rho again, two gradients written) + O(B (I + O)).
#include "bde_common.hpp"

namespace bde {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// tools/lrt_ab.py only (never defined in the product build): where the time of the wide-layer kernels goes.
//   BDE_EXP_NOMFMA   the matrix products replaced by one VALU FMA each (memory + LDS + issue side alone)
//   BDE_EXP_ALIAS    every tile reads / writes the FIRST tile's rows of the weight-shaped arrays (no HBM stream)
#ifdef BDE_EXP_NOMFMA
#define BDE_MFMA32(a, b, c) ([&] { auto c_ = (c); c_[0] = __builtin_fmaf((a), (b), c_[0]); return c_; }())
#else
#define BDE_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#endif
#ifdef BDE_EXP_ALIAS
#define BDE_EXP_ROW(r) ((r) & 31)
#else
#define BDE_EXP_ROW(r) (r)
#endif

constexpr float kLrtBwdClamp = 1e-4f;   // bbb_layers.py:66-67,71
constexpr int kLrtBwdWaves = 4;         // waves per workgroup (independent units)
#ifndef BDE_LRT_BWD_WU
#define BDE_LRT_BWD_WU 8
#endif
#ifndef BDE_LRT_BWD_XU
#define BDE_LRT_BWD_XU 4
#endif
constexpr int kPrepCols = 32, kPrepRows = 8, kPrepMaxB = 128;

// gvar [B, O] (for the weight-gradient kernel: lanes along o), the transposed copies gT / gvarT [O, b_pad] (for the
// input-gradient kernel: lanes along b; rows b >= B are written as zeros, so that kernel needs no batch masks) and
// the bias gradients.  One workgroup per 32 output columns; the transposition goes through LDS so that both the
// [B, O] reads and the [O, b_pad] writes are contiguous.
template <bool RNG>
__global__ __launch_bounds__(kPrepCols* kPrepRows) void lrt_bwd_prep_kernel(
    const float* __restrict__ g, const float* __restrict__ var, const float* __restrict__ eps, uint64_t seed,
    uint64_t stream_id, const float* __restrict__ b_rho, int clamp_bias, float* __restrict__ gvar,
    float* __restrict__ gT, float* __restrict__ gvT, float* __restrict__ g_bmu, float* __restrict__ g_brho, int B,
    int b_pad, int O) {
  __shared__ float tile_g[kPrepMaxB][kPrepCols + 1], tile_v[kPrepMaxB][kPrepCols + 1];
  __shared__ float red_g[kPrepRows][kPrepCols], red_v[kPrepRows][kPrepCols];
  const int c = threadIdx.x % kPrepCols, q = threadIdx.x / kPrepCols;
  const int o0 = blockIdx.x * kPrepCols, o = o0 + c;
  float sg = 0.f, sv = 0.f;
  for (int b = q; b < b_pad; b += kPrepRows) {
    float gg = 0.f, gv = 0.f;
    if (b < B && o < O) {
      const int64_t e = static_cast<int64_t>(b) * O + o;
      const float z = RNG ? philox_normal4(seed, stream_id, static_cast<uint64_t>(e >> 2), kDomainDiag)[e & 3] : eps[e];
      gg = g[e];
      gv = gg * z / (2.0f * __builtin_sqrtf(var[e]));
      gvar[e] = gv;
    }
    tile_g[b][c] = gg;
    tile_v[b][c] = gv;
    sg += gg;
    sv += gv;
  }
  red_g[q][c] = sg;
  red_v[q][c] = sv;
  __syncthreads();
  if (gT) {
    for (int idx = threadIdx.x; idx < kPrepCols * b_pad; idx += kPrepCols * kPrepRows) {
      const int cc = idx / b_pad, b = idx - cc * b_pad;
      if (o0 + cc < O) {
        gT[static_cast<int64_t>(o0 + cc) * b_pad + b] = tile_g[b][cc];
        gvT[static_cast<int64_t>(o0 + cc) * b_pad + b] = tile_v[b][cc];
      }
    }
  }
  if (q == 0 && o < O && g_bmu) {
    float tg = 0.f, tv = 0.f;
#pragma unroll
    for (int k = 0; k < kPrepRows; ++k) {                          // fixed order
      tg += red_g[k][c];
      tv += red_v[k][c];
    }
    const SoftplusSigmoid s = softplus_sigmoid(b_rho[o]);
    const float keep = (!clamp_bias || s.sp * s.sp >= kLrtBwdClamp) ? 1.f : 0.f;
    g_bmu[o] = tg;
    g_brho[o] = tv * keep * (2.0f * s.sp * s.sg);
  }
}

// Both weight gradients of one 32 x 32 tile of [O, I]: D[m = o][n = i] = sum_b A[o][b] B[b][i].
// PRE: `w_rho` is the cached chain-rule factor [sigma^2 >= 1e-4] * 2 sigma sigmoid(rho) (bde_lrt_sigma_cache).
template <bool PRE>
__global__ __launch_bounds__(kLrtBwdWaves * 64) void lrt_bwd_w_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ w_rho, const float* __restrict__ g,
    const float* __restrict__ gvar, int B, int I, int O, float* __restrict__ g_wmu, float* __restrict__ g_wrho) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i_tiles = (I + 31) >> 5, o_tiles = (O + 31) >> 5;
  const int unit = blockIdx.x * kLrtBwdWaves + wave;
  if (unit >= i_tiles * o_tiles) return;
  const int ot = unit / i_tiles, it = unit % i_tiles;              // neighbouring waves: neighbouring column tiles
  const int r = lane & 31, h = lane >> 5;
  const int o = ot * 32 + r, i = it * 32 + r;
  const bool o_ok = o < O, i_ok = i < I;
#ifdef BDE_EXP_ALIAS_OPS   // tools/lrt_ab.py only: every tile reads the FIRST tile's operands (L1 hits instead of L2 traffic)
  const int oc = r, ic = r;
#else
  const int oc = o_ok ? o : O - 1, ic = i_ok ? i : I - 1;         // clamped: every load is unconditional
#endif
  // the rho rows of the epilogue (C layout: lane = column i, rows o = ot*32 + (reg & 3) + 8 (reg >> 2) + 4 h) are
  // requested first and arrive while the products run
  float rho[16];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int oo = BDE_EXP_ROW(min(ot * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h, O - 1));
    rho[reg] = w_rho[static_cast<int64_t>(oo) * I + ic];
  }
  constexpr int U = BDE_LRT_BWD_WU;                                // k-steps (of 2 batch rows) per operand set
  struct Operands {
    float ag[U], av[U], xv[U];
  };
  auto load = [&](Operands& q, int s0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int bc = min(2 * (s0 + u) + h, B - 1);
      q.ag[u] = g[static_cast<int64_t>(bc) * O + oc];
      q.av[u] = gvar[static_cast<int64_t>(bc) * O + oc];
      q.xv[u] = x[static_cast<int64_t>(bc) * ldx + ic];
    }
  };
  f32x16 accm = {}, accv = {};
  const int steps = (B + 1) >> 1;
  Operands cur, nxt;
  load(cur, 0);
  for (int s0 = 0; s0 < steps; s0 += U) {
    if (s0 + U < steps) load(nxt, s0 + U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool b_ok = 2 * (s0 + u) + h < B;                      // rows past B: clamped loads, zero operands
      const float xv = cur.xv[u];
      const float bx = (b_ok && i_ok) ? xv : 0.f, bx2 = (b_ok && i_ok) ? fmaxf(xv * xv, kLrtBwdClamp) : 0.f;
      accm = BDE_MFMA32(o_ok ? cur.ag[u] : 0.f, bx, accm);
      accv = BDE_MFMA32(o_ok ? cur.av[u] : 0.f, bx2, accv);
    }
    cur = nxt;
  }
  if (!i_ok) return;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int oo = ot * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (oo < O) {
      const int64_t idx = static_cast<int64_t>(BDE_EXP_ROW(oo)) * I + i;
      g_wmu[idx] = accm[reg];
      if (PRE) {
        g_wrho[idx] = accv[reg] * rho[reg];
      } else {
        const SoftplusSigmoid sp = softplus_sigmoid(rho[reg]);
        const float keep = sp.sp * sp.sp >= kLrtBwdClamp ? 1.f : 0.f;
        g_wrho[idx] = accv[reg] * keep * (2.0f * sp.sp * sp.sg);
      }
    }
  }
}

// Input gradient of (NB batch tiles, 32 input columns) over one O-slice: D[m = b][n = i] = sum_o A[b][o] B[o][i], A
// from the transposed copies (lanes along b: one 128-byte segment per k), B = rows of W_mu / clamp(sigma_W^2)
// (lanes along i).  Register double buffering: the loads of the next 4 k-steps are issued before the softplus / MFMA
// work of the current ones.  DIRECT: the slice is all of O and the epilogue writes g_x; otherwise the two partial
// tiles go to the workspace.
// PRE: `w_rho` is the cached clamp(softplus(rho)^2, 1e-4).
template <int NB, bool DIRECT, bool PRE>
__global__ __launch_bounds__(kLrtBwdWaves * 64) void lrt_bwd_x_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ w_mu, const float* __restrict__ w_rho,
    const float* __restrict__ gT, const float* __restrict__ gvT, int B, int I, int O, int n_slices, int oslice,
    float* __restrict__ g_x, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i_tiles = (I + 31) >> 5;
  const int unit = blockIdx.x * kLrtBwdWaves + wave;
  if (unit >= i_tiles * n_slices) return;
  const int sl = unit / i_tiles, it = unit % i_tiles;
  const int r = lane & 31, h = lane >> 5;
  const int i = it * 32 + r;
  const bool i_ok = i < I;
  const int ic = i_ok ? i : I - 1;
  const int o0 = sl * oslice, o1 = min(O, o0 + oslice);
  constexpr int b_pad = NB * 32;
  constexpr int U = NB <= 2 ? BDE_LRT_BWD_XU : 4;                  // k-steps (of 2 rows of W) per operand set
  struct Operands {
    float wm[U], wr[U], ag[U][NB], av[U][NB];
  };
  auto load = [&](Operands& q, int ob0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int oc = min(ob0 + 2 * u + h, o1 - 1);
      q.wm[u] = w_mu[static_cast<int64_t>(BDE_EXP_ROW(oc)) * I + ic];
      q.wr[u] = w_rho[static_cast<int64_t>(BDE_EXP_ROW(oc)) * I + ic];
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        q.ag[u][t] = gT[static_cast<int64_t>(oc) * b_pad + t * 32 + r];
        q.av[u][t] = gvT[static_cast<int64_t>(oc) * b_pad + t * 32 + r];
      }
    }
  };
  f32x16 accm[NB], accv[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) accm[t] = accv[t] = f32x16{};
  Operands cur, nxt;
  load(cur, o0);
  for (int ob0 = o0; ob0 < o1; ob0 += 2 * U) {
    if (ob0 + 2 * U < o1) load(nxt, ob0 + 2 * U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool w_ok = (ob0 + 2 * u + h < o1) && i_ok;           // rows past the slice / columns past I add zeros
      float s2;
      if (PRE) {
        s2 = cur.wr[u];
      } else {
        const float sg = softplus(cur.wr[u]);
        s2 = fmaxf(sg * sg, kLrtBwdClamp);
      }
      const float bm = w_ok ? cur.wm[u] : 0.f, bv = w_ok ? s2 : 0.f;
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        accm[t] = BDE_MFMA32(cur.ag[u][t], bm, accm[t]);
        accv[t] = BDE_MFMA32(cur.av[u][t], bv, accv[t]);
      }
    }
    cur = nxt;
  }
  if (!i_ok) return;
#pragma unroll
  for (int t = 0; t < NB; ++t) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int b = t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (DIRECT) {
        if (b < B) {
          const float xv = x[static_cast<int64_t>(b) * ldx + i];
          const float keep = xv * xv >= kLrtBwdClamp ? 1.f : 0.f;
          g_x[static_cast<int64_t>(b) * I + i] = accm[t][reg] + accv[t][reg] * keep * (2.0f * xv);
        }
      } else {
        float* base = part + static_cast<int64_t>(sl) * 2 * b_pad * I;
        base[static_cast<int64_t>(b) * I + i] = accm[t][reg];
        base[static_cast<int64_t>(b_pad + b) * I + i] = accv[t][reg];
      }
    }
  }
}

// The same input gradient with 16-byte loads (I % 4 == 0, O % 4 == 0; up to 64 batch rows).  lrt_bwd_x_kernel issues six
// 4-byte loads per pair of weight rows; here a wave walks its O-slice in chunks of 16 weight rows:
//   * W_mu / sigma^2 rows arrive as float4s, 8 rows x 128 bytes per instruction, and are parked in a per-wave LDS tile;
//     the B operand (lane = input column, one row per k) is read back with conflict-free 4-byte LDS reads;
//   * the A operand comes straight from g / gvar [B, O] as float4s along o (no transposed copies): k-step (j, c) of a
//     chunk multiplies rows 8 j + c (lanes 0-31) and 8 j + 4 + c (lanes 32-63);
//   * the next chunk's loads are in flight while the current chunk's 16 NB products run.
// 12 memory instructions per 16 rows instead of 48 (NB = 2), and the prep kernel skips the transposed copies.  Worth
// 2-5 % of the backward at 4096 x 4096 (profiles/r03_lrt_limiter_experiments.txt: the wide kernels are bound by how
// well operand waits and the fp32 matrix pipe overlap, not by the count of memory instructions).  Same partial /
// finish scheme as lrt_bwd_x_kernel.
constexpr int kX4Rows = 16, kX4Ld = 36;
#ifndef BDE_LRT_X4_FULL_ROWS
#define BDE_LRT_X4_FULL_ROWS 16
#endif
constexpr int kX4FullRows = BDE_LRT_X4_FULL_ROWS;   // FULL variant: rows per chunk (two chunks per trip of its loop)
#ifndef BDE_LRT_BWD_X4
#define BDE_LRT_BWD_X4 1
#endif
#ifndef BDE_LRT_X4_WAVES
#define BDE_LRT_X4_WAVES 2      // waves per SIMD the register allocation aims for
#endif

// FULL: B % 32 == 0, I % 32 == 0, O % 32 == 0 (slices are whole 32-row tiles): no padding anywhere, and the product loop
// is written so that it carries next to no vector instructions besides the products -- on this chip they do not run in
// the shadow of the fp32 MFMA, each costs 6-10 % of a product (tools/mfma_rate.hip): every address is a wave-uniform base
// (scalar unit) + a per-lane 32-bit offset computed once, no masks, no register copies (two operand sets alternate).
template <int NB, bool DIRECT, bool PRE, bool FULL>
__global__ __launch_bounds__(kLrtBwdWaves * 64, FULL ? 1 : BDE_LRT_X4_WAVES) void lrt_bwd_x4_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ w_mu, const float* __restrict__ w_rho,
    const float* __restrict__ g, const float* __restrict__ gvar, int B, int I, int O, int n_slices, int oslice,
    float* __restrict__ g_x, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float tiles[kLrtBwdWaves][2 * (FULL ? kX4FullRows : kX4Rows) * kX4Ld];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i_tiles = (I + 31) >> 5;
  // (readfirstlane: the wave index is uniform, but only this tells the compiler -- slice bounds, the loop counter and
  // the row offsets then live in scalar registers)
  const int unit = __builtin_amdgcn_readfirstlane(blockIdx.x * kLrtBwdWaves + wave);
  if (unit >= i_tiles * n_slices) return;                          // (no workgroup barrier below: waves are independent)
  const int sl = unit / i_tiles, it = unit % i_tiles;
  const int r = lane & 31, h = lane >> 5;
  const int lr = lane >> 3, lc = 4 * (lane & 7);                   // row layout of the W loads: row 8 p + lr, columns lc .. lc + 3
  const int i = it * 32 + r;
  const bool i_ok = i < I;
  const int colw = min(it * 32 + lc, I - 4);                       // columns past I: any valid address (never stored)
  const int o0 = sl * oslice, o1 = min(O, o0 + oslice);
  constexpr int b_pad = NB * 32;
  float* tile = tiles[wave];
  bool b_ok[NB];
  int64_t arow[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    b_ok[t] = t * 32 + r < B;                                      // rows past B: clamped loads, zero operands
    arow[t] = static_cast<int64_t>(min(t * 32 + r, B - 1)) * O;
  }

  f32x16 accm[NB], accv[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) accm[t] = accv[t] = f32x16{};
  if constexpr (FULL) {
    constexpr int RF = kX4FullRows, PW = RF / 8;                     // rows per chunk, 8-row groups per chunk
    struct WOps {
      f32x4 wm[PW], wr[PW];
    };
    struct AOps {
      f32x4 ag[NB][PW], av[NB][PW];
    };
    // per-lane offsets (elements), computed once; the chunk's row offset is wave-uniform
    const unsigned w_off = static_cast<unsigned>(lr) * static_cast<unsigned>(I) + static_cast<unsigned>(it * 32 + lc);
    unsigned a_off[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) a_off[t] = static_cast<unsigned>(t * 32 + r) * static_cast<unsigned>(O) + 4u * h;
    const int o_last = o1 - RF;                               // requests past the slice re-read its last chunk (unused)
    auto request_w = [&](WOps& q, int oc0) {
      oc0 = min(oc0, o_last);
#pragma unroll
      for (int p = 0; p < PW; ++p) {
        const int64_t rows = static_cast<int64_t>(BDE_EXP_ROW(oc0 + 8 * p)) * I;       // uniform
        q.wm[p] = ld4(&(w_mu + rows)[w_off]);                                           // scalar base + 32-bit lane offset
        q.wr[p] = ld4(&(w_rho + rows)[w_off]);
      }
    };
    auto request_a = [&](AOps& q, int oc0) {
      oc0 = min(oc0, o_last);
#pragma unroll
      for (int j = 0; j < PW; ++j)
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          q.ag[t][j] = ld4(&(g + (oc0 + 8 * j))[a_off[t]]);
          q.av[t][j] = ld4(&(gvar + (oc0 + 8 * j))[a_off[t]]);
        }
    };
    auto park = [&](const WOps& q) {
#pragma unroll
      for (int p = 0; p < PW; ++p) {
        *reinterpret_cast<f32x4*>(tile + (8 * p + lr) * kX4Ld + lc) = q.wm[p];
        if (PRE) {
          *reinterpret_cast<f32x4*>(tile + (RF + 8 * p + lr) * kX4Ld + lc) = q.wr[p];
        } else {
          f32x4 v;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float sg = softplus(q.wr[p][c]);
            v[c] = fmaxf(sg * sg, kLrtBwdClamp);
          }
          *reinterpret_cast<f32x4*>(tile + (RF + 8 * p + lr) * kX4Ld + lc) = v;
        }
      }
    };
    auto products = [&](const AOps& q) {
#pragma unroll
      for (int j = 0; j < PW; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int krow = 8 * j + 4 * h + c;
          const float bm = tile[krow * kX4Ld + r], bv = tile[(RF + krow) * kX4Ld + r];
#pragma unroll
          for (int t = 0; t < NB; ++t) {
            accm[t] = BDE_MFMA32(q.ag[t][j][c], bm, accm[t]);
            accv[t] = BDE_MFMA32(q.av[t][j][c], bv, accv[t]);
          }
        }
    };
    // Weight rows (HBM) are requested TWO chunks ahead, the A operands (g / gvar: 1 MB each, L2) one chunk ahead; the
    // sched_barriers keep the compiler from sinking a request below the stash that waits for the previous one.
    auto step = [&](WOps& w, AOps& a_now, AOps& a_next, int oc0) {
      park(w);                                                     // chunk oc0's weight rows -> LDS tile
      __builtin_amdgcn_sched_barrier(0);
      request_w(w, oc0 + 2 * RF);
      request_a(a_next, oc0 + RF);
      __builtin_amdgcn_sched_barrier(0);
      products(a_now);
      __builtin_amdgcn_sched_barrier(0);
    };
    WOps w0, w1;
    AOps a0, a1;
    // (the prologue issues its requests in the order a trip leaves them outstanding -- w0, a1, w1, a0 -- so that the
    // waits at the top of the loop can be counted instead of "everything"; a1's first request is a dummy)
    request_w(w0, o0);
    request_a(a1, o0);
    request_w(w1, o0 + RF);
    request_a(a0, o0);
    for (int oc0 = o0; oc0 < o1; oc0 += 2 * RF) {             // slices are whole 32-row tiles: two chunks per trip
      step(w0, a0, a1, oc0);
      step(w1, a1, a0, oc0 + RF);
    }
  } else {
  struct Stage {
    f32x4 wm[2], wr[2], ag[NB][2], av[NB][2];
  };
  auto gload = [&](Stage& q, int oc0) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int64_t row = static_cast<int64_t>(BDE_EXP_ROW(min(oc0 + 8 * p + lr, O - 1))) * I + colw;
      q.wm[p] = ld4(w_mu + row);
      q.wr[p] = ld4(w_rho + row);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int oc = min(oc0 + 8 * j + 4 * h, O - 4);              // past the slice: the W rows are zero, any finite A will do
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        q.ag[t][j] = ld4(g + arow[t] + oc);
        q.av[t][j] = ld4(gvar + arow[t] + oc);
      }
    }
  };
  auto stash = [&](const Stage& q, int oc0) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const bool ok = oc0 + 8 * p + lr < o1;                       // rows past the slice add zeros
      f32x4 m, v;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        m[c] = ok ? q.wm[p][c] : 0.f;
        if (PRE) {
          v[c] = ok ? q.wr[p][c] : 0.f;
        } else {
          const float sg = softplus(q.wr[p][c]);
          v[c] = ok ? fmaxf(sg * sg, kLrtBwdClamp) : 0.f;
        }
      }
      *reinterpret_cast<f32x4*>(tile + (8 * p + lr) * kX4Ld + lc) = m;
      *reinterpret_cast<f32x4*>(tile + (kX4Rows + 8 * p + lr) * kX4Ld + lc) = v;
    }
  };
  Stage st;
  gload(st, o0);
  for (int oc0 = o0; oc0 < o1; oc0 += kX4Rows) {
    stash(st, oc0);
    f32x4 ag[NB][2], av[NB][2];
#pragma unroll
    for (int t = 0; t < NB; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        ag[t][j] = st.ag[t][j];
        av[t][j] = st.av[t][j];
      }
    if (oc0 + kX4Rows < o1) gload(st, oc0 + kX4Rows);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int krow = 8 * j + 4 * h + c;
        const float bm = tile[krow * kX4Ld + r], bv = tile[(kX4Rows + krow) * kX4Ld + r];
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          accm[t] = BDE_MFMA32(b_ok[t] ? ag[t][j][c] : 0.f, bm, accm[t]);
          accv[t] = BDE_MFMA32(b_ok[t] ? av[t][j][c] : 0.f, bv, accv[t]);
        }
      }
  }
  }
  if (!i_ok) return;
#pragma unroll
  for (int t = 0; t < NB; ++t) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int b = t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (DIRECT) {
        if (b < B) {
          const float xv = x[static_cast<int64_t>(b) * ldx + i];
          const float keep = xv * xv >= kLrtBwdClamp ? 1.f : 0.f;
          g_x[static_cast<int64_t>(b) * I + i] = accm[t][reg] + accv[t][reg] * keep * (2.0f * xv);
        }
      } else {
        float* base = part + static_cast<int64_t>(sl) * 2 * b_pad * I;
        base[static_cast<int64_t>(b) * I + i] = accm[t][reg];
        base[static_cast<int64_t>(b_pad + b) * I + i] = accv[t][reg];
      }
    }
  }
}

// Weight gradients AND input gradient in ONE pass over the weights (layers up to 2^20 weights, batch <= 64): a
// wave owns 32 input columns and an O-slice and walks the slice in tiles of 32 rows.  Per tile the rows of W_mu / W_rho
// are read ONCE and softplus / sigmoid evaluated ONCE per element, serving both products:
//   dX   D[b][i] += sum_o A[b][o] B[o][i]   B = W_mu / clamp(sigma^2) rows (lanes along i), A from the transposed copies
//   dW   D[o][i]  = sum_b A[o][b] B[b][i]   A = g / gvar (lanes along o), B = x / clamp(x^2) kept in registers for the
//                                           whole slice; epilogue * [sigma^2 >= 1e-4] * 2 sigma sigmoid(rho)
// The k-order of the dX product inside a tile is free, so k-step s uses row (s & 3) + 8 (s >> 2) + 4 h -- the row that
// accumulator register s of the dW tile holds in this lane: every lane needs sigma of exactly the 16 rows it loaded.
// Against the two separate kernels: W_rho read once instead of twice (20 -> 16 bytes per weight) and half the softplus work.
template <int NB, bool DIRECT>
__global__ __launch_bounds__(kLrtBwdWaves * 64) void lrt_bwd_fused_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ w_mu, const float* __restrict__ w_rho,
    const float* __restrict__ g, const float* __restrict__ gvar, const float* __restrict__ gT,
    const float* __restrict__ gvT, int B, int I, int O, int n_slices, int oslice, float* __restrict__ g_wmu,
    float* __restrict__ g_wrho, float* __restrict__ g_x, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i_tiles = (I + 31) >> 5;
  const int unit = blockIdx.x * kLrtBwdWaves + wave;
  if (unit >= i_tiles * n_slices) return;
  const int sl = unit / i_tiles, it = unit % i_tiles;
  const int r = lane & 31, h = lane >> 5;
  const int i = it * 32 + r;
  const bool i_ok = i < I;
  const int ic = i_ok ? i : I - 1;
  const int o0 = sl * oslice, o1 = min(O, o0 + oslice);             // oslice is a multiple of 32
  constexpr int b_pad = NB * 32, KB = NB * 16;                       // k-steps of the dW product (2 batch rows each)
  // this strip of x in the dW B-operand layout (lane: column i, batch rows 2 u + h), masked once
  float xs[KB];
#pragma unroll
  for (int u = 0; u < KB; ++u) {
    const int b = 2 * u + h;
    const float v = x[static_cast<int64_t>(min(b, B - 1)) * ldx + ic];
    xs[u] = (b < B && i_ok) ? v : 0.f;
  }
  f32x16 accxm[NB], accxv[NB];
#pragma unroll
  for (int t = 0; t < NB; ++t) accxm[t] = accxv[t] = f32x16{};

  for (int ot0 = o0; ot0 < o1; ot0 += 32) {
    // rows of this tile in the accumulator-register order (see above); clamped, masked at use
    float wm[16], wr[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = min(ot0 + (s & 3) + 8 * (s >> 2) + 4 * h, O - 1);
      wm[s] = w_mu[static_cast<int64_t>(row) * I + ic];
      wr[s] = w_rho[static_cast<int64_t>(row) * I + ic];
    }
    // dW: K = B
    const int oa = ot0 + r;
    const bool oa_ok = oa < O;
    const int oac = oa_ok ? oa : O - 1;
    f32x16 accwm = {}, accwv = {};
#pragma unroll
    for (int u0 = 0; u0 < KB; u0 += 8) {
      float ag[8], av[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int bc = min(2 * (u0 + u) + h, B - 1);
        ag[u] = g[static_cast<int64_t>(bc) * O + oac];
        av[u] = gvar[static_cast<int64_t>(bc) * O + oac];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool a_ok = oa_ok && (2 * (u0 + u) + h < B);
        const float xv = xs[u0 + u];
        const float x2 = (2 * (u0 + u) + h < B && i_ok) ? fmaxf(xv * xv, kLrtBwdClamp) : 0.f;
        accwm = BDE_MFMA32(a_ok ? ag[u] : 0.f, xv, accwm);
        accwv = BDE_MFMA32(a_ok ? av[u] : 0.f, x2, accwv);
      }
    }
    // dX k-steps + the dW epilogue, 4 rows of transposed-copy operands in flight at a time
#pragma unroll
    for (int s0 = 0; s0 < 16; s0 += 4) {
      float agT[4][NB], avT[4][NB];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int s = s0 + q;
        const int row = min(ot0 + (s & 3) + 8 * (s >> 2) + 4 * h, O - 1);
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          agT[q][t] = gT[static_cast<int64_t>(row) * b_pad + t * 32 + r];
          avT[q][t] = gvT[static_cast<int64_t>(row) * b_pad + t * 32 + r];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int s = s0 + q;
        const int row = ot0 + (s & 3) + 8 * (s >> 2) + 4 * h;
        const bool w_ok = row < o1 && i_ok;                        // rows past the slice / the matrix add nothing
        const SoftplusSigmoid sp = softplus_sigmoid(wr[s]);
        const float s2 = sp.sp * sp.sp;
        const float bm = w_ok ? wm[s] : 0.f, bv = w_ok ? fmaxf(s2, kLrtBwdClamp) : 0.f;
#pragma unroll
        for (int t = 0; t < NB; ++t) {
          accxm[t] = BDE_MFMA32(agT[q][t], bm, accxm[t]);
          accxv[t] = BDE_MFMA32(avT[q][t], bv, accxv[t]);
        }
        if (w_ok) {
          const int64_t idx = static_cast<int64_t>(row) * I + i;
          g_wmu[idx] = accwm[s];
          g_wrho[idx] = accwv[s] * (s2 >= kLrtBwdClamp ? 1.f : 0.f) * (2.0f * sp.sp * sp.sg);
        }
      }
    }
  }
  if (!i_ok) return;
#pragma unroll
  for (int t = 0; t < NB; ++t) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int b = t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (DIRECT) {
        if (b < B) {
          const float xv = x[static_cast<int64_t>(b) * ldx + i];
          const float keep = xv * xv >= kLrtBwdClamp ? 1.f : 0.f;
          g_x[static_cast<int64_t>(b) * I + i] = accxm[t][reg] + accxv[t][reg] * keep * (2.0f * xv);
        }
      } else {
        float* base = part + static_cast<int64_t>(sl) * 2 * b_pad * I;
        base[static_cast<int64_t>(b) * I + i] = accxm[t][reg];
        base[static_cast<int64_t>(b_pad + b) * I + i] = accxv[t][reg];
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void lrt_bwd_x_finish_kernel(const float* __restrict__ part, int n_slices, int b_pad,
                                                                 const float* __restrict__ x, int64_t ldx,
                                                                 float* __restrict__ g_x, int B, int I) {
  const int64_t n = static_cast<int64_t>(B) * I;
  const int64_t slice_stride = static_cast<int64_t>(2) * b_pad * I;
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(e / I), i = static_cast<int>(e % I);
    const float* pm = part + static_cast<int64_t>(b) * I + i;
    const float* pv = part + static_cast<int64_t>(b_pad + b) * I + i;
    float m = 0.f, v = 0.f;
    for (int s = 0; s < n_slices; ++s) {                          // fixed order
      m += pm[s * slice_stride];
      v += pv[s * slice_stride];
    }
    const float xv = x[static_cast<int64_t>(b) * ldx + i];
    const float keep = xv * xv >= kLrtBwdClamp ? 1.f : 0.f;
    g_x[e] = m + v * keep * (2.0f * xv);
  }
}

}  // namespace bde

using namespace bde;

// Split of the reduction over O in the input-gradient kernel: none for layers with few outputs (one launch less),
// otherwise enough (column tile, O-slice) waves to fill the chip; every slice costs a [2, b_pad, I] partial.
struct LrtBwdPlan {
  int n_slices, oslice;
};
#ifndef BDE_LRT_BWD_TARGET_WAVES
#define BDE_LRT_BWD_TARGET_WAVES 2048
#endif

#ifndef BDE_LRT_BWD_X4_TARGET_WAVES
#define BDE_LRT_BWD_X4_TARGET_WAVES 1024   // lrt_bwd_x4_kernel: one wave per SIMD and half the partials beat two (A/B: 73 vs 78 us)
#endif
static inline LrtBwdPlan lrt_bwd_plan(int I, int O, int target_waves = BDE_LRT_BWD_TARGET_WAVES) {
  // slices are whole 32-row tiles (the fused kernel's dW tiles must not straddle two slices)
  const int i_tiles = (I + 31) / 32, o_tiles = (O + 31) / 32;
  if (o_tiles < 4) return LrtBwdPlan{1, o_tiles * 32};
  int want = (target_waves + i_tiles - 1) / i_tiles;
  if (want > o_tiles) want = o_tiles;
  if (want < 1) want = 1;
  const int oslice = (o_tiles + want - 1) / want * 32;
  return LrtBwdPlan{(O + oslice - 1) / oslice, oslice};
}
static inline int lrt_bwd_nb(int B) {
  const int nbt = (B + 31) / 32;
  return nbt == 3 ? 4 : nbt;
}
static inline size_t pad64(size_t n) { return (n + 63) / 64 * 64; }

// workspace: gvar [B, O] | gT [O, b_pad] | gvarT [O, b_pad] | input-gradient partials [n_slices, 2, b_pad, I]
extern "C" size_t bde_lrt_linear_bwd_ws_bytes(int B, int I, int O) {
  if (!bde_lrt_linear_supported(B, I, O)) return 0;
  const LrtBwdPlan plan = lrt_bwd_plan(I, O);
  const size_t b_pad = static_cast<size_t>(lrt_bwd_nb(B)) * 32;
  size_t floats = pad64(static_cast<size_t>(B) * O) + 2 * pad64(static_cast<size_t>(O) * b_pad);
  if (plan.n_slices > 1) floats += static_cast<size_t>(plan.n_slices) * 2 * b_pad * I;
  return sizeof(float) * floats;
}

extern "C" int bde_lrt_linear_bwd(const float* x, int64_t ldx, const float* w_mu, const float* w_rho, const float* w_s2,
                                  const float* w_ds2, const float* b_rho, int clamp_bias_var, const float* g, const float* var,
                                  const float* eps, uint64_t seed, uint64_t stream_id, float* g_x, float* g_wmu,
                                  float* g_wrho, float* g_bmu, float* g_brho, int B, int I, int O, void* ws,
                                  void* stream) {
  if (!x || !w_mu || !w_rho || !g || !var || !g_wmu || !g_wrho || !ws || !bde_lrt_linear_supported(B, I, O) || ldx < I)
    return BDE_ERR_INVALID;
  if ((b_rho == nullptr) != (g_bmu == nullptr) || (g_bmu == nullptr) != (g_brho == nullptr)) return BDE_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nb = lrt_bwd_nb(B), b_pad = nb * 32;
  float* gvar = static_cast<float*>(ws);
  float* gT = gvar + pad64(static_cast<size_t>(B) * O);
  float* gvT = gT + pad64(static_cast<size_t>(O) * b_pad);
  float* part = gvT + pad64(static_cast<size_t>(O) * b_pad);
  const int pgrid = (O + kPrepCols - 1) / kPrepCols;
  const int i_tiles = (I + 31) / 32, o_tiles = (O + 31) / 32;
  // One pass over the weights for layers up to 2^20 weights (the reference's sizes: 23 vs 38 us at the iWildCam head).
  // Wide layers keep the two kernels: the fused kernel's register set (352 VGPRs at 2 batch tiles, spills at 4) leaves
  // one wave per SIMD and its per-tile chain un-overlapped (4096 x 4096 at batch 64: 303 vs 180 us).
  const bool fused = g_x && nb <= 2 && static_cast<int64_t>(I) * O <= (int64_t{1} << 20);
  const bool pre = !fused && w_s2 && w_ds2;    // cached sigma^2 / its rho-derivative of this weight version
  // input gradient with 16-byte loads: float4-addressable rows of the weights and of g / gvar
  const bool x4 = g_x && !fused && BDE_LRT_BWD_X4 && nb <= 2 && (I % 4 == 0) && (O % 4 == 0) && aligned16(w_mu) &&
                  aligned16(pre ? w_s2 : w_rho) && aligned16(g) && aligned16(gvar);
  float* gT_arg = (g_x && !x4) ? gT : nullptr;                       // the transposed copies: lrt_bwd_x_kernel's A operand
  if (eps)
    hipLaunchKernelGGL(lrt_bwd_prep_kernel<false>, dim3(pgrid), dim3(kPrepCols * kPrepRows), 0, s, g, var, eps, seed,
                       stream_id, b_rho, clamp_bias_var, gvar, gT_arg, gvT, g_bmu, g_brho, B, b_pad, O);
  else
    hipLaunchKernelGGL(lrt_bwd_prep_kernel<true>, dim3(pgrid), dim3(kPrepCols * kPrepRows), 0, s, g, var, eps, seed,
                       stream_id, b_rho, clamp_bias_var, gvar, gT_arg, gvT, g_bmu, g_brho, B, b_pad, O);
  int rc = to_err(hipGetLastError());
  if (rc) return rc;
  if (!fused) {                          // weight gradients: one wave per 32 x 32 tile
    const int64_t w_units = static_cast<int64_t>(i_tiles) * o_tiles;
    const dim3 wgrid(static_cast<unsigned>((w_units + kLrtBwdWaves - 1) / kLrtBwdWaves));
    if (pre)
      hipLaunchKernelGGL(lrt_bwd_w_kernel<true>, wgrid, dim3(kLrtBwdWaves * 64), 0, s, x, ldx, w_ds2, g, gvar, B, I, O, g_wmu,
                         g_wrho);
    else
      hipLaunchKernelGGL(lrt_bwd_w_kernel<false>, wgrid, dim3(kLrtBwdWaves * 64), 0, s, x, ldx, w_rho, g, gvar, B, I, O,
                         g_wmu, g_wrho);
    rc = to_err(hipGetLastError());
    if (rc || !g_x) return rc;
  }
  const LrtBwdPlan plan = x4 ? lrt_bwd_plan(I, O, BDE_LRT_BWD_X4_TARGET_WAVES) : lrt_bwd_plan(I, O);   // (never more slices: ws_bytes)
  const int x_units = i_tiles * plan.n_slices;
  const int xgrid = (x_units + kLrtBwdWaves - 1) / kLrtBwdWaves;
#define BDE_LRT_X(NB, DIRECT)                                                                                              \
  do {                                                                                                                     \
    if (pre)                                                                                                               \
      hipLaunchKernelGGL((lrt_bwd_x_kernel<NB, DIRECT, true>), dim3(xgrid), dim3(kLrtBwdWaves * 64), 0, s, x, ldx, w_mu,    \
                         w_s2, gT, gvT, B, I, O, plan.n_slices, plan.oslice, g_x, part);                                   \
    else                                                                                                                   \
      hipLaunchKernelGGL((lrt_bwd_x_kernel<NB, DIRECT, false>), dim3(xgrid), dim3(kLrtBwdWaves * 64), 0, s, x, ldx, w_mu,   \
                         w_rho, gT, gvT, B, I, O, plan.n_slices, plan.oslice, g_x, part);                                  \
  } while (0)
#define BDE_LRT_F(NB, DIRECT) \
  hipLaunchKernelGGL((lrt_bwd_fused_kernel<NB, DIRECT>), dim3(xgrid), dim3(kLrtBwdWaves * 64), 0, s, x, ldx, w_mu, w_rho, g, \
                     gvar, gT, gvT, B, I, O, plan.n_slices, plan.oslice, g_wmu, g_wrho, g_x, part)
  const bool direct = plan.n_slices == 1;
  const bool x4_full = (B % 32 == 0) && (I % 32 == 0) && (O % 32 == 0) && (plan.oslice % (2 * kX4FullRows) == 0) &&
                       static_cast<int64_t>(std::max(I, B)) * O < (int64_t{1} << 30);   // 32-bit per-lane offsets
#define BDE_LRT_X4_(NB, DIRECT, PRE, FULL)                                                                                 \
  hipLaunchKernelGGL((lrt_bwd_x4_kernel<NB, DIRECT, PRE, FULL>), dim3(xgrid), dim3(kLrtBwdWaves * 64), 0, s, x, ldx, w_mu,  \
                     (PRE) ? w_s2 : w_rho, g, gvar, B, I, O, plan.n_slices, plan.oslice, g_x, part)
#define BDE_LRT_X4(NB, DIRECT)                                                                                             \
  do {                                                                                                                     \
    if (pre) {                                                                                                             \
      if (x4_full) BDE_LRT_X4_(NB, DIRECT, true, true);                                                                    \
      else BDE_LRT_X4_(NB, DIRECT, true, false);                                                                           \
    } else {                                                                                                               \
      if (x4_full) BDE_LRT_X4_(NB, DIRECT, false, true);                                                                   \
      else BDE_LRT_X4_(NB, DIRECT, false, false);                                                                          \
    }                                                                                                                      \
  } while (0)
  if (x4) {
    if (nb == 1) { if (direct) BDE_LRT_X4(1, true); else BDE_LRT_X4(1, false); }
    else { if (direct) BDE_LRT_X4(2, true); else BDE_LRT_X4(2, false); }
  } else if (!fused) {
    if (nb == 1) { if (direct) BDE_LRT_X(1, true); else BDE_LRT_X(1, false); }
    else if (nb == 2) { if (direct) BDE_LRT_X(2, true); else BDE_LRT_X(2, false); }
    else { if (direct) BDE_LRT_X(4, true); else BDE_LRT_X(4, false); }
  } else if (nb == 1) {
    if (direct) BDE_LRT_F(1, true); else BDE_LRT_F(1, false);
  } else {
    if (direct) BDE_LRT_F(2, true); else BDE_LRT_F(2, false);
  }
#undef BDE_LRT_X
#undef BDE_LRT_X4
#undef BDE_LRT_X4_
#undef BDE_LRT_F
  rc = to_err(hipGetLastError());
  if (rc || plan.n_slices == 1) return rc;
  hipLaunchKernelGGL(lrt_bwd_x_finish_kernel, dim3(stream_grid(static_cast<int64_t>(B) * I)), dim3(kBlock), 0, s, part,
                     plan.n_slices, b_pad, x, ldx, g_x, B, I);
  return to_err(hipGetLastError());
}

// bde_init(): load this translation unit's code object on the current device now (HIP otherwise uploads it at the
// first launch of one of its kernels).  Internal to the library (not exported).
extern "C" __attribute__((visibility("hidden"))) int bde_internal_load_lrt_bwd(void) {
  hipFuncAttributes attr;
  return bde::to_err(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(&bde::lrt_bwd_prep_kernel<true>)));
}
