// Shared by the fused-convolution kernels (conv_lrt.hip, conv_lrt_bwd.hip): staging an input patch into LDS.
#pragma once
#include "bde_common.hpp"

namespace bde {

using f32x2 = __attribute__((ext_vector_type(2))) float;

// XCD-aware work assignment.  The hardware hands consecutive workgroups of a launch to the 8 XCDs round robin (workgroup L runs
// on XCD L % 8), and every XCD has its own L2.  The kernels here have groups of `inner` workgroups that read the SAME input
// (the channel tiles of one image band; the column groups of one share of items): this maps the launch's linear workgroup
// index to a work index such that the workgroups of one XCD get CONSECUTIVE work indices -- with `inner` fastest, the
// workgroups that share an input sit on one XCD, next to each other in time, and the second one finds the input in that L2.
// A bijection on [0, total) for any total (XCD x owns (total - x + 7) / 8 of them).
__device__ __forceinline__ int xcd_work_index(int linear, int total) {
  constexpr int kXcds = 8;
  const int q = total / kXcds, r = total % kXcds;
  const int x = linear % kXcds;
  return x * q + (x < r ? x : r) + linear / kXcds;
}

// Stage the patch of one channel chunk -- planes (image, channel) of PH x PWP elements each, LDS layout
// [img][c][PH][PWP] of PAIRS: .x <- the tensor (zero outside it), .y <- clamp(x^2, 1e-4) of the same element (MODE 0; zero in
// the padding: F.conv2d pads the clamped tensor) or the matching element of a second tensor (MODE 1: g and g_var of the
// input-gradient pass, optionally zero-dilated by (dh, dw)).  The two products of the layer read their B operands at the same
// element: interleaved, ONE ds_read_b64 (one address) serves both, and the staging pass writes one ds_write_b64 per element.
//
// Round 4 staged one plane per wave trip with the lanes along a patch ROW, four rows of loads in flight: a 10 x 10 patch
// of an 8 x 8 image kept 10 of 64 lanes busy and put 12 load instructions of 4 in flight behind each other per plane --
// at the 64-channel layers the staging latency was several times the matrix work of the chunk.  Here a plane's elements
// are FLAT over the lanes (element e = 64 step + lane; (row, column) = (e / PWP, e % PWP) by an exact float reciprocal,
// below), a wave walks its planes (wave, wave + 4, ...) as one sequence of (plane, step) items, and EIGHT items' loads
// (sixteen in MODE 1) are issued before the first LDS store: 2 instead of 12 load instructions per 10 x 10 plane, all of a
// batch in flight at once, consecutive lanes on consecutive addresses of a row (and of LDS: conflict-free stores).
//
// e / PWP: int((e + 0.5f) * rcp_pwp) with rcp_pwp = fl(1 / PWP).  Exact for e < 8192 (a plane never has more elements: two
// images of it must fit 64 KB): the product's relative error is below 2^-23, i.e. below 9.8e-4 / PWP absolute for
// e / PWP <= 8192 / PWP, while (e + 0.5) / PWP is at least 0.5 / PWP away from an integer (tests/test_conv_emulated.py
// checks every (e, PWP) exhaustively).
// DEPTH: items (loads) in flight per lane and batch: 8, or 16 where the caller has the registers.
// DIL: the source images are zero-dilated by (dh, dw) (the single-launch input gradient of a strided layer): the per-element
// modulo / division only exists in that instantiation.  The batch is straight-line code: the cursor over (plane, step) items
// advances with wave-uniform selects, an item past the last plane is a dropped load of element 0 -- no branch, and no
// register written on two paths, separates the eight (sixteen) loads.
template <int MODE, bool DIL, int DEPTH = 8>
__device__ __forceinline__ void conv_stage_patch(const float* __restrict__ src1, const float* __restrict__ src2,
                                                 f32x2* __restrict__ xq, int wave, int lane,
                                                 int planes, int cc, int row_elems, int PWP, float rcp_pwp, int img_floats,
                                                 int img0, int N, int C, int c0, int H, int W, int hi0, int pw, int dh,
                                                 int dw) {
  const int S = (row_elems + 63) >> 6;                     // steps of 64 lanes per plane
  const int64_t hw = static_cast<int64_t>(H) * W;
  const int dq = 4 / cc, dr = 4 % cc;                       // a wave's next plane is 4 further: (image, channel) += (dq, dr)
  int pl = wave, img = wave / cc, c = wave % cc, step = 0;  // the cursor: wave-uniform
  while (pl < planes) {
    float v[DEPTH], v2[DEPTH];
    int off[DEPTH];
    bool okv[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      // wave-uniform part: is there an item, and the plane it reads (plane 0 stands in when there is none: its element 0 is
      // loaded and dropped) -- a scalar base pointer, the lane adds a 32-bit element offset
      const bool plane_ok = pl < planes && img0 + img < N;
      const int64_t plane_at = plane_ok ? (static_cast<int64_t>(img0 + img) * C + c0 + c) * hw : int64_t{0};
      const float* __restrict__ p1 = src1 + plane_at;
      const float* __restrict__ p2 = src2 + plane_at;
      const int lds_base = pl < planes ? img * img_floats + c * row_elems : -1;
      // per lane
      const int e = step * 64 + lane;
      const int py = static_cast<int>((static_cast<float>(e) + 0.5f) * rcp_pwp);
      const int px = e - py * PWP;
      int hi = hi0 + py, wi = px - pw;
      bool ok = plane_ok && e < row_elems;
      if (DIL) {
        ok = ok && hi >= 0 && wi >= 0 && hi % dh == 0 && wi % dw == 0;
        hi /= dh;
        wi /= dw;
      }
      ok = ok && static_cast<unsigned>(hi) < static_cast<unsigned>(H) && static_cast<unsigned>(wi) < static_cast<unsigned>(W);
      off[u] = (lds_base >= 0 && e < row_elems) ? lds_base + e : -1;
      // the load itself is unconditional (element 0 stands in for whatever is not loaded and is dropped below): a load under a
      // divergent branch would get a basic block and a wait of its own instead of joining the batch in flight
      const int idx = ok ? hi * W + wi : 0;
      v[u] = p1[idx];
      v2[u] = MODE == 1 ? p2[idx] : 0.f;
      okv[u] = ok;
      const bool wrap = step + 1 == S;                       // this wave's next plane
      step = wrap ? 0 : step + 1;
      const int c_next = c + dr >= cc ? c + dr - cc : c + dr;
      const int img_next = img + dq + (c + dr >= cc ? 1 : 0);
      pl = wrap ? pl + 4 : pl;
      c = wrap ? c_next : c;
      img = wrap ? img_next : img;
    }
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      if (off[u] >= 0) {
        f32x2 pair;
        pair.x = okv[u] ? v[u] : 0.f;
        pair.y = !okv[u] ? 0.f : (MODE == 1 ? v2[u] : fmaxf(v[u] * v[u], 1e-4f));
        xq[off[u]] = pair;
      }
    }
  }
}

// The weight-gradient kernel's A operands: rows (output channel o, image img) of g and g_var -- bpi contiguous pixels of a
// band each -- as PAIRS (g, g_var) into gq [o][GP] at column img * bpi + p; zero for rows / pixels outside the tensor.  The same scheme as
// the patch: a wave's rows (wave, wave + 4, ...) x steps of 64 lanes as one sequence of items, eight items (sixteen loads) in
// flight, straight-line (round 4: one row per trip, two loads in flight -- the staging latency exceeded the matrix work).
template <int DEPTH = 8>
__device__ __forceinline__ void conv_stage_rows(const float* __restrict__ g, const float* __restrict__ gvar,
                                                f32x2* __restrict__ gq, int wave, int lane, int rows,
                                                int NI, int bpi, int valid_pixels, int GP, int img0, int N, int o0, int O,
                                                int64_t howo, int64_t band_at) {
  const int S = (bpi + 63) >> 6;
  const int dq = 4 / NI, dr = 4 % NI;                       // row r = o * NI + img; the wave's next row is 4 further
  int r = wave, o = wave / NI, img = wave % NI, step = 0;   // the cursor: wave-uniform
  while (r < rows) {
    float a[DEPTH], b[DEPTH];
    int off[DEPTH];
    bool okv[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const bool row_ok = r < rows && img0 + img < N && o0 + o < O;
      const int64_t row_at = row_ok ? (static_cast<int64_t>(img0 + img) * O + o0 + o) * howo + band_at : int64_t{0};
      const float* __restrict__ p1 = g + row_at;
      const float* __restrict__ p2 = gvar + row_at;
      const int lds_base = r < rows ? o * GP + img * bpi : -1;
      const int p = step * 64 + lane;
      const bool ok = row_ok && p < valid_pixels;
      off[u] = (lds_base >= 0 && p < bpi) ? lds_base + p : -1;
      const int idx = ok ? p : 0;                            // unconditional loads: element 0 stands in and is dropped
      a[u] = p1[idx];
      b[u] = p2[idx];
      okv[u] = ok;
      const bool wrap = step + 1 == S;
      step = wrap ? 0 : step + 1;
      const int img_next = img + dr >= NI ? img + dr - NI : img + dr;
      const int o_next = o + dq + (img + dr >= NI ? 1 : 0);
      r = wrap ? r + 4 : r;
      img = wrap ? img_next : img;
      o = wrap ? o_next : o;
    }
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      if (off[u] >= 0) {
        f32x2 pair;
        pair.x = okv[u] ? a[u] : 0.f;
        pair.y = okv[u] ? b[u] : 0.f;
        gq[off[u]] = pair;
      }
    }
  }
}

}  // namespace bde
