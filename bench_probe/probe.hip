// bench-only helper (NOT part of libbde_hip.so, not imported by the package): the "same-shape stream probes" bench.py
// uses to normalise roofline numbers across the pool's devices.
//
// A probe has the read/write SHAPE of one of the product's streaming kernels -- NR rows of D floats read, NW rows
// written, NRMW rows read-modify-written in place, rows a leading dimension apart or interleaved in pieces
// ([piece][row][2^lp floats], the SWAG statistics layout) -- the same grid-stride walk over float4 columns, the same
// load / store flavours (non-temporal loads on read-once streams; result rows stored either way, as the kernel stores
// them), 2048 workgroups of 256 threads, and trivial arithmetic (sums).  What it reaches is what the silicon gives a
// streaming kernel of that shape on THIS device in THIS allocation state; frac_of_probe = kernel rate / probe rate is
// the device-independent figure (round 3 had this for the SVGD combine kernel only, shape R16 W8).
#include <hip/hip_runtime.h>
#include <stdint.h>

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct Rows {
  const float* base;   // first row
  int64_t ld;          // floats between rows (contiguous rows) or between the rows of one piece (pieces)
  int log2_piece;      // 0: contiguous rows; else rows cut into pieces of 2^log2_piece floats
  int64_t piece_stride;
};
__device__ __forceinline__ int64_t elem_off(const Rows& r, int64_t e) {
  if (r.log2_piece == 0) return e;
  return (e >> r.log2_piece) * r.piece_stride + (e & ((int64_t{1} << r.log2_piece) - 1));
}

template <int NR, int NW, int NRMW, bool NT_STORE>
__global__ __launch_bounds__(256) void stream_probe_kernel(Rows rd, Rows wr, Rows rw, int64_t n4) {
  const int64_t step = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; c < n4; c += step) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int64_t ro = elem_off(rd, 4 * c), wo = elem_off(wr, 4 * c), mo = elem_off(rw, 4 * c);
#pragma unroll
    for (int r = 0; r < NR; ++r) acc += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rd.base + r * rd.ld + ro));
    f32x4 old[NRMW > 0 ? NRMW : 1];
#pragma unroll
    for (int r = 0; r < NRMW; ++r) old[r] = *reinterpret_cast<const f32x4*>(rw.base + r * rw.ld + mo);
#pragma unroll
    for (int r = 0; r < NRMW; ++r)
      *reinterpret_cast<f32x4*>(const_cast<float*>(rw.base) + r * rw.ld + mo) = old[r] + acc;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      f32x4* p = reinterpret_cast<f32x4*>(const_cast<float*>(wr.base) + w * wr.ld + wo);
      if (NT_STORE) __builtin_nontemporal_store(acc + static_cast<float>(w), p);
      else *p = acc + static_cast<float>(w);
    }
  }
}

template <int NR, int NW, int NRMW>
static int launch(bool nt_store, Rows rd, Rows wr, Rows rw, int64_t n4, hipStream_t s) {
  if (nt_store) hipLaunchKernelGGL((stream_probe_kernel<NR, NW, NRMW, true>), dim3(2048), dim3(256), 0, s, rd, wr, rw, n4);
  else hipLaunchKernelGGL((stream_probe_kernel<NR, NW, NRMW, false>), dim3(2048), dim3(256), 0, s, rd, wr, rw, n4);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// The shapes of the product's streaming kernels (reads, writes, read-modify-writes):
//   (16,8,0) svgd_combine            (2,1,0) gauss_draw_fwd / local_reparam_fwd     (2,0,2) gauss_kl accumulate / gauss_draw_bwd
//   (2,2,0) gauss_kl overwrite       (22,1,0) swag_sample K = 20                    (22,30,0) swag_sample_batched K = 20, S = 30
//   (1,1,2) swag_update              (2,1,1) ivon_sample                            (2,0,3) ivon_update          (1,1,0) copy
//   (8,0,0) svgd_gram (pure read)    (32,30,0) batched sampler at K = 30
// n = valid floats per row; rows of a group are `ld` floats apart; log2_piece > 0: rows interleaved in pieces.
extern "C" int bde_bench_probe(int n_read, int n_write, int n_rmw, int nt_store,
                               const float* rd, int64_t ld_rd, int lp_rd, int64_t ps_rd,
                               float* wr, int64_t ld_wr, int lp_wr, int64_t ps_wr,
                               float* rw, int64_t ld_rw, int64_t n, void* stream) {
  if (n < 4) return -1;
  const Rows R{rd, ld_rd, lp_rd, ps_rd}, W{wr, ld_wr, lp_wr, ps_wr}, M{rw, ld_rw, 0, 0};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t n4 = n >> 2;
  const bool nt = nt_store != 0;
#define SHAPE(a, b, c) if (n_read == a && n_write == b && n_rmw == c) return launch<a, b, c>(nt, R, W, M, n4, s)
  SHAPE(16, 8, 0); SHAPE(2, 1, 0); SHAPE(2, 0, 2); SHAPE(2, 2, 0); SHAPE(22, 1, 0); SHAPE(22, 30, 0); SHAPE(1, 1, 2);
  SHAPE(2, 1, 1); SHAPE(2, 0, 3); SHAPE(1, 1, 0); SHAPE(8, 0, 0); SHAPE(32, 30, 0);
#undef SHAPE
  return -3;
}

int probe_two_groups(const float* a, const float* b, float* out, int64_t ld, int64_t n4, hipStream_t s);

// round 3's entry point (svgd_combine's shape: a, b: [8, ld] each read -- two separate allocations --, out: [8, ld] written)
extern "C" int bde_bench_probe_r16w8(const float* a, const float* b, float* out, int64_t ld, int64_t n, void* stream) {
  if (!a || !b || !out || n < 4 || ld < n || (ld & 3)) return -1;
  return probe_two_groups(a, b, out, ld, n >> 2, static_cast<hipStream_t>(stream));
}

__global__ __launch_bounds__(256) void stream_probe_r8r8w8_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                 float* __restrict__ out, int64_t ld, int64_t n4) {
  const int64_t step = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; c < n4; c += step) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      acc += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a + r * ld + 4 * c));
      acc += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(b + r * ld + 4 * c));
    }
#pragma unroll
    for (int w = 0; w < 8; ++w)
      __builtin_nontemporal_store(acc + static_cast<float>(w), reinterpret_cast<f32x4*>(out + w * ld + 4 * c));
  }
}

int probe_two_groups(const float* a, const float* b, float* out, int64_t ld, int64_t n4, hipStream_t s) {
  hipLaunchKernelGGL(stream_probe_r8r8w8_kernel, dim3(2048), dim3(256), 0, s, a, b, out, ld, n4);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
