// bench-only helper (NOT part of libbde_hip.so, not imported by the package): the "same-shape stream probe" bench.py
// uses to normalise roofline numbers across the pool's devices.  A kernel with the read/write SHAPE of
// svgd_combine_kernel<8, true> -- 16 rows of D floats read (8 particle rows + 8 gradient rows, 95 MB apart), 8 rows
// written, grid-stride walk over float4 columns, non-temporal loads and stores, 2048 workgroups of 256 threads --
// and trivial arithmetic (a sum).  What it reaches is what the silicon gives a streaming kernel of that shape
// (tools/kexp5.hip "probe R16 W8"; 5.1-5.7 TB/s depending on the device); roofline.frac_of_probe = kernel / probe.
#include <hip/hip_runtime.h>
#include <stdint.h>

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NR, int NW>
__global__ __launch_bounds__(256) void stream_probe_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ out, int64_t ld, int64_t n4) {
  const int64_t step = static_cast<int64_t>(gridDim.x) * 256;
  for (int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; c < n4; c += step) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NR / 2; ++r) {
      acc += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a + r * ld + 4 * c));
      acc += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(b + r * ld + 4 * c));
    }
#pragma unroll
    for (int w = 0; w < NW; ++w)
      __builtin_nontemporal_store(acc + static_cast<float>(w), reinterpret_cast<f32x4*>(out + w * ld + 4 * c));
  }
}

// a, b: [8, ld] each (read), out: [8, ld] (written); n = valid floats per row.  Enqueues one launch on `stream`.
extern "C" int bde_bench_probe_r16w8(const float* a, const float* b, float* out, int64_t ld, int64_t n, void* stream) {
  if (!a || !b || !out || n < 4 || ld < n || (ld & 3)) return -1;
  hipLaunchKernelGGL((stream_probe_kernel<16, 8>), dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), a, b, out, ld,
                     n >> 2);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
