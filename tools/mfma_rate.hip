// Issue rate of v_mfma_f32_32x32x2_f32 on gfx950: N dependent-distance-4 products per wave, W waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o tools/bin/mfma_rate && tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

// VALU: independent vector instructions (v_fma_f32) issued after every MFMA -- do they run in its shadow?
template <int CHAINS, int VALU = 0>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int n) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x16{};
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  float va[VALU + 1];
  for (int v = 0; v <= VALU; ++v) va[v] = a + v;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < VALU; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(va[v]) : "v"(b));
    }
  }
  float s = 0.f;
  for (int v = 0; v < VALU; ++v) s += va[v];
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][15];
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CHAINS, int VALU = 0>
void run(int waves_per_simd, int n) {
  float* out; unsigned long long* cyc;
  const int blocks = 256 * waves_per_simd;   // 4 waves per block, one per SIMD
  hipMalloc(&out, sizeof(float) * blocks * 256); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS, VALU>), dim3(blocks), dim3(256), 0, 0, out, cyc, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double mfmas = double(n) * CHAINS;
  printf("VALU per MFMA %d  chains %d  waves/SIMD %d: %8.1f us, %7.1f cycles per MFMA of one wave (%.1f per SIMD slot), %.2f GHz by s_memtime, %.1f TFLOP/s\n",
         VALU, CHAINS, waves_per_simd, ms * 1e3, c / mfmas, c / mfmas / waves_per_simd, c / (ms * 1e-3) / 1e9,
         mfmas * 4096.0 * 1024 * waves_per_simd / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 4}) { run<1>(w, 4096); run<2>(w, 2048); run<4>(w, 1024); }
  for (int w : {1, 2}) { run<2, 1>(w, 2048); run<2, 2>(w, 2048); run<2, 4>(w, 2048); run<2, 8>(w, 2048); }
  return 0;
}
