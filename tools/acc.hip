// accuracy probe for fast softplus / sigmoid / log formulations (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ __forceinline__ float sp_ref(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
// variant A: accurate expf + accurate logf + correction
__device__ __forceinline__ void sp_a(float x, float& sp, float& sg) {
  const float e = expf(-fabsf(x));
  const float u = 1.0f + e;
  const float l = logf(u) - ((u - 1.0f) - e) / u;
  sp = fmaxf(x, 0.f) + l;
  const float r = __builtin_amdgcn_rcpf(u);
  sg = x >= 0.f ? r : e * r;
}
// variant B: native exp/log
__device__ __forceinline__ void sp_b(float x, float& sp, float& sg) {
  const float e = __expf(-fabsf(x));
  const float u = 1.0f + e;
  const float r = __builtin_amdgcn_rcpf(u);
  const float l = __logf(u) - ((u - 1.0f) - e) * r;
  sp = fmaxf(x, 0.f) + l;
  sg = x >= 0.f ? r : e * r;
}
// variant C: accurate expf, native log with correction
__device__ __forceinline__ void sp_c(float x, float& sp, float& sg) {
  const float e = expf(-fabsf(x));
  const float u = 1.0f + e;
  const float r = __builtin_amdgcn_rcpf(u);
  const float l = __logf(u) - ((u - 1.0f) - e) * r;
  sp = fmaxf(x, 0.f) + l;
  sg = x >= 0.f ? r : e * r;
}
__global__ void k(const float* x, float* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a, b;
  out[i] = sp_ref(x[i]);
  sp_a(x[i], a, b); out[n + i] = a; out[2 * n + i] = b;
  sp_b(x[i], a, b); out[3 * n + i] = a; out[4 * n + i] = b;
  sp_c(x[i], a, b); out[5 * n + i] = a; out[6 * n + i] = b;
  out[7 * n + i] = __logf(sp_ref(x[i]));      // native log of softplus
  out[8 * n + i] = logf(sp_ref(x[i]));
  out[9 * n + i] = 1.0f / (1.0f + expf(-x[i]));
}
int main() {
  const int n = 1 << 20;
  std::vector<float> x(n), o(10 * n);
  for (int i = 0; i < n; ++i) x[i] = -30.f + 55.f * i / n;
  float *dx, *dout; hipMalloc(&dx, n * 4); hipMalloc(&dout, 10 * n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 10 * n * 4, hipMemcpyDeviceToHost);
  const char* names[] = {"sp ref(log1pf(expf))", "sp A", "sg A", "sp B(native)", "sg B", "sp C", "sg C", "native log(sp)", "logf(sp)", "sg ref"};
  for (int v = 0; v < 10; ++v) {
    double mx = 0; float at = 0;
    for (int i = 0; i < n; ++i) {
      double xd = x[i];
      double sp = xd > 0 ? xd + log1p(exp(-xd)) : log1p(exp(xd));
      double sg = 1.0 / (1.0 + exp(-xd));
      double want = (v == 0 || v == 1 || v == 3 || v == 5) ? sp : (v == 7 || v == 8) ? log((double)(float)sp) : sg;
      if (v == 7 || v == 8) { double e = fabs(o[(size_t)v * n + i] - log((double)o[i])); double rel = e / fmax(fabs(log((double)o[i])), 1e-3); if (rel > mx) { mx = rel; at = x[i]; } continue; }
      double rel = fabs(o[(size_t)v * n + i] - want) / fabs(want);
      if (rel > mx) { mx = rel; at = x[i]; }
    }
    printf("%-24s max rel err %.3e at x=%.3f\n", names[v], mx, at);
  }
  return 0;
}
