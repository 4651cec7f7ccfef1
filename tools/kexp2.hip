// iVON update / streaming-pattern experiments (development tool)
#include "../beyond_deep_ensembles_amd/csrc/ivon.hip"
#include <cstdio>
#include <vector>
#include <functional>
#include <string>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
using namespace bde;

__device__ __forceinline__ void ivon_elem_fast(float& mean, float& mom, float& prec, float dsum, float acc, const IvonScalars& k,
                                               float imc, float ibc1, float ibc2) {
  const float gradient = acc * imc;
  const float g_mu = k.lam * mean + gradient;
  mom = k.beta1 * mom + k.omb1 * g_mu;
  const float g_s = ((k.lam - prec) + (((k.n_eff * prec) * dsum) * imc) * gradient) + k.damping;
  const float cm = mom * ibc1;
  const float cp = prec * ibc2;
  const float rp = __builtin_amdgcn_rcpf(prec);
  mean = mean - (k.lr * cm) * __builtin_amdgcn_rcpf(cp);
  prec = prec + (k.omb2 + ((k.c2 * g_s) * rp)) * g_s;
}
template <int MODE>   // 0 exact elem in-place, 1 fast elem in-place, 2 exact, nt loads of read-only streams, 3 no math (pure traffic)
__global__ __launch_bounds__(kBlock) void ivon_update_v(float* __restrict__ mean, float* __restrict__ momentum, float* __restrict__ prec,
                                                        const float* __restrict__ delta_sum, const float* __restrict__ acc_grad, IvonScalars k, int64_t n) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float imc = 1.0f / k.mc, ibc1 = 1.0f / k.bc1, ibc2 = 1.0f / k.bc2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 m = ld4(mean + 4 * i), mo = ld4(momentum + 4 * i), pr = ld4(prec + 4 * i);
    const f32x4 ds = (MODE == 2) ? ld4_nt(delta_sum + 4 * i) : ld4(delta_sum + 4 * i);
    const f32x4 ag = (MODE == 2) ? ld4_nt(acc_grad + 4 * i) : ld4(acc_grad + 4 * i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = m[j], b = mo[j], c = pr[j];
      if (MODE == 1) ivon_elem_fast(a, b, c, ds[j], ag[j], k, imc, ibc1, ibc2);
      else if (MODE == 3) { a += ds[j]; b += ag[j]; c += a; }
      else ivon_elem(a, b, c, ds[j], ag[j], k);
      m[j] = a; mo[j] = b; pr[j] = c;
    }
    st4(mean + 4 * i, m); st4(momentum + 4 * i, mo); st4(prec + 4 * i, pr);
  }
}
struct Variant { std::string name; std::function<void()> fn; double bytes; };
int main() {
  const int64_t D = 23880950, ld = (D + 16 + 63) / 64 * 64;
  float* b[5];
  std::vector<float> h(ld);
  uint32_t s = 12345;
  for (int i = 0; i < 5; ++i) {
    CK(hipMalloc(&b[i], sizeof(float) * ld));
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f)) * 0.01f + 0.001f; }
    CK(hipMemcpy(b[i], h.data(), sizeof(float) * ld, hipMemcpyHostToDevice));
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  const IvonScalars k{7.7e-4f, 129809.f, 2.f, 0.9f, 0.1f, 0.001f, 5e-7f, 0.271f, 0.003f, 1e-12f, 1e-3f};
  std::vector<Variant> vs;
  const double B = 32.0 * D;
  for (int g : {1024, 2048, 4096}) {
    vs.push_back({"ivon exact g" + std::to_string(g), [&, g] { hipLaunchKernelGGL(ivon_update_v<0>, dim3(g), dim3(256), 0, st, b[0], b[1], b[2], b[3], b[4], k, D); }, B});
  }
  vs.push_back({"ivon fast(rcp) g2048", [&] { hipLaunchKernelGGL(ivon_update_v<1>, dim3(2048), dim3(256), 0, st, b[0], b[1], b[2], b[3], b[4], k, D); }, B});
  vs.push_back({"ivon exact nt-ro g2048", [&] { hipLaunchKernelGGL(ivon_update_v<2>, dim3(2048), dim3(256), 0, st, b[0], b[1], b[2], b[3], b[4], k, D); }, B});
  vs.push_back({"ivon traffic-only g2048", [&] { hipLaunchKernelGGL(ivon_update_v<3>, dim3(2048), dim3(256), 0, st, b[0], b[1], b[2], b[3], b[4], k, D); }, B});
  const int rounds = 7, inner = 5;
  std::vector<std::vector<float>> times(vs.size());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds; ++r)
    for (size_t v = 0; v < vs.size(); ++v) {
      vs[v].fn();
      CK(hipEventRecord(e0, st));
      for (int q = 0; q < inner; ++q) vs[v].fn();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      times[v].push_back(ms / inner);
    }
  printf("%-36s %9s %9s %9s\n", "variant", "min ms", "med ms", "TB/s(min)");
  for (size_t v = 0; v < vs.size(); ++v) {
    auto t = times[v]; std::sort(t.begin(), t.end());
    printf("%-36s %9.4f %9.4f %9.3f\n", vs[v].name.c_str(), t[0], t[t.size() / 2], vs[v].bytes / (t[0] * 1e-3) / 1e12);
  }
  return 0;
}
