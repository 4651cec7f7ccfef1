// SWAG sample kernel experiments (development tool)
#include "../beyond_deep_ensembles_amd/csrc/swag.hip"
#include <cstdio>
#include <vector>
#include <functional>
#include <string>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
using namespace bde;

template <int UNROLL, int BLOCK, bool NTM>
__global__ __launch_bounds__(BLOCK) void sample_v(const float* __restrict__ mean, const float* __restrict__ sq, const float* __restrict__ dev, int K,
                                                  int64_t ld, const float* __restrict__ wg, uint64_t seed, uint64_t stream_id, float* __restrict__ out, int64_t D) {
  __shared__ float w[BDE_MAX_RANK];
  for (int r = threadIdx.x; r < K; r += blockDim.x) w[r] = wg[r];
  __syncthreads();
  const int64_t n4 = D >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float* col = dev + 4 * i;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll(UNROLL)
    for (int r = 0; r < K; ++r) {
      const f32x4 d = ld4_nt(col + (int64_t)r * ld);
      const float wr = w[r];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(d[j], wr, acc[j]);
    }
    const f32x4 m = NTM ? ld4_nt(mean + 4 * i) : ld4(mean + 4 * i);
    const f32x4 s = NTM ? ld4_nt(sq + 4 * i) : ld4(sq + 4 * i);
    const f32x4 z = philox_normal4(seed, stream_id, (uint64_t)i, kDomainDiag);
    st4_nt(out + 4 * i, (m + acc) + diag_std(m, s) * z);
  }
}
// K fixed at compile time: all loads issued before the FMA chain
template <int KK, int BLOCK>
__global__ __launch_bounds__(BLOCK) void sample_k(const float* __restrict__ mean, const float* __restrict__ sq, const float* __restrict__ dev,
                                                  int64_t ld, const float* __restrict__ wg, uint64_t seed, uint64_t stream_id, float* __restrict__ out, int64_t D) {
  __shared__ float w[KK];
  for (int r = threadIdx.x; r < KK; r += blockDim.x) w[r] = wg[r];
  __syncthreads();
  const int64_t n4 = D >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float* col = dev + 4 * i;
    f32x4 d[KK];
#pragma unroll
    for (int r = 0; r < KK; ++r) d[r] = ld4_nt(col + (int64_t)r * ld);
    const f32x4 m = ld4_nt(mean + 4 * i);
    const f32x4 s = ld4_nt(sq + 4 * i);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < KK; ++r) {
      const float wr = w[r];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(d[r][j], wr, acc[j]);
    }
    const f32x4 z = philox_normal4(seed, stream_id, (uint64_t)i, kDomainDiag);
    st4_nt(out + 4 * i, (m + acc) + diag_std(m, s) * z);
  }
}
// pure read of K+2 rows (ceiling for this access pattern)
__global__ __launch_bounds__(256) void read_rows(const float* __restrict__ dev, int K, int64_t ld, float* __restrict__ sink, int64_t D) {
  const int64_t n4 = D >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  f32x4 acc = {0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
#pragma unroll 11
    for (int r = 0; r < K; ++r) acc += ld4_nt(dev + (int64_t)r * ld + 4 * i);
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
struct Variant { std::string name; std::function<void()> fn; double bytes; };
int main() {
  const int K = 20;
  const int64_t D = 23880950, ld = (D + 16 + 63) / 64 * 64;
  float *dev, *mean, *sq, *out, *wg;
  CK(hipMalloc(&dev, sizeof(float) * (K + 2) * ld)); CK(hipMalloc(&out, sizeof(float) * ld)); CK(hipMalloc(&wg, 1024));
  mean = dev + (int64_t)K * ld; sq = mean + ld;
  std::vector<float> h(ld);
  uint32_t s = 12345;
  for (int i = 0; i < K + 2; ++i) {
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f)) * 0.01f + 0.001f; }
    CK(hipMemcpy(dev + (int64_t)i * ld, h.data(), sizeof(float) * ld, hipMemcpyHostToDevice));
  }
  CK(hipMemcpy(wg, h.data(), 1024, hipMemcpyHostToDevice));
  float* mean2; CK(hipMalloc(&mean2, sizeof(float) * 2 * ld)); CK(hipMemcpy(mean2, mean, sizeof(float) * 2 * ld, hipMemcpyDeviceToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<Variant> vs;
  const double B = 4.0 * D * (K + 3);
  vs.push_back({"product swag_sample", [&] { bde_swag_sample(mean, sq, dev, K, ld, 3, nullptr, nullptr, 1, 2, out, D, st); }, B});
  vs.push_back({"read K+2 rows only g2048", [&] { hipLaunchKernelGGL(read_rows, dim3(2048), dim3(256), 0, st, dev, K + 2, ld, out, D); }, 4.0 * D * (K + 2)});
#define V(U, BL, G) vs.push_back({"unroll" #U " b" #BL " g" #G, [&] { hipLaunchKernelGGL((sample_v<U, BL, true>), dim3(G), dim3(BL), 0, st, mean, sq, dev, K, ld, wg, 1, 2, out, D); }, B});
  V(4, 256, 2048) V(5, 256, 2048) V(10, 256, 2048) V(20, 256, 2048) V(10, 256, 1024) V(10, 512, 1024) V(10, 256, 4096) V(20, 256, 1024)
#define W(BL, G) vs.push_back({"K20 all-loads-first b" #BL " g" #G, [&] { hipLaunchKernelGGL((sample_k<20, BL>), dim3(G), dim3(BL), 0, st, mean, sq, dev, ld, wg, 1, 2, out, D); }, B});
  W(256, 1024) W(256, 2048) W(256, 1280) W(512, 512) W(128, 4096)
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
  CK(hipGetLastError());
  printf("%-36s %9s %9s %9s\n", "variant", "min ms", "med ms", "TB/s(min)");
  for (size_t v = 0; v < vs.size(); ++v) {
    auto t = times[v]; std::sort(t.begin(), t.end());
    printf("%-36s %9.4f %9.4f %9.3f\n", vs[v].name.c_str(), t[0], t[t.size() / 2], vs[v].bytes / (t[0] * 1e-3) / 1e12);
  }
  return 0;
}
