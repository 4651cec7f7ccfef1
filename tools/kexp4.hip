// RNG cost experiment (development tool): normals/s of counter-based generators on gfx950, VALU only
#include "../beyond_deep_ensembles_amd/csrc/bde_common.hpp"
#include <cstdio>
#include <vector>
#include <functional>
#include <string>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
using namespace bde;

template <int ROUNDS>
__device__ __forceinline__ uint4 philox_r(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
    c = make_uint4((uint32_t)(p1 >> 32) ^ c.y ^ k.x, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k.y, (uint32_t)p0);
    k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ uint32_t rotl(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }
template <int ROUNDS>
__device__ __forceinline__ uint4 threefry_r(uint4 in, uint4 key) {
  const int R[8][2] = {{10, 26}, {11, 21}, {13, 27}, {23, 5}, {6, 20}, {17, 11}, {25, 10}, {18, 20}};
  uint32_t ks[5] = {key.x, key.y, key.z, key.w, 0x1BD11BDAu ^ key.x ^ key.y ^ key.z ^ key.w};
  uint32_t X0 = in.x + ks[0], X1 = in.y + ks[1], X2 = in.z + ks[2], X3 = in.w + ks[3];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    if ((r & 1) == 0) { X0 += X1; X1 = rotl(X1, R[r & 7][0]) ^ X0; X2 += X3; X3 = rotl(X3, R[r & 7][1]) ^ X2; }
    else              { X0 += X3; X3 = rotl(X3, R[r & 7][0]) ^ X0; X2 += X1; X1 = rotl(X1, R[r & 7][1]) ^ X2; }
    if (((r + 1) & 3) == 0) {
      const int s = (r + 1) >> 2;
      X0 += ks[s % 5]; X1 += ks[(s + 1) % 5]; X2 += ks[(s + 2) % 5]; X3 += ks[(s + 3) % 5] + s;
    }
  }
  return make_uint4(X0, X1, X2, X3);
}
__device__ __forceinline__ f32x4 box_muller(uint4 r) {
  const float u0 = ((float)(r.x >> 8) + 1.0f) * (1.0f / 16777216.0f), u1 = (float)(r.y >> 8) * (1.0f / 16777216.0f);
  const float u2 = ((float)(r.z >> 8) + 1.0f) * (1.0f / 16777216.0f), u3 = (float)(r.w >> 8) * (1.0f / 16777216.0f);
  const float r0 = __builtin_sqrtf(-2.0f * __logf(u0)), r1 = __builtin_sqrtf(-2.0f * __logf(u2));
  return f32x4{r0 * __builtin_amdgcn_cosf(u1), r0 * __builtin_amdgcn_sinf(u1), r1 * __builtin_amdgcn_cosf(u3), r1 * __builtin_amdgcn_sinf(u3)};
}
template <int GEN, int ROUNDS, bool BM>
__global__ __launch_bounds__(256) void gen_kernel(float* __restrict__ out, int64_t n4, int reps) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < reps; ++s) {
      uint4 c = make_uint4((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)s, 0);
      uint4 r = GEN == 0 ? philox_r<ROUNDS>(c, make_uint2(1234u, 77u)) : threefry_r<ROUNDS>(c, make_uint4(1234u, 77u, 0u, 0u));
      if (BM) acc += box_muller(r); else acc += f32x4{(float)r.x, (float)r.y, (float)r.z, (float)r.w};
    }
    st4(out + 4 * i, acc);
  }
}
struct Variant { std::string name; std::function<void()> fn; };
int main() {
  const int64_t n4 = 23880950 / 4; const int reps = 30;
  float* out; CK(hipMalloc(&out, n4 * 16));
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<Variant> vs;
#define V(NAME, G, R, B) vs.push_back({NAME, [&] { hipLaunchKernelGGL((gen_kernel<G, R, B>), dim3(2048), dim3(256), 0, st, out, n4, reps); }});
  V("philox4x32-10 + BM", 0, 10, true) V("philox4x32-7 + BM", 0, 7, true) V("threefry4x32-20 + BM", 1, 20, true) V("threefry4x32-12 + BM", 1, 12, true)
  V("philox4x32-10 bits only", 0, 10, false) V("threefry4x32-20 bits only", 1, 20, false) V("box-muller only (0 rounds)", 0, 0, true)
  std::vector<std::vector<float>> times(vs.size());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < 5; ++r)
    for (size_t v = 0; v < vs.size(); ++v) {
      vs[v].fn();
      CK(hipEventRecord(e0, st)); vs[v].fn(); vs[v].fn(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); times[v].push_back(ms / 2);
    }
  for (size_t v = 0; v < vs.size(); ++v) {
    auto t = times[v]; std::sort(t.begin(), t.end());
    printf("%-30s %8.3f ms for %d x %lld normals  (%.1f Gnormals/s)\n", vs[v].name.c_str(), t[0], reps, (long long)(n4 * 4), reps * n4 * 4 / (t[0] * 1e-3) / 1e9);
  }
  return 0;
}
