// Kernel experiment harness (development tool, not part of the product):
// times kernel variants in ONE process with hipEvents (interleaved rounds).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I include tools/kexp.hip -o gpurun_out/kexp
#include "../beyond_deep_ensembles_amd/csrc/svgd.hip"
#include <cstdio>
#include <vector>
#include <functional>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

using namespace bde;

// ---- reference streams -----------------------------------------------------
__global__ __launch_bounds__(256) void read_kernel(const float* __restrict__ a, float* __restrict__ sink, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  f32x4 acc = {0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) acc += ld4(a + 4 * i);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
template <bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const float* __restrict__ a, float* __restrict__ b, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    if (NT) st4_nt(b + 4 * i, ld4_nt(a + 4 * i)); else st4(b + 4 * i, ld4(a + 4 * i));
  }
}
// 8-row read (the Gram's traffic) with combine-style coalescing: each lane reads its column of all rows
template <int M>
__global__ __launch_bounds__(256) void read_rows_kernel(const float* __restrict__ P, int64_t ld, float* __restrict__ sink, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  f32x4 acc = {0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
#pragma unroll
    for (int j = 0; j < M; ++j) acc += ld4(P + j * ld + 4 * i);
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

// ---- Gram variants -----------------------------------------------------------
// V1: register double-buffered loads (prefetch next tile before the MFMAs of this one)
template <int U>
__global__ __launch_bounds__(256) void gram_prefetch_kernel(const float* __restrict__ P, int M, int64_t D, int64_t ld, float* __restrict__ ws) {
  constexpr int W4 = 8, MP = 8;
  __shared__ float tile[4][16][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int prow = r16 & 7;
  const int c4 = (r16 >> 3) * 4 + kq;
  const bool valid = prow < M;
  const float inv_m = 1.0f / (float)M;
  const float* rowp = P + (int64_t)(valid ? prow : 0) * ld;
  const int64_t tile4 = (int64_t)U * W4;
  const int64_t n_tiles = (D / 4) / tile4;            // full tiles only (experiment)
  const int64_t waves_total = (int64_t)gridDim.x * 4;
  f32x4acc acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  int64_t t = (int64_t)blockIdx.x * 4 + wave;
  f32x4 cur[U], nxt[U];
  if (t < n_tiles) {
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = ld4(rowp + 4 * (t * tile4 + c4 + u * W4));
  }
  for (; t < n_tiles; t += waves_total) {
    const int64_t tn = t + waves_total;
    if (tn < n_tiles) {
#pragma unroll
      for (int u = 0; u < U; ++u) nxt[u] = ld4(rowp + 4 * (tn * tile4 + c4 + u * W4));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = valid ? cur[u][j] : 0.f;
        const float s = group_sum<2>(x);
        const float q = valid ? (x - s * inv_m) : 0.f;
        if ((j & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(q, q, acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q, q, acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = nxt[u];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[wave][4 * kq + r][r16] = acc0[r] + acc1[r];
  __syncthreads();
  if (threadIdx.x < MP * MP) {
    const int pi = threadIdx.x / MP, pj = threadIdx.x % MP;
    float s = 0.f;
    for (int w = 0; w < 4; ++w) s += tile[w][pi][pj] + tile[w][pi + 8][pj + 8];
    ws[kWsHeaderFloats + (int64_t)blockIdx.x * 64 + threadIdx.x] = s;
  }
}

// V1b: as V1, but only the first `thr4` float4 columns are loaded with allocating loads (to stay in the
// Infinity Cache for the combine pass); the rest is streamed non-temporally.
template <int U>
__global__ __launch_bounds__(256) void gram_prefetch_nt_kernel(const float* __restrict__ P, int M, int64_t D, int64_t ld, float* __restrict__ ws, int64_t thr4) {
  constexpr int W4 = 8, MP = 8;
  __shared__ float tile[4][16][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int prow = r16 & 7;
  const int c4 = (r16 >> 3) * 4 + kq;
  const bool valid = prow < M;
  const float inv_m = 1.0f / (float)M;
  const float* rowp = P + (int64_t)(valid ? prow : 0) * ld;
  const int64_t tile4 = (int64_t)U * W4;
  const int64_t n_tiles = (D / 4) / tile4;
  const int64_t waves_total = (int64_t)gridDim.x * 4;
  f32x4acc acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  int64_t t = (int64_t)blockIdx.x * 4 + wave;
  f32x4 cur[U], nxt[U];
  auto load = [&](int64_t tt, f32x4 (&v)[U]) {
    const bool keep = (tt + 1) * tile4 <= thr4;       // wave-uniform
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = keep ? ld4(rowp + 4 * (tt * tile4 + c4 + u * W4)) : ld4_nt(rowp + 4 * (tt * tile4 + c4 + u * W4));
  };
  if (t < n_tiles) load(t, cur);
  for (; t < n_tiles; t += waves_total) {
    const int64_t tn = t + waves_total;
    if (tn < n_tiles) load(tn, nxt);
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = valid ? cur[u][j] : 0.f;
        const float s = group_sum<2>(x);
        const float q = valid ? (x - s * inv_m) : 0.f;
        if ((j & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(q, q, acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q, q, acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = nxt[u];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[wave][4 * kq + r][r16] = acc0[r] + acc1[r];
  __syncthreads();
  if (threadIdx.x < MP * MP) {
    const int pi = threadIdx.x / MP, pj = threadIdx.x % MP;
    float s = 0.f;
    for (int w = 0; w < 4; ++w) s += tile[w][pi][pj] + tile[w][pi + 8][pj + 8];
    ws[kWsHeaderFloats + (int64_t)blockIdx.x * 64 + threadIdx.x] = s;
  }
}

// V2: VALU Gram with combine-style loads: each lane holds its column of all 8 rows; 36 pair products
__global__ __launch_bounds__(256) void gram_valu_kernel(const float* __restrict__ P, int64_t D, int64_t ld, float* __restrict__ ws) {
  constexpr int M = 8;
  __shared__ float red[4][36];
  const int64_t n4 = D >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float acc[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) acc[k] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 x[M];
#pragma unroll
    for (int j = 0; j < M; ++j) x[j] = ld4(P + j * ld + 4 * i);
    f32x4 mean = x[0];
#pragma unroll
    for (int j = 1; j < M; ++j) mean += x[j];
    mean *= (1.0f / M);
#pragma unroll
    for (int j = 0; j < M; ++j) x[j] -= mean;
    int k = 0;
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
      for (int b = a; b < M; ++b) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[k] = __builtin_fmaf(x[a][c], x[b][c], acc[k]);
        ++k;
      }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 36; ++k) {
    float v = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 36) ws[kWsHeaderFloats + (int64_t)blockIdx.x * 64 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ---- combine variants --------------------------------------------------------
template <int M, bool NT>
__global__ __launch_bounds__(256) void combine_v_kernel(const float* __restrict__ P, const float* G, float* out, int64_t D, int64_t ld,
                                                        const float* __restrict__ cgT, const float* __restrict__ cpT) {
  const int64_t n4 = D >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += stride) {
    f32x4 acc[M];
#pragma unroll
    for (int i = 0; i < M; ++i) acc[i] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const f32x4 p = NT ? ld4_nt(P + j * ld + 4 * i4) : ld4(P + j * ld + 4 * i4);
      const f32x4 g = NT ? __builtin_nontemporal_load((const f32x4*)(G + j * ld + 4 * i4)) : *(const f32x4*)(G + j * ld + 4 * i4);
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const float a = cgT[j * M + i], b = cpT[j * M + i];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = __builtin_fmaf(b, p[c], __builtin_fmaf(a, g[c], acc[i][c]));
      }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
      if (NT) __builtin_nontemporal_store(acc[i], (f32x4*)(out + i * ld + 4 * i4)); else *(f32x4*)(out + i * ld + 4 * i4) = acc[i];
    }
  }
}

template <int M, int MODE>   // MODE 0: nt loads+stores; 1: nt stores only; 2: nt loads only; 3: reversed traversal + nt
__global__ __launch_bounds__(256) void combine_m_kernel(const float* __restrict__ P, const float* G, float* out, int64_t D, int64_t ld,
                                                        const float* __restrict__ cgT, const float* __restrict__ cpT) {
  const int64_t n4 = D >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < n4; it += stride) {
    const int64_t i4 = (MODE == 3) ? (n4 - 1 - it) : it;
    f32x4 acc[M];
#pragma unroll
    for (int i = 0; i < M; ++i) acc[i] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const bool ntl = (MODE == 0 || MODE == 2 || MODE == 3);
      const f32x4 p = (MODE == 3) ? ld4(P + j * ld + 4 * i4) : (ntl ? ld4_nt(P + j * ld + 4 * i4) : ld4(P + j * ld + 4 * i4));
      const f32x4 g = ntl ? __builtin_nontemporal_load((const f32x4*)(G + j * ld + 4 * i4)) : *(const f32x4*)(G + j * ld + 4 * i4);
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const float a = cgT[j * M + i], b = cpT[j * M + i];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = __builtin_fmaf(b, p[c], __builtin_fmaf(a, g[c], acc[i][c]));
      }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
      if (MODE != 2) __builtin_nontemporal_store(acc[i], (f32x4*)(out + i * ld + 4 * i4)); else *(f32x4*)(out + i * ld + 4 * i4) = acc[i];
    }
  }
}

#define ST_ASM(NAME, MODS) __device__ __forceinline__ void NAME(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off " MODS :: "v"(p), "v"(v) : "memory"); }
ST_ASM(st4_sc1, "sc1")
ST_ASM(st4_sc0sc1, "sc0 sc1")
ST_ASM(st4_sc0sc1nt, "sc0 sc1 nt")
ST_ASM(st4_ntonly, "nt")
template <int M, int SMODE>
__global__ __launch_bounds__(256) void combine_s_kernel(const float* __restrict__ P, const float* G, float* out, int64_t D, int64_t ld,
                                                        const float* __restrict__ cgT, const float* __restrict__ cpT) {
  const int64_t n4 = D >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += stride) {
    f32x4 acc[M];
#pragma unroll
    for (int i = 0; i < M; ++i) acc[i] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < M; ++j) {
      const f32x4 p = ld4_nt(P + j * ld + 4 * i4);
      const f32x4 g = __builtin_nontemporal_load((const f32x4*)(G + j * ld + 4 * i4));
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const float a = cgT[j * M + i], b = cpT[j * M + i];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = __builtin_fmaf(b, p[c], __builtin_fmaf(a, g[c], acc[i][c]));
      }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
      float* q = out + i * ld + 4 * i4;
      if (SMODE == 0) st4_sc1(q, acc[i]); else if (SMODE == 1) st4_sc0sc1(q, acc[i]); else if (SMODE == 2) st4_sc0sc1nt(q, acc[i]); else st4_ntonly(q, acc[i]);
    }
  }
}

struct Variant { std::string name; std::function<void()> fn; double bytes; };

int main(int argc, char** argv) {
  const int M = 8;
  const int64_t D = 23880950, ld = (D + 16 + 63) / 64 * 64;
  float *P, *G, *O, *ws, *ks;
  CK(hipMalloc(&P, sizeof(float) * M * ld)); CK(hipMalloc(&G, sizeof(float) * M * ld)); CK(hipMalloc(&O, sizeof(float) * M * ld));
  CK(hipMalloc(&ws, bde_svgd_ws_bytes(M) + 4096)); CK(hipMalloc(&ks, sizeof(float) * 1024));
  std::vector<float> h(M * ld);
  uint32_t s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.1f; }
  CK(hipMemcpy(P, h.data(), sizeof(float) * M * ld, hipMemcpyHostToDevice));
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.02f; }
  CK(hipMemcpy(G, h.data(), sizeof(float) * M * ld, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  bde_svgd_gram(P, M, D, ld, ws, st);
  bde_svgd_kstats(ws, M, 0.f, 1.f, 129809.f, -1.f, 0.f, 0, ks, st);
  CK(hipStreamSynchronize(st));
  const float* cg = ks + 2 * 64 + 8 + 4; const float* cp = cg + 64;
  const int64_t n4 = D / 4, tot4 = (int64_t)M * ld / 4;
  std::vector<Variant> vs;
  const double B = 4.0 * M * D;
  vs.push_back({"read 764MB flat g2048", [&] { hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, st, P, ws, tot4); }, 4.0 * M * ld});
  vs.push_back({"read 764MB flat g4096", [&] { hipLaunchKernelGGL(read_kernel, dim3(4096), dim3(256), 0, st, P, ws, tot4); }, 4.0 * M * ld});
  vs.push_back({"read 8 rows coalesced g2048", [&] { hipLaunchKernelGGL(read_rows_kernel<8>, dim3(2048), dim3(256), 0, st, P, ld, ws, n4); }, B});
  vs.push_back({"copy 764MB g2048", [&] { hipLaunchKernelGGL(copy_kernel<false>, dim3(2048), dim3(256), 0, st, P, O, tot4); }, 8.0 * M * ld});
  vs.push_back({"copy 764MB nt g2048", [&] { hipLaunchKernelGGL(copy_kernel<true>, dim3(2048), dim3(256), 0, st, P, O, tot4); }, 8.0 * M * ld});
  vs.push_back({"copy 764MB g8192", [&] { hipLaunchKernelGGL(copy_kernel<false>, dim3(8192), dim3(256), 0, st, P, O, tot4); }, 8.0 * M * ld});
  vs.push_back({"gram product (mfma, g2048)", [&] { bde_svgd_gram(P, M, D, ld, ws, st); }, B});
  for (int g : {1024, 2048, 4096})
    vs.push_back({"gram mfma prefetch U4 g" + std::to_string(g), [&, g] { hipLaunchKernelGGL(gram_prefetch_kernel<4>, dim3(g), dim3(256), 0, st, P, M, D, ld, ws); }, B});
  vs.push_back({"gram mfma prefetch U8 g2048", [&] { hipLaunchKernelGGL(gram_prefetch_kernel<8>, dim3(2048), dim3(256), 0, st, P, M, D, ld, ws); }, B});
  vs.push_back({"gram mfma prefetch U2 g2048", [&] { hipLaunchKernelGGL(gram_prefetch_kernel<2>, dim3(2048), dim3(256), 0, st, P, M, D, ld, ws); }, B});
  for (int g : {1024, 2048})
    vs.push_back({"gram valu g" + std::to_string(g), [&, g] { hipLaunchKernelGGL(gram_valu_kernel, dim3(g), dim3(256), 0, st, P, D, ld, ws); }, B});
  vs.push_back({"combine product", [&] { bde_svgd_combine(P, G, O, M, D, ld, ld, ks, st); }, 3 * B});
  for (int g : {1024, 1280, 1536, 2048, 2560, 4096})
    vs.push_back({"combine plain g" + std::to_string(g), [&, g] { hipLaunchKernelGGL((combine_v_kernel<8, false>), dim3(g), dim3(256), 0, st, P, G, O, D, ld, cg, cp); }, 3 * B});
  for (int g : {1280, 2048})
    vs.push_back({"combine nt g" + std::to_string(g), [&, g] { hipLaunchKernelGGL((combine_v_kernel<8, true>), dim3(g), dim3(256), 0, st, P, G, O, D, ld, cg, cp); }, 3 * B});

  vs.push_back({"combine nt-stores-only g2048", [&] { hipLaunchKernelGGL((combine_m_kernel<8, 1>), dim3(2048), dim3(256), 0, st, P, G, O, D, ld, cg, cp); }, 3 * B});
  vs.push_back({"combine nt-loads-only g2048", [&] { hipLaunchKernelGGL((combine_m_kernel<8, 2>), dim3(2048), dim3(256), 0, st, P, G, O, D, ld, cg, cp); }, 3 * B});
  vs.push_back({"combine inplace nt g2048", [&] { hipLaunchKernelGGL((combine_m_kernel<8, 0>), dim3(2048), dim3(256), 0, st, P, O, O, D, ld, cg, cp); }, 3 * B});
  vs.push_back({"PAIR gram(pref g1024)+combine nt fwd", [&] { hipLaunchKernelGGL(gram_prefetch_kernel<4>, dim3(1024), dim3(256), 0, st, P, M, D, ld, ws);
      hipLaunchKernelGGL((combine_m_kernel<8, 0>), dim3(2048), dim3(256), 0, st, P, G, O, D, ld, cg, cp); }, 4 * B});
  vs.push_back({"PAIR gram(pref g1024)+combine nt REV", [&] { hipLaunchKernelGGL(gram_prefetch_kernel<4>, dim3(1024), dim3(256), 0, st, P, M, D, ld, ws);
      hipLaunchKernelGGL((combine_m_kernel<8, 3>), dim3(2048), dim3(256), 0, st, P, G, O, D, ld, cg, cp); }, 4 * B});
  for (int64_t thrM : {0, 4, 6, 8, 10, 24}) {
    const int64_t thr4 = thrM * 1000000 / 4;
    vs.push_back({"PAIR gram(keep first " + std::to_string(thrM) + "M cols)+combine nt", [&, thr4] {
        hipLaunchKernelGGL(gram_prefetch_nt_kernel<4>, dim3(1024), dim3(256), 0, st, P, M, D, ld, ws, thr4);
        hipLaunchKernelGGL((combine_m_kernel<8, 0>), dim3(2048), dim3(256), 0, st, P, G, O, D, ld, cg, cp); }, 4 * B});
  }
  vs.push_back({"PAIR gram + combine st sc1", [&] { bde_svgd_gram(P, M, D, ld, ws, st); hipLaunchKernelGGL((combine_s_kernel<8, 0>), dim3(2048), dim3(256), 0, st, P, G, O, D, ld, cg, cp); }, 4 * B});
  vs.push_back({"PAIR gram + combine st sc0 sc1", [&] { bde_svgd_gram(P, M, D, ld, ws, st); hipLaunchKernelGGL((combine_s_kernel<8, 1>), dim3(2048), dim3(256), 0, st, P, G, O, D, ld, cg, cp); }, 4 * B});
  vs.push_back({"PAIR gram + combine st sc0 sc1 nt", [&] { bde_svgd_gram(P, M, D, ld, ws, st); hipLaunchKernelGGL((combine_s_kernel<8, 2>), dim3(2048), dim3(256), 0, st, P, G, O, D, ld, cg, cp); }, 4 * B});
  vs.push_back({"PAIR gram + combine st nt(asm)", [&] { bde_svgd_gram(P, M, D, ld, ws, st); hipLaunchKernelGGL((combine_s_kernel<8, 3>), dim3(2048), dim3(256), 0, st, P, G, O, D, ld, cg, cp); }, 4 * B});
  vs.push_back({"PAIR product gram + product combine", [&] { bde_svgd_gram(P, M, D, ld, ws, st); bde_svgd_combine(P, G, O, M, D, ld, ld, ks, st); }, 4 * B});
  vs.push_back({"PAIR product gram + product combine INPLACE", [&] { bde_svgd_gram(P, M, D, ld, ws, st); bde_svgd_combine(P, O, O, M, D, ld, ld, ks, st); }, 4 * B});
  vs.push_back({"PAIR product gram+kstats+combine", [&] { bde_svgd_step(P, G, O, M, D, ld, 0.f, 1.f, 129809.f, -1.f, ws, ks, st); }, 4 * B});
  const int rounds = 7, inner = 5;
  std::vector<std::vector<float>> times(vs.size());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds; ++r) {
    for (size_t v = 0; v < vs.size(); ++v) {
      vs[v].fn();                                  // warm
      CK(hipEventRecord(e0, st));
      for (int k = 0; k < inner; ++k) vs[v].fn();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      times[v].push_back(ms / inner);
    }
  }
  CK(hipGetLastError());
  printf("%-36s %9s %9s %9s\n", "variant", "min ms", "med ms", "TB/s(min)");
  for (size_t v = 0; v < vs.size(); ++v) {
    auto t = times[v]; std::sort(t.begin(), t.end());
    printf("%-36s %9.4f %9.4f %9.3f\n", vs[v].name.c_str(), t[0], t[t.size() / 2], vs[v].bytes / (t[0] * 1e-3) / 1e12);
  }
  return 0;
}
