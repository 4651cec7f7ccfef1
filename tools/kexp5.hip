// Round-2 experiments (development tool; prints tables, results are copied into profiles/):
//   1. same-shape HBM probes: what a kernel that reads NR rows and writes NW rows of D floats can reach on
//      this device (the ceiling every streaming kernel of the library is compared with);
//   2. partitioning / loads-in-flight variants of the sampling kernels (swag_sample, gauss_draw_fwd);
//   3. Gram-pass variants; 4. small-model SVGD step: single launch vs three stages; 5. product kernels.
// Build: hipcc -O3 --offload-arch=gfx950 tools/kexp5.hip -Lbeyond_deep_ensembles_amd/lib -lbde_hip ...
#include "../beyond_deep_ensembles_amd/csrc/svgd_gram.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
#include <functional>
#include <string>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
using namespace bde;

// ------------------------------------------------------------------ probes --
// reads NR rows (stride ld), writes NW rows: out_w = sum of the reads (+ w); U float4 columns in flight per thread.
// CONTIG: every workgroup owns one contiguous chunk of columns; otherwise grid-stride.
template <int NR, int NW, int U, bool CONTIG, bool NT>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ in, float* __restrict__ out, int64_t ld, int64_t n4) {
  int64_t lo, hi, step;
  if (CONTIG) {
    const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
    lo = blockIdx.x * per + threadIdx.x;
    hi = std::min<int64_t>(n4, (blockIdx.x + 1) * per);
    step = 256;
  } else {
    lo = (int64_t)blockIdx.x * 256 + threadIdx.x;
    hi = n4;
    step = (int64_t)gridDim.x * 256;
  }
  f32x4 sink = {0, 0, 0, 0};
  for (int64_t i = lo; i < hi; i += step * U) {
    f32x4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t c = i + u * step;
        if (c < hi) acc[u] += NT ? ld4_nt(in + r * ld + 4 * c) : ld4(in + r * ld + 4 * c);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t c = i + u * step;
      if (c < hi) {
        if (NW == 0) sink += acc[u];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          if (NT) st4_nt(out + w * ld + 4 * c, acc[u] + (float)w);
          else st4(out + w * ld + 4 * c, acc[u] + (float)w);
        }
      }
    }
  }
  if (NW == 0 && sink[0] + sink[1] + sink[2] + sink[3] == 12345.678f) out[0] = sink[0];
}

// The batched sampler's ACCESS GRANULARITY without its arithmetic: a wave owns a tile of 32 float4 columns (512 B per
// row); lanes 0-31 and 32-63 read two DIFFERENT rows per instruction (the MFMA B-operand layout) and store two different
// output rows per instruction.  WIDE: a wave owns 64 float4 columns and every instruction touches 1 KB of ONE row.
template <int NR, int NW, bool WIDE>
__global__ __launch_bounds__(256) void probe_half(const float* __restrict__ in, float* __restrict__ out, int64_t ld, int64_t n4) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t cols = WIDE ? 64 : 32;
  const int64_t n_tiles = (n4 + cols - 1) / cols;
  const int64_t waves_total = (int64_t)gridDim.x * 4;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n_tiles; t += waves_total) {
    const int64_t c = t * cols + (WIDE ? lane : (lane & 31));
    if (c >= n4) continue;
    f32x4 acc = {0, 0, 0, 0};
    if (WIDE) {
#pragma unroll
      for (int r = 0; r < NR; ++r) acc += ld4_nt(in + r * ld + 4 * c);
#pragma unroll
      for (int w = 0; w < NW; ++w) st4_nt(out + w * ld + 4 * c, acc + (float)w);
    } else {
      const int half = lane >> 5;
#pragma unroll
      for (int r = 0; r < NR / 2; ++r) acc += ld4_nt(in + (2 * r + half) * ld + 4 * c);
#pragma unroll
      for (int w = 0; w < NW / 2; ++w) st4_nt(out + (2 * w + half) * ld + 4 * c, acc + (float)w);
    }
  }
}

// Layout probe: the same NR-in / NW-out shape, but the rows INTERLEAVED at `CH` floats: memory = [chunk][row][CH]
// (what a per-tensor blocked particle layout would look like to the combine kernel) -- one contiguous region per
// chunk instead of NR + NW streams 95 MB apart.  A thread still owns one float4 column of all rows.
template <int NR, int NW, int CH>
__global__ __launch_bounds__(256) void probe_blocked(const float* __restrict__ in, float* __restrict__ out, int64_t n4) {
  constexpr int C4 = CH / 4;
  const int64_t step = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) {
    const int64_t chunk = i / C4, c = i % C4;
    const float* src = in + chunk * (int64_t)NR * CH + 4 * c;
    float* dst = out + chunk * (int64_t)NW * CH + 4 * c;
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < NR; ++r) acc += ld4_nt(src + (int64_t)r * CH);
#pragma unroll
    for (int w = 0; w < NW; ++w) st4_nt(dst + (int64_t)w * CH, acc + (float)w);
  }
}
// Adjacent pairs: each lane owns TWO adjacent float4 columns (32 contiguous bytes per row), rows 95 MB apart.
template <int NR, int NW>
__global__ __launch_bounds__(256) void probe_pair(const float* __restrict__ in, float* __restrict__ out, int64_t ld, int64_t n4) {
  const int64_t step = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; 2 * i + 1 < n4; i += step) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      a0 += ld4_nt(in + r * ld + 8 * i);
      a1 += ld4_nt(in + r * ld + 8 * i + 4);
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      st4_nt(out + w * ld + 8 * i, a0 + (float)w);
      st4_nt(out + w * ld + 8 * i + 4, a1 + (float)w);
    }
  }
}

// Shape of the fused SVGD step: M particle rows read-modify-written in place, M gradient rows read, one state row
// read-modify-written.  PB / GB: that operand stored blocked ([chunk][M][CH]) instead of as M rows `ld` apart.
template <int M, bool PB, bool GB, int CH>
__global__ __launch_bounds__(256) void probe_fused(float* __restrict__ P, const float* __restrict__ G, float* __restrict__ buf,
                                                   int64_t ld, int64_t n4) {
  constexpr int C4 = CH / 4;
  const int64_t step = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) {
    const int64_t chunk = i / C4, c = i % C4;
    const int64_t pb = PB ? chunk * (int64_t)M * CH + 4 * c : 4 * i, ps = PB ? CH : ld;
    const int64_t gb = GB ? chunk * (int64_t)M * CH + 4 * c : 4 * i, gs = GB ? CH : ld;
    f32x4 p[M];
    f32x4 acc = ld4(buf + 4 * i);
#pragma unroll
    for (int r = 0; r < M; ++r) {
      p[r] = ld4_nt(P + pb + r * ps);
      acc += ld4_nt(G + gb + r * gs);
    }
#pragma unroll
    for (int r = 0; r < M; ++r) st4_nt(P + pb + r * ps, p[r] * 0.999f + acc * 1e-6f);
    st4(buf + 4 * i, acc * 0.5f);
  }
}

// ----------------------------------------------------------- swag sample variants --
__device__ __forceinline__ f32x4 dstd(f32x4 m, f32x4 s) {
  f32x4 v = s - m * m, r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = __builtin_sqrtf(0.5f * (fmaxf(v[j], 0.0f) + 1e-6f));
  return r;
}
template <int U, bool CONTIG, int RB>
__global__ __launch_bounds__(256) void sample_v(const float* __restrict__ mean, const float* __restrict__ sq, const float* __restrict__ dev, int K,
                                                int64_t ld, const float* __restrict__ wg, uint64_t seed, uint64_t stream_id, float* __restrict__ out, int64_t D) {
  __shared__ float w[BDE_MAX_RANK];
  for (int r = threadIdx.x; r < K; r += blockDim.x) w[r] = wg[r];
  __syncthreads();
  const int64_t n4 = D >> 2;
  int64_t lo, hi, step;
  if (CONTIG) {
    const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
    lo = blockIdx.x * per + threadIdx.x; hi = std::min<int64_t>(n4, (blockIdx.x + 1) * per); step = 256;
  } else {
    lo = (int64_t)blockIdx.x * 256 + threadIdx.x; hi = n4; step = (int64_t)gridDim.x * 256;
  }
  for (int64_t i = lo; i < hi; i += step * U) {
    f32x4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = f32x4{0, 0, 0, 0};
#pragma unroll RB
    for (int r = 0; r < K; ++r) {
      const float wr = w[r];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t c = i + u * step;
        if (c < hi) {
          const f32x4 d = ld4_nt(dev + (int64_t)r * ld + 4 * c);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[u][j] = __builtin_fmaf(d[j], wr, acc[u][j]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t c = i + u * step;
      if (c < hi) {
        const f32x4 m = ld4_nt(mean + 4 * c), s = ld4_nt(sq + 4 * c);
        const f32x4 z = philox_normal4(seed, stream_id, (uint64_t)c, kDomainDiag);
        st4_nt(out + 4 * c, (m + acc[u]) + dstd(m, s) * z);
      }
    }
  }
}

// ----------------------------------------------------------- gauss draw variants --
template <int U, bool CONTIG, bool PLAINST>
__global__ __launch_bounds__(256) void draw_v(const float* __restrict__ mean, const float* __restrict__ rho, uint64_t seed, uint64_t stream_id,
                                              float* __restrict__ w, int64_t n) {
  const int64_t n4 = n >> 2;
  int64_t lo, hi, step;
  if (CONTIG) {
    const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
    lo = blockIdx.x * per + threadIdx.x; hi = std::min<int64_t>(n4, (blockIdx.x + 1) * per); step = 256;
  } else {
    lo = (int64_t)blockIdx.x * 256 + threadIdx.x; hi = n4; step = (int64_t)gridDim.x * 256;
  }
  for (int64_t i = lo; i < hi; i += step * U) {
    f32x4 m[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t c = i + u * step;
      if (c < hi) { m[u] = ld4_nt(mean + 4 * c); r[u] = ld4_nt(rho + 4 * c); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t c = i + u * step;
      if (c < hi) {
        const f32x4 e = philox_normal4(seed, stream_id, (uint64_t)c, kDomainDiag);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = m[u][j] + e[j] * softplus(r[u][j]);
        if (PLAINST) st4(w + 4 * c, o); else st4_nt(w + 4 * c, o);
      }
    }
  }
}

// ------------------------------------------------------------------ gram variants --
template <int UU, int WPB>
__global__ __launch_bounds__(WPB * 64) void gram_v(const float* __restrict__ P, int M, int64_t D, int64_t ld, float* __restrict__ ws) {
  constexpr int W4 = 8;
  __shared__ float tile[WPB][16][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int c4 = (r16 >> 3) * 4 + kq;
  const int prow = r16 & 7;
  const bool valid = prow < M;
  const float inv_m = 1.0f / (float)M;
  const float* rowp = P + (int64_t)(valid ? prow : 0) * ld;
  const int64_t n4 = D >> 2;                      // experiments: full columns only
  const int64_t tile4 = (int64_t)UU * W4;
  const int64_t n_tiles = n4 / tile4;
  const int64_t waves_total = (int64_t)gridDim.x * WPB;
  f32x4acc acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  f32x4 cur[UU], nxt[UU];
  int64_t t = (int64_t)blockIdx.x * WPB + wave;
  if (t < n_tiles) {
#pragma unroll
    for (int u = 0; u < UU; ++u) cur[u] = ld4_nt(rowp + 4 * (t * tile4 + c4 + u * W4));
  }
  for (; t < n_tiles; t += waves_total) {
    const int64_t tn = t + waves_total;
    if (tn < n_tiles) {
#pragma unroll
      for (int u = 0; u < UU; ++u) nxt[u] = ld4_nt(rowp + 4 * (tn * tile4 + c4 + u * W4));
    }
#pragma unroll
    for (int u = 0; u < UU; ++u) {
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
    for (int u = 0; u < UU; ++u) cur[u] = nxt[u];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) tile[wave][4 * kq + r][r16] = acc0[r] + acc1[r];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int pi = threadIdx.x / 8, pj = threadIdx.x % 8;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) s += tile[w][pi][pj] + tile[w][pi + 8][pj + 8];
    ws[64 + (int64_t)blockIdx.x * 64 + threadIdx.x] = s;
  }
}

struct Variant { std::string name; std::function<void()> fn; double bytes; };

static void run_table(const char* title, std::vector<Variant>& vs, hipStream_t st, int rounds = 7, int inner = 5, bool us = false) {
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
  printf("\n== %s\n%-46s %10s %10s %9s %9s\n", title, "variant", us ? "min us" : "min ms", us ? "med us" : "med ms", "TB/s(min)", "TB/s(med)");
  for (size_t v = 0; v < vs.size(); ++v) {
    auto t = times[v]; std::sort(t.begin(), t.end());
    const double k = us ? 1e3 : 1.0;
    printf("%-46s %10.4f %10.4f %9.3f %9.3f\n", vs[v].name.c_str(), t[0] * k, t[t.size() / 2] * k, vs[v].bytes / (t[0] * 1e-3) / 1e12,
           vs[v].bytes / (t[t.size() / 2] * 1e-3) / 1e12);
  }
  fflush(stdout);
}

int main(int argc, char** argv) {
  const char* only = argc > 1 ? argv[1] : "all";
  auto want = [&](const char* s) { return !strcmp(only, "all") || !strcmp(only, s); };
  const int K = 20, M = 8;
  const int64_t D = 23880950, ld = (D + 16 + 63) / 64 * 64, n4 = D >> 2;
  hipStream_t st; CK(hipStreamCreate(&st));
  float *in, *out;
  const int NROWS = 24;
  CK(hipMalloc(&in, sizeof(float) * NROWS * ld));
  CK(hipMalloc(&out, sizeof(float) * 32 * ld));
  {
    std::vector<float> h(ld);
    uint32_t s = 12345;
    for (int i = 0; i < NROWS; ++i) {
      for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f)) * 0.01f + 0.001f; }
      CK(hipMemcpy(in + (int64_t)i * ld, h.data(), sizeof(float) * ld, hipMemcpyHostToDevice));
    }
  }
  float* wg; CK(hipMalloc(&wg, 1024)); CK(hipMemcpy(wg, in, 1024, hipMemcpyDeviceToDevice));

  if (want("probe")) {
    std::vector<Variant> vs;
#define PRB(NR, NW, U, C, NT, G) vs.push_back({std::string("probe R" #NR " W" #NW " U" #U) + (C ? " contig" : " stride") + (NT ? " nt" : " plain") + " g" #G, \
      [&] { hipLaunchKernelGGL((probe<NR, NW, U, C, NT>), dim3(G), dim3(256), 0, st, in, out, ld, n4); }, 4.0 * D * (NR + NW)});
#define PRB_SET(NR, NW) PRB(NR, NW, 1, false, true, 2048) PRB(NR, NW, 2, false, true, 2048) PRB(NR, NW, 4, false, true, 2048) \
      PRB(NR, NW, 1, true, true, 2048) PRB(NR, NW, 2, true, true, 2048) PRB(NR, NW, 4, true, true, 2048) PRB(NR, NW, 2, true, true, 1024) \
      PRB(NR, NW, 2, true, true, 4096) PRB(NR, NW, 2, true, false, 2048) PRB(NR, NW, 4, true, true, 1024) PRB(NR, NW, 2, false, true, 4096) \
      PRB(NR, NW, 2, true, true, 8192) PRB(NR, NW, 1, true, true, 8192)
    PRB_SET(1, 0) PRB_SET(1, 1) PRB_SET(2, 1) PRB_SET(3, 2) PRB_SET(5, 3)
    PRB(22, 0, 1, false, true, 2048) PRB(22, 0, 1, true, true, 2048) PRB(22, 0, 2, true, true, 2048) PRB(22, 0, 1, true, true, 4096)
    PRB(22, 1, 1, false, true, 2048) PRB(22, 1, 1, true, true, 2048) PRB(22, 1, 2, true, true, 2048) PRB(22, 1, 1, true, true, 4096) PRB(22, 1, 1, true, true, 1024)
    PRB(8, 0, 1, false, true, 2048) PRB(8, 0, 1, true, true, 2048) PRB(8, 0, 2, true, true, 2048)
    PRB(16, 8, 1, false, true, 2048) PRB(16, 8, 1, true, true, 2048) PRB(16, 8, 1, true, true, 1024) PRB(16, 8, 1, true, true, 4096)
    PRB(22, 30, 1, true, true, 2048) PRB(22, 30, 1, false, true, 2048)
    run_table("HBM probes at D = 23,880,950 (bytes = 4 D (NR + NW))", vs, st);
  }

  if (want("layout")) {
    std::vector<Variant> vs;
    const double B16 = 4.0 * D * 24;
    vs.push_back({"R16 W8 rows 95 MB apart, stride g2048", [&] { hipLaunchKernelGGL((probe<16, 8, 1, false, true>), dim3(2048), dim3(256), 0, st, in, out, ld, n4); }, B16});
    vs.push_back({"R16 W8 adjacent float4 pairs g2048", [&] { hipLaunchKernelGGL((probe_pair<16, 8>), dim3(2048), dim3(256), 0, st, in, out, ld, n4); }, B16});
    vs.push_back({"R16 W8 adjacent float4 pairs g1024", [&] { hipLaunchKernelGGL((probe_pair<16, 8>), dim3(1024), dim3(256), 0, st, in, out, ld, n4); }, B16});
#define PBL(CH, G) vs.push_back({"R16 W8 blocked layout chunk " #CH " floats g" #G, [&] { hipLaunchKernelGGL((probe_blocked<16, 8, CH>), dim3(G), dim3(256), 0, st, in, out, n4); }, B16});
    PBL(256, 2048) PBL(1024, 2048) PBL(4096, 2048) PBL(16384, 2048) PBL(65536, 2048) PBL(1024, 4096) PBL(4096, 1024)
    vs.push_back({"R8 W0 rows apart (gram's shape)", [&] { hipLaunchKernelGGL((probe<8, 0, 1, false, true>), dim3(2048), dim3(256), 0, st, in, out, ld, n4); }, 4.0 * D * 8});
#define PBR(CH) vs.push_back({"R8 W0 blocked chunk " #CH, [&] { hipLaunchKernelGGL((probe_blocked<8, 0, CH>), dim3(2048), dim3(256), 0, st, in, out, n4); }, 4.0 * D * 8});
    PBR(1024) PBR(16384)
    vs.push_back({"R22 W30 rows apart (batched sampler's shape)", [&] { hipLaunchKernelGGL((probe<22, 30, 1, false, true>), dim3(2048), dim3(256), 0, st, in, out, ld, n4); }, 4.0 * D * 52});
    vs.push_back({"R22 W30 blocked chunk 4096", [&] { hipLaunchKernelGGL((probe_blocked<22, 30, 4096>), dim3(2048), dim3(256), 0, st, in, out, n4); }, 4.0 * D * 52});
    run_table("layout probes: separate rows vs rows interleaved per chunk (bytes = 4 D (NR + NW))", vs, st);
    {
      std::vector<Variant> vf;
      float* Pm = out;                      // 8 rows of the 32-row output buffer as the in-place particle matrix
      float* buf = out + 16 * ld;
      const double BF = 4.0 * D * (3 * 8 + 2);
#define PF(PB, GB, CH, G) vf.push_back({std::string("fused shape P ") + (PB ? "blocked" : "rows") + " G " + (GB ? "blocked" : "rows") + " chunk " #CH " g" #G, \
        [&] { hipLaunchKernelGGL((probe_fused<8, PB, GB, CH>), dim3(G), dim3(256), 0, st, Pm, in, buf, ld, n4); }, BF});
      PF(false, false, 4096, 2048) PF(false, true, 4096, 2048) PF(true, true, 4096, 2048) PF(false, true, 16384, 2048) PF(true, true, 16384, 2048)
      PF(false, true, 1024, 2048) PF(false, false, 4096, 1024) PF(false, true, 4096, 1024)
      run_table("fused-step shape: particles RMW in place + gradients read + state RMW (bytes = 4 D (3 M + 2))", vf, st);
    }
  }

  if (want("ldsweep")) {
    // does the row stride matter?  same R16 W8 / fused shapes, rows `ld + pad` floats apart (the buffers are large
    // enough: 24 input rows and 32 output rows were allocated with the base ld)
    std::vector<Variant> vs;
    const double B16 = 4.0 * D * 24;
    for (int64_t pad : {0LL, 64LL, 192LL, 448LL, 1024LL, 1088LL, 4096LL, 4160LL, 16448LL, 65600LL, 262208LL, 524352LL}) {
      const int64_t l2 = ld + pad;
      if (16 * l2 > (int64_t)NROWS * ld || 8 * l2 > 32 * ld) continue;
      vs.push_back({"R16 W8 rows, ld + " + std::to_string(pad) + " floats", [=] { hipLaunchKernelGGL((probe<16, 8, 1, false, true>), dim3(2048), dim3(256), 0, st, in, out, l2, n4); }, B16});
    }
    run_table("row-stride sweep (bytes = 4 D 24)", vs, st);
  }

  if (want("sample")) {
    float *mean = in + (int64_t)K * ld, *sq = mean + ld;
    std::vector<Variant> vs;
    const double B = 4.0 * D * (K + 3);
    vs.push_back({"product bde_swag_sample", [&] { bde_swag_sample(mean, sq, in, K, ld, 3, nullptr, nullptr, 1, 2, out, D, st); }, B});
#define SV(U, C, RB, G) vs.push_back({std::string("sample U" #U) + (C ? " contig" : " stride") + " rb" #RB " g" #G, \
      [&] { hipLaunchKernelGGL((sample_v<U, C, RB>), dim3(G), dim3(256), 0, st, mean, sq, in, K, ld, wg, 1, 2, out, D); }, B});
    SV(1, false, 10, 2048) SV(1, true, 10, 2048) SV(1, true, 20, 2048) SV(1, true, 5, 2048) SV(2, true, 10, 2048) SV(2, true, 5, 2048) SV(2, false, 5, 2048)
    SV(1, true, 10, 4096) SV(1, true, 10, 1024) SV(2, true, 10, 1024) SV(1, true, 10, 8192) SV(1, true, 20, 4096) SV(2, true, 4, 2048)
    run_table("swag_sample variants (K = 20, bytes = 4 D (K + 3))", vs, st);
  }

  if (want("draw")) {
    float *mean = in, *rho = in + ld;
    CK(hipMemset(rho, 0xC0, sizeof(float) * ld));      // 0xC0C0C0C0 = -6.02
    std::vector<Variant> vs;
    const double B = 12.0 * D;
    vs.push_back({"product bde_gauss_draw_fwd", [&] { bde_gauss_draw_fwd(mean, rho, nullptr, 1, 0, out, nullptr, D, st); }, B});
    vs.push_back({"product bde_local_reparam_fwd", [&] { bde_local_reparam_fwd(mean, in + 2 * ld, nullptr, 1, 0, out, D, st); }, B});
#define DV(U, C, P, G) vs.push_back({std::string("draw U" #U) + (C ? " contig" : " stride") + (P ? " plainst" : " ntst") + " g" #G, \
      [&] { hipLaunchKernelGGL((draw_v<U, C, P>), dim3(G), dim3(256), 0, st, mean, rho, 1, 0, out, D); }, B});
    DV(1, false, false, 2048) DV(1, true, false, 2048) DV(2, true, false, 2048) DV(4, true, false, 2048) DV(2, false, false, 2048) DV(2, true, true, 2048)
    DV(2, true, false, 1024) DV(2, true, false, 4096) DV(4, true, false, 1024) DV(4, true, false, 4096) DV(2, true, false, 8192) DV(1, true, false, 8192)
    run_table("gauss_draw_fwd variants (bytes = 12 D)", vs, st);
  }

  if (want("gram")) {
    std::vector<Variant> vs;
    const double B = 4.0 * M * D;
    float* ws; CK(hipMalloc(&ws, sizeof(float) * (64 + 4096 * 64)));
    CK(hipMemset(ws, 0, sizeof(float) * (64 + 4096 * 64)));
    float* P; CK(hipMalloc(&P, sizeof(float) * M * ld));
    CK(hipMemcpy(P, in, sizeof(float) * M * ld, hipMemcpyDeviceToDevice));
    vs.push_back({"product bde_svgd_gram", [&] { bde_svgd_gram(P, M, D, ld, ws, st); }, B});
#define GV(UU, WPB, G) vs.push_back({"gram U" #UU " waves" #WPB " g" #G, [&] { hipLaunchKernelGGL((gram_v<UU, WPB>), dim3(G), dim3(WPB * 64), 0, st, P, M, D, ld, ws); }, B});
    GV(4, 4, 1024) GV(8, 4, 1024) GV(8, 4, 2048) GV(4, 4, 2048) GV(8, 8, 512) GV(8, 8, 1024) GV(8, 4, 512) GV(4, 8, 1024) GV(2, 4, 2048) GV(8, 2, 2048) GV(8, 4, 768)
    vs.push_back({"probe R8 stride g2048 (same bytes)", [&] { hipLaunchKernelGGL((probe<8, 0, 1, false, true>), dim3(2048), dim3(256), 0, st, P, out, ld, n4); }, B});
    run_table("svgd_gram variants (M = 8, bytes = 4 M D)", vs, st);
    // combine + step
    float *G, *o, *ks;
    CK(hipMalloc(&G, sizeof(float) * M * ld)); CK(hipMalloc(&o, sizeof(float) * M * ld)); CK(hipMalloc(&ks, 4096));
    CK(hipMemcpy(G, in + 8 * ld, sizeof(float) * M * ld, hipMemcpyDeviceToDevice));
    bde_svgd_step(P, G, o, M, D, ld, 0.f, 1.f, 129809.f, -1.f, ws, ks, st);
    std::vector<Variant> v2;
    v2.push_back({"product bde_svgd_combine", [&] { bde_svgd_combine(P, G, o, M, D, ld, ld, ks, st); }, 12.0 * M * D});
    v2.push_back({"product bde_svgd_step", [&] { bde_svgd_step(P, G, o, M, D, ld, 0.f, 1.f, 129809.f, -1.f, ws, ks, st); }, 16.0 * M * D});
    v2.push_back({"probe R16 W8 contig g2048 (combine's bytes)", [&] { hipLaunchKernelGGL((probe<16, 8, 1, true, true>), dim3(2048), dim3(256), 0, st, in, out, ld, n4); }, 12.0 * M * D});
    run_table("svgd combine / step at D = 23,880,950", v2, st);
    CK(hipFree(G)); CK(hipFree(o)); CK(hipFree(P));
  }

  if (want("small")) {
    const int64_t d = 273610, l = (d + 16 + 63) / 64 * 64;
    float *P, *G, *o, *ws, *ks;
    CK(hipMalloc(&P, sizeof(float) * M * l)); CK(hipMalloc(&G, sizeof(float) * M * l)); CK(hipMalloc(&o, sizeof(float) * M * l));
    CK(hipMalloc(&ws, bde_svgd_ws_bytes(M))); CK(hipMemset(ws, 0, bde_svgd_ws_bytes(M))); CK(hipMalloc(&ks, 4096));
    for (int i = 0; i < M; ++i) {
      CK(hipMemcpy(P + i * l, in + i * ld, sizeof(float) * l, hipMemcpyDeviceToDevice));
      CK(hipMemcpy(G + i * l, in + (8 + i) * ld, sizeof(float) * l, hipMemcpyDeviceToDevice));
    }
    std::vector<Variant> vs;
    const double B = 16.0 * M * d;
    vs.push_back({"single launch bde_svgd_step_small", [&] { bde_svgd_step_small(P, G, o, M, d, l, 3e-4f, 1.f, 50000.f, -1.f, 0.f, 0, ws, ks, st); }, B});
    vs.push_back({"three stages gram+kstats+combine", [&] { bde_svgd_gram(P, M, d, l, ws, st); bde_svgd_kstats(ws, M, 3e-4f, 1.f, 50000.f, -1.f, 0.f, 0, ks, st);
                                                          bde_svgd_combine(P, G, o, M, d, l, l, ks, st); }, B});
    vs.push_back({"  gram only", [&] { bde_svgd_gram(P, M, d, l, ws, st); }, 4.0 * M * d});
    vs.push_back({"  kstats only", [&] { bde_svgd_kstats(ws, M, 3e-4f, 1.f, 50000.f, -1.f, 0.f, 0, ks, st); }, 0});
    vs.push_back({"  combine only", [&] { bde_svgd_combine(P, G, o, M, d, l, l, ks, st); }, 12.0 * M * d});
    vs.push_back({"single launch, in place (out = G)", [&] { bde_svgd_step_small(P, G, G, M, d, l, 3e-4f, 1.f, 50000.f, -1.f, 0.f, 0, ws, ks, st); }, B});
    run_table("SVGD step at D = 273,610 (CIFAR ResNet-20), M = 8; 16 M D = 35.0 MB algorithmic", vs, st, 9, 50, true);
  }

  if (want("batched")) {
    float *mean = in + (int64_t)K * ld, *sq = mean + ld;
    std::vector<Variant> vs;
    const int S = 30;
    vs.push_back({"product bde_swag_sample_batched S30", [&] { bde_swag_sample_batched(mean, sq, in, K, ld, 3, nullptr, nullptr, 1, 0, out, ld, S, D, st); }, 4.0 * D * (K + 2 + S)});
    vs.push_back({"product batched S30 supplied eps_d", [&] { bde_swag_sample_batched(mean, sq, in, K, ld, 3, nullptr, out, 1, 0, out, ld, S, D, st); }, 4.0 * D * (K + 2 + 2 * S)});
    vs.push_back({"probe R22 W30 grid-stride 4 KB per row per WG g2048", [&] { hipLaunchKernelGGL((probe<22, 30, 1, false, true>), dim3(2048), dim3(256), 0, st, in, out, ld, n4); }, 4.0 * D * (K + 2 + S)});
    vs.push_back({"probe R22 W30 half-wave rows (512 B), g1024", [&] { hipLaunchKernelGGL((probe_half<22, 30, false>), dim3(1024), dim3(256), 0, st, in, out, ld, n4); }, 4.0 * D * (K + 2 + S)});
    vs.push_back({"probe R22 W30 half-wave rows (512 B), g2048", [&] { hipLaunchKernelGGL((probe_half<22, 30, false>), dim3(2048), dim3(256), 0, st, in, out, ld, n4); }, 4.0 * D * (K + 2 + S)});
    vs.push_back({"probe R22 W30 whole-wave rows (1 KB), g1024", [&] { hipLaunchKernelGGL((probe_half<22, 30, true>), dim3(1024), dim3(256), 0, st, in, out, ld, n4); }, 4.0 * D * (K + 2 + S)});
    vs.push_back({"probe R22 W30 whole-wave rows (1 KB), g2048", [&] { hipLaunchKernelGGL((probe_half<22, 30, true>), dim3(2048), dim3(256), 0, st, in, out, ld, n4); }, 4.0 * D * (K + 2 + S)});
    vs.push_back({"probe R22 W30 contig (same bytes)", [&] { hipLaunchKernelGGL((probe<22, 30, 1, true, true>), dim3(2048), dim3(256), 0, st, in, out, ld, n4); }, 4.0 * D * (K + 2 + S)});
    run_table("swag_sample_batched (K = 20, S = 30)", vs, st, 5, 3);
  }
  return 0;
}
