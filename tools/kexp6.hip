// Round-2 experiments, part 2 (development tool): per-phase timestamps of the single-launch SVGD step,
// A/B of load/store flavours of the fused SVGD kernel and of the sampling kernels' output stores.
// Self-contained: includes the product sources (optionally with -D switches), does not link libbde_hip.
#include "../beyond_deep_ensembles_amd/csrc/svgd.hip"
#include "../beyond_deep_ensembles_amd/csrc/svgd_small.hip"
#include "../beyond_deep_ensembles_amd/csrc/svgd_fused.hip"
#include "../beyond_deep_ensembles_amd/csrc/swag.hip"
#include "../beyond_deep_ensembles_amd/csrc/gauss.hip"
#include <cstdio>
#include <cstring>
#include <vector>
#include <functional>
#include <string>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
#ifndef KEXP_TAG
#define KEXP_TAG "default"
#endif
using namespace bde;

struct Variant { std::string name; std::function<void()> fn; double bytes; };
static void run_table(const char* title, std::vector<Variant>& vs, hipStream_t st, int rounds, int inner, bool us) {
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
  printf("\n== [%s] %s\n%-46s %10s %10s %9s\n", KEXP_TAG, title, "variant", us ? "min us" : "min ms", us ? "med us" : "med ms", "TB/s(med)");
  for (size_t v = 0; v < vs.size(); ++v) {
    auto t = times[v]; std::sort(t.begin(), t.end());
    const double k = us ? 1e3 : 1.0;
    printf("%-46s %10.4f %10.4f %9.3f\n", vs[v].name.c_str(), t[0] * k, t[t.size() / 2] * k, vs[v].bytes / (t[t.size() / 2] * 1e-3) / 1e12);
  }
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int M = 8;
  hipStream_t st; CK(hipStreamCreate(&st));
  // ------------------------------------------------ small SVGD step
  {
    const int64_t d = 273610, l = (d + 16 + 63) / 64 * 64;
    float *P, *G, *o, *ws, *ks;
    CK(hipMalloc(&P, sizeof(float) * M * l)); CK(hipMalloc(&G, sizeof(float) * M * l)); CK(hipMalloc(&o, sizeof(float) * M * l));
    CK(hipMalloc(&ws, bde_svgd_ws_bytes(M))); CK(hipMemset(ws, 0, bde_svgd_ws_bytes(M))); CK(hipMalloc(&ks, 4096));
    std::vector<float> h(M * l);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f)) * 0.1f - 0.05f; }
    CK(hipMemcpy(P, h.data(), sizeof(float) * M * l, hipMemcpyHostToDevice));
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f)) * 0.02f - 0.01f; }
    CK(hipMemcpy(G, h.data(), sizeof(float) * M * l, hipMemcpyHostToDevice));
    std::vector<Variant> vs;
    const double B = 16.0 * M * d;
    vs.push_back({"single launch bde_svgd_step_small", [&] { bde_svgd_step_small(P, G, o, M, d, l, 3e-4f, 1.f, 50000.f, -1.f, 0.f, 0, ws, ks, st); }, B});
    vs.push_back({"three stages gram+kstats+combine", [&] { bde_svgd_gram(P, M, d, l, ws, st); bde_svgd_kstats(ws, M, 3e-4f, 1.f, 50000.f, -1.f, 0.f, 0, ks, st);
                                                          bde_svgd_combine(P, G, o, M, d, l, l, ks, st); }, B});
    vs.push_back({"  kstats only (small ws)", [&] { bde_svgd_kstats(ws, M, 3e-4f, 1.f, 50000.f, -1.f, 0.f, 0, ks, st); }, 0});
    run_table("SVGD step at D = 273,610, M = 8", vs, st, 9, 50, true);
#ifdef BDE_SMALL_TIMING
    for (int rep = 0; rep < 3; ++rep) {
      bde_svgd_step_small(P, G, o, M, d, l, 3e-4f, 1.f, 50000.f, -1.f, 0.f, 0, ws, ks, st);
      CK(hipStreamSynchronize(st));
      std::vector<unsigned long long> ts(256 * 16);
      CK(hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(g_small_ts), sizeof(unsigned long long) * 256 * 16));
      const int64_t n_tiles = (((d + 3) >> 2) + 31) / 32;
      const int tpw = (int)((n_tiles + 255) / 256);
      const int grid = (int)((n_tiles + tpw - 1) / tpw);
      unsigned long long t0 = ~0ull;
      for (int b = 0; b < grid; ++b) t0 = std::min(t0, ts[b * 16]);
      printf("\n[%s] timestamps rep %d (grid %d, tiles/wg %d), us since the first workgroup started: phase: min / median / max over workgroups\n", KEXP_TAG, rep, grid, tpw);
      const char* names[8] = {"start", "gram done (LDS)", "partial published", "arrived", "go seen", "partials reduced", "stats done", "stores issued"};
      for (int k = 0; k < 8; ++k) {
        std::vector<double> v;
        for (int b = 0; b < grid; ++b) v.push_back((double)(ts[b * 16 + k] - t0) * 0.01);
        std::sort(v.begin(), v.end());
        printf("  %-20s %8.2f %8.2f %8.2f\n", names[k], v[0], v[v.size() / 2], v.back());
      }
    }
#endif
    CK(hipFree(P)); CK(hipFree(G)); CK(hipFree(o)); CK(hipFree(ws)); CK(hipFree(ks));
  }
  // ------------------------------------------------ fused step + sample at ResNet-50 size
  {
    const int K = 20;
    const int64_t D = 23880950, ld = (D + 16 + 63) / 64 * 64;
    float *P, *G, *buf, *ws, *ks;
    CK(hipMalloc(&P, sizeof(float) * M * ld)); CK(hipMalloc(&G, sizeof(float) * M * ld)); CK(hipMalloc(&buf, sizeof(float) * ld));
    CK(hipMalloc(&ws, bde_svgd_ws_bytes(M))); CK(hipMemset(ws, 0, bde_svgd_ws_bytes(M))); CK(hipMalloc(&ks, 4096));
    CK(hipMemset(buf, 0, sizeof(float) * ld));
    {
      std::vector<float> h(ld);
      uint32_t s = 777;
      for (int i = 0; i < M; ++i) {
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f)) * 0.1f - 0.05f; }
        CK(hipMemcpy(P + (int64_t)i * ld, h.data(), sizeof(float) * ld, hipMemcpyHostToDevice));
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f)) * 0.02f - 0.01f; }
        CK(hipMemcpy(G + (int64_t)i * ld, h.data(), sizeof(float) * ld, hipMemcpyHostToDevice));
      }
    }
    bde_svgd_gram(P, M, D, ld, ws, st);
    std::vector<Variant> vs;
    vs.push_back({"gram (product)", [&] { bde_svgd_gram(P, M, D, ld, ws, st); }, 4.0 * M * D});
    vs.push_back({"kstats + fused sgd + next gram", [&] { bde_svgd_kstats(ws, M, 0.f, 1.f, 129809.f, -1.f, 0.f, 0, ks, st);
                                                        bde_svgd_fused_sgd(P, G, buf, M, D, ld, ld, ks, 1e-12, 0.9, 0.0, 3e-4, 1, 0, ws, st); }, (12.0 * M + 8) * D});
    vs.push_back({"kstats + fused sgd (no gram)", [&] { bde_svgd_kstats(ws, M, 0.f, 1.f, 129809.f, -1.f, 0.f, 0, ks, st);
                                                      bde_svgd_fused_sgd(P, G, buf, M, D, ld, ld, ks, 1e-12, 0.9, 0.0, 3e-4, 1, 0, nullptr, st); }, (12.0 * M + 8) * D});
    vs.push_back({"kstats only (full ws)", [&] { bde_svgd_kstats(ws, M, 0.f, 1.f, 129809.f, -1.f, 0.f, 0, ks, st); }, 0});
    vs.push_back({"combine in place (out = G)", [&] { bde_svgd_combine(P, G, G, M, D, ld, ld, ks, st); }, 12.0 * M * D});
    vs.push_back({"step: gram + kstats + combine", [&] { bde_svgd_step(P, G, G, M, D, ld, 0.f, 1.f, 129809.f, -1.f, ws, ks, st); }, 16.0 * M * D});
    run_table("fused SVGD step at D = 23,880,950", vs, st, 7, 5, false);
    // swag sample (mean/sq/ring carved out of P and G)
    float *mean = G, *sq = G + ld, *out = G + 2 * ld;
    std::vector<Variant> v2;
    v2.push_back({"swag_sample K20 (product)", [&] { bde_swag_sample(mean, sq, P, std::min(K, 8), ld, 3, nullptr, nullptr, 1, 2, out, D, st); }, 4.0 * D * (8 + 3)});
    run_table("swag_sample K = 8 rows available here (sanity)", v2, st, 5, 5, false);
    // full-size sampling kernels (store-flavour A/B): a K = 20 ring needs its own allocation
    {
      const int K2 = 20;
      float *ring, *o2;
      CK(hipMalloc(&ring, sizeof(float) * (K2 + 2) * ld)); CK(hipMalloc(&o2, sizeof(float) * ld));
      for (int r = 0; r < K2 + 2; ++r) CK(hipMemcpy(ring + (int64_t)r * ld, P + (int64_t)(r % M) * ld, sizeof(float) * ld, hipMemcpyDeviceToDevice));
      float *mean2 = ring + (int64_t)K2 * ld, *sq2 = mean2 + ld;
      std::vector<Variant> v3;
      v3.push_back({"swag_sample K20", [&] { bde_swag_sample(mean2, sq2, ring, K2, ld, 3, nullptr, nullptr, 1, 2, o2, D, st); }, 4.0 * D * (K2 + 3)});
      v3.push_back({"gauss_draw_fwd", [&] { bde_gauss_draw_fwd(mean2, sq2, nullptr, 1, 0, o2, nullptr, D, st); }, 12.0 * D});
      v3.push_back({"local_reparam_fwd", [&] { bde_local_reparam_fwd(mean2, sq2, nullptr, 1, 0, o2, D, st); }, 12.0 * D});
      run_table("single-output sampling kernels at D = 23,880,950", v3, st, 9, 5, false);
      CK(hipFree(ring)); CK(hipFree(o2));
    }
  }
  return 0;
}
