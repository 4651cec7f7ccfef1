// TEST INFRASTRUCTURE: a CPU execution model for the kernels of beyond_deep_ensembles_amd/csrc/*.hip.
//
// The product is gfx950 code and has no CPU path; this header exists so that the *unchanged kernel sources* (and the
// host planners / C-ABI entry points in the same files) can be executed in this GPU-less container by the `not gpu`
// tests: every lane of a workgroup is a fiber, `__syncthreads()` and the wave-level operations (MFMA, shuffles, DPP,
// readlane, ballot, LDS-DMA) are real rendezvous between the fibers with the CDNA4 register layouts of
// /opt/skills/guides/cdna_hip_programming.md, dynamic LDS is poisoned with NaNs before each workgroup and ends at a
// guard page, buffers handed out by hip_emu_alloc() are fenced by guard pages on both sides, and an LDS-DMA request
// only lands at the `s_waitcnt vmcnt` that covers it (a kernel that reads the slab early reads NaNs).  tests/hip_emu/
// build.py compiles the sources against this header with the host clang; nothing here is shipped or measured.
#pragma once
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>

#undef __shared__
#define __shared__ static thread_local
#undef __launch_bounds__
#define __launch_bounds__(...)

namespace hip_emu {

struct Idx {
  unsigned x, y, z;
};

struct Wave;
struct Lane {
  Idx tid, bid;
  dim3 bdim, gdim;
  int lane;          // 0..63
  int linear;        // thread index in the workgroup
  unsigned parity;   // which exchange slot the next wave-level operation uses
  Wave* wave;
};

extern thread_local Lane* cur;

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body, const void* kernel = nullptr);
void block_sync();
void wave_sync();
void* dyn_lds();
void waitcnt_vm(int outstanding);
void dma_request(const void* src, void* lds_dst, int bytes);
uint64_t* wave_slot(unsigned parity);     // [64][2] 64-bit words of the current wave
uint64_t wave_live_mask();
void count_mfma(int which);

template <typename T>
inline T exchange(T mine, int from_lane) {   // every live lane of the wave calls this; returns lane `from_lane`'s value
  static_assert(sizeof(T) <= 8, "exchange: at most 8 bytes");
  Lane* me = cur;
  const unsigned p = me->parity;
  me->parity ^= 1u;
  uint64_t* s = wave_slot(p);
  uint64_t w = 0;
  std::memcpy(&w, &mine, sizeof(T));
  s[me->lane * 2] = w;
  wave_sync();
  T out;
  std::memcpy(&out, &s[(from_lane & 63) * 2], sizeof(T));
  return out;
}

template <typename T>
inline T shfl_down(T v, unsigned off, int width) {
  const int lane = cur->lane;
  const int src = lane + static_cast<int>(off);
  const bool ok = (src / width) == (lane / width) && src < 64;
  const T got = exchange(v, ok ? src : lane);
  return got;
}
template <typename T>
inline T shfl(T v, int src, int width) {
  const int lane = cur->lane;
  return exchange(v, (lane / width) * width + (src % width));
}

inline unsigned long long ballot(bool pred) {
  Lane* me = cur;
  const unsigned p = me->parity;
  me->parity ^= 1u;
  uint64_t* s = wave_slot(p);
  s[me->lane * 2] = pred ? 1u : 0u;
  wave_sync();
  unsigned long long m = 0;
  const uint64_t live = wave_live_mask();
  for (int l = 0; l < 64; ++l)
    if (((live >> l) & 1u) && s[l * 2]) m |= 1ull << l;
  return m;
}

template <typename T>
inline T readfirstlane(T v) {
  // lowest live lane of the wave (the kernels only call this with all lanes active)
  const uint64_t live = wave_live_mask();
  return exchange(v, __builtin_ctzll(live));
}

// v_mov_b32 with a DPP control, bound_ctrl = true, full row / bank masks: the controls the kernels use.
inline int update_dpp(int /*old*/, int src, int ctrl, int /*row_mask*/, int /*bank_mask*/, bool /*bound_ctrl*/) {
  const int lane = cur->lane;
  int from = lane;
  if (ctrl >= 0 && ctrl <= 0xFF) {                 // quad_perm
    from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  } else if (ctrl == 0x140) {                      // row_mirror: 16 lanes reversed
    from = (lane & ~15) | (15 - (lane & 15));
  } else if (ctrl == 0x141) {                      // row_half_mirror: 8 lanes reversed
    from = (lane & ~7) | (7 - (lane & 7));
  } else {
    std::abort();
  }
  return exchange(src, from);
}

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: A[i][k] from lane k*32 + i, B[k][j] from lane k*32 + j; lane l holds D[8*(v/4) + 4*(l/32) + v%4][l%32].
inline floatx16 mfma_32x32x2f32(float a, float b, floatx16 c, int, int, int) {
  Lane* me = cur;
  const unsigned p = me->parity;
  me->parity ^= 1u;
  float* s = reinterpret_cast<float*>(wave_slot(p));     // [64][4] floats
  s[me->lane * 4] = a;
  s[me->lane * 4 + 1] = b;
  wave_sync();
  count_mfma(0);
  const int l = me->lane, j = l & 31, hi = l >> 5;
  for (int v = 0; v < 16; ++v) {
    const int i = 8 * (v / 4) + 4 * hi + (v % 4);
    float acc = c[v];
    for (int k = 0; k < 2; ++k) acc = std::fma(s[(k * 32 + i) * 4], s[(k * 32 + j) * 4 + 1], acc);
    c[v] = acc;
  }
  return c;
}
// v_mfma_f32_16x16x4_f32: A[i][k] from lane k*16 + i, B[k][j] from lane k*16 + j; lane l holds D[4*(l/16) + v][l%16].
inline floatx4 mfma_16x16x4f32(float a, float b, floatx4 c, int, int, int) {
  Lane* me = cur;
  const unsigned p = me->parity;
  me->parity ^= 1u;
  float* s = reinterpret_cast<float*>(wave_slot(p));
  s[me->lane * 4] = a;
  s[me->lane * 4 + 1] = b;
  wave_sync();
  count_mfma(1);
  const int l = me->lane, j = l & 15, g = l >> 4;
  for (int v = 0; v < 4; ++v) {
    const int i = 4 * g + v;
    float acc = c[v];
    for (int k = 0; k < 4; ++k) acc = std::fma(s[(k * 16 + i) * 4], s[(k * 16 + j) * 4 + 1], acc);
    c[v] = acc;
  }
  return c;
}

inline void global_load_lds(const __attribute__((address_space(1))) void* src, __attribute__((address_space(3))) void* dst, int bytes,
                            int offset, int /*aux*/) {
  // M0 = the wave's (uniform) LDS base; lane l's `bytes` land at base + offset + l * bytes
  char* base = readfirstlane(reinterpret_cast<char*>((void*)dst));
  dma_request((const void*)src, base + offset + cur->lane * bytes, bytes);
}

inline uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c, uint32_t table) {
  uint32_t out = 0;
  for (int bit = 0; bit < 32; ++bit) {
    const uint32_t idx = (((a >> bit) & 1u) << 2) | (((b >> bit) & 1u) << 1) | ((c >> bit) & 1u);
    out |= ((table >> idx) & 1u) << bit;
  }
  return out;
}

}  // namespace hip_emu

#define threadIdx (hip_emu::cur->tid)
#define blockIdx (hip_emu::cur->bid)
#define blockDim (hip_emu::cur->bdim)
#define gridDim (hip_emu::cur->gdim)

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
  hip_emu::launch(dim3(grid), dim3(block), static_cast<size_t>(lds), [=]() { kernel(__VA_ARGS__); }, reinterpret_cast<const void*>(kernel))

inline void __syncthreads() { hip_emu::block_sync(); }

#define __builtin_amdgcn_mfma_f32_32x32x2f32 hip_emu::mfma_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hip_emu::mfma_16x16x4f32
#define __builtin_amdgcn_readfirstlane(v) hip_emu::readfirstlane(v)
#define __builtin_amdgcn_readlane(v, l) hip_emu::exchange((v), (l))
#define __builtin_amdgcn_update_dpp hip_emu::update_dpp
#define __builtin_amdgcn_global_load_lds hip_emu::global_load_lds
#define __builtin_amdgcn_bitop3_b32 hip_emu::bitop3
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_sqrtf(x) std::sqrt(static_cast<float>(x))
#define __builtin_amdgcn_logf(x) std::log2(static_cast<float>(x))
#define __builtin_amdgcn_rcpf(x) (1.0f / static_cast<float>(x))
#define __builtin_amdgcn_cosf(x) static_cast<float>(std::cos(6.283185307179586 * static_cast<double>(x)))
#define __builtin_amdgcn_sinf(x) static_cast<float>(std::sin(6.283185307179586 * static_cast<double>(x)))

#define __shfl_down(v, off, width) hip_emu::shfl_down((v), (off), (width))
#define __shfl(v, src, width) hip_emu::shfl((v), (src), (width))
#define __ballot(p) hip_emu::ballot(p)

inline float __logf(float x) { return std::log(x); }
inline float __expf(float x) { return std::exp(x); }
inline float __int_as_float(int v) { float f; std::memcpy(&f, &v, 4); return f; }
inline float __uint_as_float(unsigned v) { float f; std::memcpy(&f, &v, 4); return f; }
inline int __float_as_int(float f) { int v; std::memcpy(&v, &f, 4); return v; }
inline unsigned __float_as_uint(float f) { unsigned v; std::memcpy(&v, &f, 4); return v; }

#include <algorithm>
using std::max;
using std::min;

#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
namespace hip_emu {
template <typename T>
inline void atomic_store(T* p, T v) { __atomic_store(p, &v, __ATOMIC_SEQ_CST); }
template <typename T>
inline T atomic_load(const T* p) { T v; __atomic_load(const_cast<T*>(p), &v, __ATOMIC_SEQ_CST); return v; }
}  // namespace hip_emu
#define __hip_atomic_store(p, v, order, scope) hip_emu::atomic_store((p), (v))
#define __hip_atomic_load(p, order, scope) hip_emu::atomic_load((p))
