// TEST INFRASTRUCTURE (see hip_emu.hpp): the fiber scheduler, barriers, LDS / buffer guard pages and the handful of HIP
// runtime entry points the host side of libbde_hip calls.  One OS thread per worker, one workgroup at a time per worker,
// one ucontext fiber per lane; fibers switch only inside a rendezvous (block or wave barrier), round-robin.
#include "hip_emu.hpp"

#include <dlfcn.h>
#include <link.h>
#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <thread>
#include <vector>

namespace hip_emu {

thread_local Lane* cur = nullptr;

struct Dma {
  const void* src;
  void* dst;
  int bytes;
};

// ---- LDS bank-conflict accounting (HIP_EMU_LDS_STATS=1) -----------------------------------------------------------------
// The k-th LDS access of every lane of a wave is one wave-instruction (the lanes take turns access by access, see
// on_access): its 64 addresses, the access size and direction give the LDS-array cycles by the CDNA4 banking rules of
// /opt/skills/guides/MI355X_MICROARCH.md "LDS" -- lane groups per instruction, bank = (a/4) mod 32 or 64, identical
// addresses broadcast, every further distinct address on a busy bank of a group adds a cycle.  Accumulated per code address.
struct LdsSite {
  uint64_t insts = 0, ideal = 0, cycles = 0;
  int size = 0;
  bool write = false;
};
struct LdsPending {
  unsigned tag = ~0u;
  uintptr_t pc = 0;
  int size = 0;
  bool write = false;
  uint64_t mask = 0;
  uintptr_t addr[64];
};

struct Wave {
  int live = 0, arrived = 0;
  unsigned gen = 0;
  uint64_t live_mask = 0;
  alignas(16) uint64_t slot[2][64 * 2];
  LdsPending pend[4];
};

struct Fiber {
  ucontext_t uc;
  Lane lane;
  bool done = true;
  unsigned lds_k = 0;             // LDS accesses of this lane so far in this workgroup
  std::vector<Dma> dma;
};

constexpr size_t kStack = 256 * 1024;
constexpr size_t kLdsMax = 160 * 1024;
static size_t page() { return static_cast<size_t>(sysconf(_SC_PAGESIZE)); }

struct Worker {
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  char* stacks = nullptr;
  size_t n_stacks = 0;
  char* lds_map = nullptr;      // [kLdsMax + slack][guard page]
  char* lds = nullptr;          // start of this launch's dynamic LDS: ends exactly at the guard page
  size_t lds_bytes = 0;
  ucontext_t main_uc;
  int n = 0, index = 0, live = 0;
  int b_arrived = 0;
  unsigned b_gen = 0;
  unsigned long progress = 0;
  const std::function<void()>* body = nullptr;
  std::unordered_map<uintptr_t, LdsSite> lds_sites;
  uint64_t traffic[6] = {0, 0, 0, 0, 0, 0};
  std::vector<uint64_t> buf_traffic;
  std::vector<int> ring_next;     // block ring: the fiber that runs after fiber i (waves in the order HIP_EMU_WAVE_ORDER asks for)
  int first = 0;
  const char* tls_lo = nullptr;   // this thread's TLS block of the emulated library: where the static __shared__ arrays live
  const char* tls_hi = nullptr;
  ~Worker() {
    if (stacks) munmap(stacks, n_stacks * kStack);
    if (lds_map) munmap(lds_map, (kLdsMax + page() - 1) / page() * page() + page());
  }
};

static thread_local Worker* wk = nullptr;

[[noreturn]] static void die(const char* what) {
  Lane* l = cur;
  std::fprintf(stderr, "hip_emu: %s (block %u,%u,%u thread %d)\n", what, l ? l->bid.x : 0, l ? l->bid.y : 0, l ? l->bid.z : 0,
               l ? l->linear : -1);
  std::abort();
}

static void switch_to_next() {
  Worker* w = wk;
  Fiber* me = &w->fibers[w->index];
  if (w->live == 0) {
    swapcontext(&me->uc, &w->main_uc);
    return;
  }
  int nxt = w->index;
  do {
    nxt = w->ring_next[nxt];
  } while (w->fibers[nxt].done);
  if (nxt == w->index) return;
  w->index = nxt;
  cur = &w->fibers[nxt].lane;
  swapcontext(&me->uc, &w->fibers[nxt].uc);
}

// Wait (yielding) until *gen moves past `g`; a full idle cycle of every other fiber without any state change is a deadlock:
// a barrier some lanes never reach.
static void wait_gen(const unsigned* gen, unsigned g) {
  Worker* w = wk;
  unsigned long seen = w->progress;
  int idle = 0;
  while (*const_cast<const volatile unsigned*>(gen) == g) {
    switch_to_next();
    if (w->progress == seen) {
      if (++idle > 4 * w->n + 8) die("deadlock: a barrier or wave-level operation is not reached by all lanes");
    } else {
      seen = w->progress;
      idle = 0;
    }
  }
}

void block_sync() {
  Worker* w = wk;
  ++w->progress;
  const unsigned g = w->b_gen;
  if (++w->b_arrived == w->live) {
    w->b_arrived = 0;
    ++w->b_gen;
  } else {
    wait_gen(&w->b_gen, g);
  }
}

void wave_sync() {
  Worker* w = wk;
  Wave* wv = cur->wave;
  ++w->progress;
  const unsigned g = wv->gen;
  if (++wv->arrived == wv->live) {
    wv->arrived = 0;
    ++wv->gen;
  } else {
    wait_gen(&wv->gen, g);
  }
}

// ---- lockstep within a wave ------------------------------------------------------------------------------------------
// The kernel sources are compiled with clang's -fsanitize=thread instrumentation and THIS file supplies the callbacks
// (the sanitizer's runtime is not linked): every load / store the kernel code performs arrives here first.  Before an
// access to LDS (the launch's dynamic LDS, or a static __shared__ array = a thread_local of the emulated library) the
// lane hands over to the next lane of its wave, so the 64 lanes of a wave take turns LDS access by LDS access: when a
// lane performs its k-th access, every lane of the wave has performed its first k-1 -- what the lockstep execution
// of a wavefront (plus the compiler's s_waitcnt) guarantees to code that hands data across lanes through LDS without
// a workgroup barrier.
static int tls_probe(struct dl_phdr_info* info, size_t, void* out) {
  const char* me = reinterpret_cast<const char*>(&cur);
  if (!info->dlpi_tls_data) return 0;
  for (int i = 0; i < info->dlpi_phnum; ++i) {
    if (info->dlpi_phdr[i].p_type != PT_TLS) continue;
    const char* lo = static_cast<const char*>(info->dlpi_tls_data);
    const char* hi = lo + info->dlpi_phdr[i].p_memsz;
    if (me >= lo && me < hi) {
      static_cast<const char**>(out)[0] = lo;
      static_cast<const char**>(out)[1] = hi;
      return 1;
    }
  }
  return 0;
}

// ---- traffic accounting (HIP_EMU_TRAFFIC=1): bytes the kernels request from / send to the guard-paged buffers (what
// hip_emu_alloc handed out: every tensor that went through HipOps), LDS bytes, matrix instructions.  Requests, before any
// cache: requested / algorithmic bytes > 1 is re-reading that only L2 / the Infinity Cache can absorb on the device.
static std::atomic<bool> g_traffic_on{std::getenv("HIP_EMU_TRAFFIC") != nullptr};      // or hip_emu_count_traffic(1)
static std::mutex g_buf_mu;
static std::map<uintptr_t, uintptr_t> g_buffers;                   // start -> end of every live guarded buffer
static std::atomic<uint64_t> g_traffic[6];                         // global read, global write, LDS read, LDS write, mfma 32x32x2, mfma 16x16x4
struct BufferSnapshot {
  std::vector<std::pair<uintptr_t, uintptr_t>> v;
};
static BufferSnapshot g_snapshot;                                  // taken at launch (buffers do not change during one)

static std::map<uintptr_t, std::pair<uint64_t, uint64_t>> g_buf_traffic;   // buffer start -> bytes read, written (per buffer:
                                                                            // a kernel's coefficient tables are wave-uniform scalar
                                                                            // loads on the device; the streams are what counts)
static inline int buffer_index(uintptr_t a) {
  const auto& v = g_snapshot.v;
  size_t lo = 0, hi = v.size();
  while (lo < hi) {
    const size_t mid = (lo + hi) / 2;
    if (v[mid].second <= a) lo = mid + 1; else hi = mid;
  }
  return lo < v.size() && v[lo].first <= a ? static_cast<int>(lo) : -1;
}

static const bool g_lds_stats = std::getenv("HIP_EMU_LDS_STATS") != nullptr;
static std::mutex g_lds_mu;
static std::map<uintptr_t, LdsSite> g_lds_sites;

static void lds_flush(Worker* w, LdsPending& p) {
  if (p.tag == ~0u || !p.mask) {
    p.tag = ~0u;
    p.mask = 0;
    return;
  }
  // lane groups of the instruction (MI355X_MICROARCH.md, LDS table)
  static const int g128[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                  {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                  {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                  {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  int ngroups, gsize, banks;
  const int dwords = p.size >= 4 ? p.size / 4 : 1;
  bool table128 = false;
  if (!p.write) {
    if (p.size <= 4) { ngroups = 2; gsize = 32; banks = 32; }
    else if (p.size == 8) { ngroups = 2; gsize = 32; banks = 64; }
    else { ngroups = 4; gsize = 16; banks = 64; table128 = true; }
  } else {
    banks = 32;
    if (p.size <= 4) { ngroups = 2; gsize = 32; }
    else if (p.size == 8) { ngroups = 4; gsize = 16; }
    else { ngroups = 8; gsize = 8; }
  }
  uint64_t ideal = 0, cycles = 0;
  for (int g = 0; g < ngroups; ++g) {
    uintptr_t seen[64][16];
    int nseen[64] = {0};
    bool any = false;
    for (int q = 0; q < gsize; ++q) {
      const int lane = table128 ? g128[g][q] : g * gsize + q;
      if (!((p.mask >> lane) & 1u)) continue;
      any = true;
      for (int d = 0; d < dwords; ++d) {
        const uintptr_t dw = p.addr[lane] / 4 + static_cast<uintptr_t>(d);
        const int b = static_cast<int>(dw % static_cast<uintptr_t>(banks));
        bool dup = false;
        for (int i = 0; i < nseen[b]; ++i) dup |= seen[b][i] == dw;
        if (!dup && nseen[b] < 16) seen[b][nseen[b]++] = dw;
      }
    }
    if (!any) continue;
    int worst = 1;
    for (int b = 0; b < banks; ++b) worst = std::max(worst, nseen[b]);
    ideal += 1;
    cycles += static_cast<uint64_t>(worst);
  }
  LdsSite& site = w->lds_sites[p.pc];
  site.insts += 1;
  site.ideal += ideal;
  site.cycles += cycles;
  site.size = p.size;
  site.write = p.write;
  p.tag = ~0u;
  p.mask = 0;
}

static void lds_flush_wave(Worker* w, Wave* wv) {
  for (auto& p : wv->pend) lds_flush(w, p);
}

static inline void on_access(const void* a, int size, bool write, uintptr_t pc) {
  if (!cur) return;                                   // host code of the library
  Worker* w = wk;
  const char* p = static_cast<const char*>(a);
  const bool dyn = p >= w->lds && p < w->lds + w->lds_bytes;
  const bool stat = p >= w->tls_lo && p < w->tls_hi && p != reinterpret_cast<const char*>(&cur);
  if (!dyn && !stat) {
    if (g_traffic_on) {
      const int bi = buffer_index(reinterpret_cast<uintptr_t>(a));
      if (bi >= 0) {
        w->traffic[write ? 1 : 0] += static_cast<uint64_t>(size);
        if (w->buf_traffic.size() < 2 * g_snapshot.v.size()) w->buf_traffic.resize(2 * g_snapshot.v.size(), 0);
        w->buf_traffic[2 * static_cast<size_t>(bi) + (write ? 1 : 0)] += static_cast<uint64_t>(size);
      }
    }
    return;
  }
  if (g_traffic_on) w->traffic[write ? 3 : 2] += static_cast<uint64_t>(size);
  const int me = w->index, base = me & ~63, end = std::min(base + 64, w->n);
  if (g_lds_stats) {
    Fiber& f = w->fibers[me];
    Wave* wv = f.lane.wave;
    const unsigned k = f.lds_k++;
    LdsPending& pd = wv->pend[k & 3u];
    if (pd.tag != k || pd.pc != pc) {
      lds_flush(w, pd);
      pd.tag = k;
      pd.pc = pc;
      pd.size = size;
      pd.write = write;
    }
    pd.addr[f.lane.lane] = reinterpret_cast<uintptr_t>(a);
    pd.mask |= 1ull << f.lane.lane;
    if (pd.mask == wv->live_mask) lds_flush(w, pd);
  }
  int nxt = me;
  do {
    nxt = nxt + 1 == end ? base : nxt + 1;
  } while (nxt != me && w->fibers[nxt].done);
  if (nxt == me) return;
  ++w->progress;
  w->index = nxt;
  cur = &w->fibers[nxt].lane;
  swapcontext(&w->fibers[me].uc, &w->fibers[nxt].uc);
}

uint64_t* wave_slot(unsigned parity) { return cur->wave->slot[parity & 1u]; }
void count_mfma(int which) {
  if (g_traffic_on && cur->lane == __builtin_ctzll(cur->wave->live_mask)) ++wk->traffic[4 + which];
}
uint64_t wave_live_mask() { return cur->wave->live_mask; }
void* dyn_lds() { return wk->lds; }

static Fiber* cur_fiber() { return &wk->fibers[wk->index]; }

static void dma_land(Fiber* f, size_t keep) {
  while (f->dma.size() > keep) {
    const Dma d = f->dma.front();
    std::memcpy(d.dst, d.src, static_cast<size_t>(d.bytes));
    f->dma.erase(f->dma.begin());
  }
}
void dma_request(const void* src, void* lds_dst, int bytes) {
  Worker* w = wk;
  if (static_cast<char*>(lds_dst) < w->lds || static_cast<char*>(lds_dst) + bytes > w->lds + w->lds_bytes) die("LDS-DMA outside the dynamic LDS");
  if (g_traffic_on) {                                  // a global read and an LDS write that no load / store instruction shows
    const int bi = buffer_index(reinterpret_cast<uintptr_t>(src));
    if (bi >= 0) {
      w->traffic[0] += static_cast<uint64_t>(bytes);
      if (w->buf_traffic.size() < 2 * g_snapshot.v.size()) w->buf_traffic.resize(2 * g_snapshot.v.size(), 0);
      w->buf_traffic[2 * static_cast<size_t>(bi)] += static_cast<uint64_t>(bytes);
    }
    w->traffic[3] += static_cast<uint64_t>(bytes);
  }
  cur_fiber()->dma.push_back({src, lds_dst, bytes});
}
void waitcnt_vm(int outstanding) { dma_land(cur_fiber(), static_cast<size_t>(outstanding < 0 ? 0 : outstanding)); }

static void fiber_entry() {
  Worker* w = wk;
  (*w->body)();
  // the fiber that resumes here may have been switched: re-read
  w = wk;
  Fiber* me = cur_fiber();
  dma_land(me, 0);
  me->done = true;
  --w->live;
  ++w->progress;
  Wave* wv = me->lane.wave;
  if (g_lds_stats) lds_flush_wave(w, wv);
  --wv->live;
  wv->live_mask &= ~(1ull << me->lane.lane);
  if (wv->live > 0 && wv->arrived == wv->live) {     // the lanes still waiting are now complete
    wv->arrived = 0;
    ++wv->gen;
  }
  if (w->live > 0 && w->b_arrived == w->live) {
    w->b_arrived = 0;
    ++w->b_gen;
  }
  switch_to_next();
  die("a finished fiber was resumed");
}

static void worker_setup(Worker* w, int n, size_t lds_bytes) {
  const size_t pg = page();
  if (static_cast<int>(w->fibers.size()) < n) w->fibers.resize(static_cast<size_t>(n));
  if (w->n_stacks < static_cast<size_t>(n)) {
    if (w->stacks) munmap(w->stacks, w->n_stacks * kStack);
    w->stacks = static_cast<char*>(mmap(nullptr, static_cast<size_t>(n) * kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    if (w->stacks == MAP_FAILED) die("mmap of the fiber stacks failed");
    w->n_stacks = static_cast<size_t>(n);
  }
  if (!w->lds_map) {
    const size_t body = (kLdsMax + pg - 1) / pg * pg;
    w->lds_map = static_cast<char*>(mmap(nullptr, body + pg, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    if (w->lds_map == MAP_FAILED) die("mmap of the LDS failed");
    mprotect(w->lds_map + body, pg, PROT_NONE);
  }
  if (!w->tls_lo) {
    const char* range[2] = {nullptr, nullptr};
    dl_iterate_phdr(tls_probe, range);
    w->tls_lo = range[0];
    w->tls_hi = range[1];
  }
  if (lds_bytes > kLdsMax) die("more than 160 KB of dynamic LDS requested");
  const size_t body = (kLdsMax + pg - 1) / pg * pg;
  const size_t rounded = (lds_bytes + 15) / 16 * 16;
  w->lds = w->lds_map + body - rounded;
  w->lds_bytes = rounded;
  w->n = n;
  w->waves.resize(static_cast<size_t>((n + 63) / 64));
}

static void run_block(Worker* w, dim3 grid, dim3 block, unsigned bx, unsigned by, unsigned bz) {
  const int n = w->n;
  // uninitialised LDS reads must not look like zeros
  uint32_t* l32 = reinterpret_cast<uint32_t*>(w->lds);
  for (size_t i = 0; i < w->lds_bytes / 4; ++i) l32[i] = std::getenv("HIP_EMU_LDS_ZERO") ? 0u : 0x7fc00badu;
  for (auto& wv : w->waves) {
    wv.live = wv.arrived = 0;
    wv.gen = 0;
    wv.live_mask = 0;
    for (auto& p : wv.pend) {
      p.tag = ~0u;
      p.mask = 0;
    }
  }
  w->live = n;
  w->b_arrived = 0;
  w->b_gen = 0;
  for (int t = 0; t < n; ++t) {
    Fiber& f = w->fibers[t];
    f.done = false;
    f.lds_k = 0;
    f.dma.clear();
    Lane& l = f.lane;
    l.tid = {static_cast<unsigned>(t % block.x), static_cast<unsigned>((t / block.x) % block.y), static_cast<unsigned>(t / (block.x * block.y))};
    l.bid = {bx, by, bz};
    l.bdim = block;
    l.gdim = grid;
    l.linear = t;
    l.lane = t & 63;
    l.parity = 0;
    l.wave = &w->waves[t >> 6];
    ++l.wave->live;
    l.wave->live_mask |= 1ull << l.lane;
    getcontext(&f.uc);
    f.uc.uc_stack.ss_sp = w->stacks + static_cast<size_t>(t) * kStack;
    f.uc.uc_stack.ss_size = kStack;
    f.uc.uc_link = nullptr;
    makecontext(&f.uc, fiber_entry, 0);
  }
  // The order in which the waves of the workgroup get their turns.  Waves only interleave at rendezvous, so a missing
  // __syncthreads() between a producer and a consumer wave stays invisible while the producer happens to run first:
  // HIP_EMU_WAVE_ORDER=reverse runs the last wave first, =random shuffles the waves per workgroup (seeded by the block).
  const int nwaves = (n + 63) / 64;
  std::vector<int> order(static_cast<size_t>(nwaves));
  for (int i = 0; i < nwaves; ++i) order[i] = i;
  const char* mode = std::getenv("HIP_EMU_WAVE_ORDER");
  if (mode && mode[0] == 'r' && mode[1] == 'e') {
    for (int i = 0; i < nwaves; ++i) order[i] = nwaves - 1 - i;
  } else if (mode && mode[0] == 'r' && mode[1] == 'a') {
    uint64_t st = 0x9E3779B97F4A7C15ull * (1 + bx + 131ull * by + 17161ull * bz) + static_cast<uint64_t>(std::atoll(mode + 6));
    for (int i = nwaves - 1; i > 0; --i) {
      st = st * 6364136223846793005ull + 1442695040888963407ull;
      std::swap(order[i], order[static_cast<int>((st >> 33) % static_cast<uint64_t>(i + 1))]);
    }
  }
  w->ring_next.assign(static_cast<size_t>(n), 0);
  int prev = -1;
  w->first = -1;
  for (int q = 0; q < nwaves; ++q)
    for (int t = order[q] * 64; t < std::min(n, order[q] * 64 + 64); ++t) {
      if (prev >= 0) w->ring_next[prev] = t; else w->first = t;
      prev = t;
    }
  w->ring_next[prev] = w->first;
  w->index = w->first;
  cur = &w->fibers[w->first].lane;
  swapcontext(&w->main_uc, &w->fibers[w->first].uc);
  cur = nullptr;
}

static int n_workers() {
  const char* e = std::getenv("HIP_EMU_WORKERS");
  int n = e ? std::atoi(e) : static_cast<int>(std::thread::hardware_concurrency());
  if (n < 1) n = 1;
  if (n > 16) n = 16;
  return n;
}

// launches per kernel (address of the __global__ function's instantiation): which template variants the tests reach
static std::mutex g_launch_mu;
static std::map<const void*, uint64_t> g_launches;

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body, const void* kernel) {
  const int n = static_cast<int>(block.x * block.y * block.z);
  const long total = static_cast<long>(grid.x) * grid.y * grid.z;
  if (kernel) {
    std::lock_guard<std::mutex> lock(g_launch_mu);
    ++g_launches[kernel];
  }
  if (n <= 0 || n > 1024 || total <= 0) return;
  if (g_traffic_on) {
    std::lock_guard<std::mutex> lock(g_buf_mu);
    g_snapshot.v.assign(g_buffers.begin(), g_buffers.end());
  }
  std::atomic<long> next{0};
  auto work = [&]() {
    static thread_local Worker worker;
    wk = &worker;
    worker.body = &body;
    worker_setup(&worker, n, lds_bytes);
    for (;;) {
      const long b = next.fetch_add(1);
      if (b >= total) break;
      run_block(&worker, grid, block, static_cast<unsigned>(b % grid.x), static_cast<unsigned>((b / grid.x) % grid.y),
                static_cast<unsigned>(b / (static_cast<long>(grid.x) * grid.y)));
    }
    if (g_traffic_on) {
      for (int i = 0; i < 6; ++i) {
        g_traffic[i] += worker.traffic[i];
        worker.traffic[i] = 0;
      }
      std::lock_guard<std::mutex> lock(g_buf_mu);
      for (size_t i = 0; 2 * i + 1 < worker.buf_traffic.size() && i < g_snapshot.v.size(); ++i) {
        auto& t = g_buf_traffic[g_snapshot.v[i].first];
        t.first += worker.buf_traffic[2 * i];
        t.second += worker.buf_traffic[2 * i + 1];
      }
      worker.buf_traffic.assign(worker.buf_traffic.size(), 0);
    }
    if (g_lds_stats && !worker.lds_sites.empty()) {
      std::lock_guard<std::mutex> lock(g_lds_mu);
      for (const auto& kv : worker.lds_sites) {
        LdsSite& t = g_lds_sites[kv.first];
        t.insts += kv.second.insts;
        t.ideal += kv.second.ideal;
        t.cycles += kv.second.cycles;
        t.size = kv.second.size;
        t.write = kv.second.write;
      }
      worker.lds_sites.clear();
    }
  };
  // watchdog: a launch that does not finish within HIP_EMU_LAUNCH_TIMEOUT seconds (default 900) is a hang the rendezvous
  // bookkeeping did not recognise as a deadlock -- say where the workers are and abort instead of stalling the test run
  std::mutex done_mu;
  std::condition_variable done_cv;
  bool done = false;
  const char* lim = std::getenv("HIP_EMU_LAUNCH_TIMEOUT");
  const long limit_s = lim ? std::atol(lim) : 900;
  std::thread watchdog([&]() {
    std::unique_lock<std::mutex> lock(done_mu);
    if (!done_cv.wait_for(lock, std::chrono::seconds(limit_s), [&] { return done; })) {
      Dl_info info;
      const char* name = kernel && dladdr(kernel, &info) && info.dli_sname ? info.dli_sname : "?";
      std::fprintf(stderr, "hip_emu: launch of %s (grid %u x %u x %u, block %d, %zu B LDS) still running after %ld s: block %ld of %ld "
                   "handed out\n", name, grid.x, grid.y, grid.z, n, lds_bytes, limit_s, next.load(), total);
      std::abort();
    }
  });
  const int nw = static_cast<int>(std::min<long>(n_workers(), total));
  {
    std::vector<std::thread> ts;                                     // never on the caller's (Python's) stack and TLS
    for (int i = 0; i < std::max(nw, 1); ++i) ts.emplace_back(work);
    for (auto& t : ts) t.join();
  }
  {
    std::lock_guard<std::mutex> lock(done_mu);
    done = true;
  }
  done_cv.notify_one();
  watchdog.join();
}

}  // namespace hip_emu

// ---- -fsanitize=thread callbacks (see on_access) -----------------------------------------------------------------------
extern "C" {
void __tsan_init() {}
#define HIP_EMU_PC reinterpret_cast<uintptr_t>(__builtin_return_address(0))
// The compiler calls the plain hooks where the access's type promises n-byte alignment (a float4 / f32x2 load is ONE
// b128 / b64 instruction on the device and the LDS forms of those need the alignment): a kernel that breaks the promise
// stops here instead of working by accident on the host.
static inline void hip_emu_aligned(void* a, int n, bool write) {
  if (reinterpret_cast<uintptr_t>(a) % static_cast<uintptr_t>(n) != 0) {
    std::fprintf(stderr, "hip_emu: misaligned %d-byte %s at %p in a kernel\n", n, write ? "store" : "load", a);
    std::abort();
  }
}
#define HIP_EMU_HOOK(n)                                                                              \
  void __tsan_read##n(void* a) { hip_emu_aligned(a, n, false); hip_emu::on_access(a, n, false, HIP_EMU_PC); }   \
  void __tsan_write##n(void* a) { hip_emu_aligned(a, n, true); hip_emu::on_access(a, n, true, HIP_EMU_PC); }    \
  void __tsan_unaligned_read##n(void* a) { hip_emu::on_access(a, n, false, HIP_EMU_PC); }            \
  void __tsan_unaligned_write##n(void* a) { hip_emu::on_access(a, n, true, HIP_EMU_PC); }
HIP_EMU_HOOK(1)
HIP_EMU_HOOK(2)
HIP_EMU_HOOK(4)
HIP_EMU_HOOK(8)
HIP_EMU_HOOK(16)
void __tsan_read_range(void* a, unsigned long n) { hip_emu::on_access(a, static_cast<int>(n), false, HIP_EMU_PC); }
void __tsan_write_range(void* a, unsigned long n) { hip_emu::on_access(a, static_cast<int>(n), true, HIP_EMU_PC); }
void __tsan_vptr_update(void**, void*) {}
void __tsan_vptr_read(void**) {}
void __tsan_func_entry(void*) {}
void __tsan_func_exit() {}
}

// ---- the HIP runtime entry points the host side of the library uses ------------------------------------------------
extern "C" {

hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipFuncGetAttributes(hipFuncAttributes* attr, const void*) {
  std::memset(attr, 0, sizeof(*attr));
  attr->maxThreadsPerBlock = 1024;
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) {
  *d = 0;
  return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
  *v = a == hipDeviceAttributeMultiprocessorCount ? 256 : 0;
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  std::memset(p, v, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
  std::memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) {
  *n = 2;
  return hipSuccess;
}

// Buffers with a guard page before and after: a read or write past either end faults at once.  The payload is placed so
// that it ENDS at the trailing guard page (rounded to 16 bytes, the alignment every kernel may assume).
// LDS statistics gathered so far (HIP_EMU_LDS_STATS=1): one line per code address,
// "<offset in this library> <bytes> <r|w> <wave-instructions> <conflict-free cycles> <cycles>"; clears the table.
int hip_emu_lds_report(const char* path) {
  std::lock_guard<std::mutex> lock(hip_emu::g_lds_mu);
  FILE* f = std::fopen(path, "w");
  if (!f) return -1;
  Dl_info info;
  uintptr_t base = 0;
  if (dladdr(reinterpret_cast<void*>(&hip_emu_lds_report), &info)) base = reinterpret_cast<uintptr_t>(info.dli_fbase);
  for (const auto& kv : hip_emu::g_lds_sites)
    std::fprintf(f, "0x%lx %d %c %lu %lu %lu\n", static_cast<unsigned long>(kv.first - base - 1), kv.second.size,
                 kv.second.write ? 'w' : 'r', static_cast<unsigned long>(kv.second.insts), static_cast<unsigned long>(kv.second.ideal),
                 static_cast<unsigned long>(kv.second.cycles));
  std::fclose(f);
  hip_emu::g_lds_sites.clear();
  return 0;
}

// One line per kernel launched so far: "<symbol> <launches>" (the mangled name of the instantiation).
int hip_emu_launch_report(const char* path) {
  std::lock_guard<std::mutex> lock(hip_emu::g_launch_mu);
  FILE* f = std::fopen(path, "w");
  if (!f) return -1;
  for (const auto& kv : hip_emu::g_launches) {
    Dl_info info;
    const char* name = dladdr(kv.first, &info) && info.dli_sname ? info.dli_sname : "?";
    std::fprintf(f, "%s %lu\n", name, static_cast<unsigned long>(kv.second));
  }
  std::fclose(f);
  return 0;
}

void* hip_emu_alloc(size_t bytes) {
  const size_t pg = hip_emu::page();
  const size_t rounded = (bytes + 15) / 16 * 16;
  const size_t body = (rounded + pg - 1) / pg * pg;
  char* m = static_cast<char*>(mmap(nullptr, body + 2 * pg, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
  if (m == MAP_FAILED) return nullptr;
  mprotect(m, pg, PROT_NONE);
  mprotect(m + pg + body, pg, PROT_NONE);
  char* p = m + pg + body - rounded;
  std::memset(m + pg, 0xCB, body);
  {
    std::lock_guard<std::mutex> lock(hip_emu::g_buf_mu);
    hip_emu::g_buffers[reinterpret_cast<uintptr_t>(p)] = reinterpret_cast<uintptr_t>(p) + bytes;
  }
  return p;
}
// out[6] = bytes read from / written to guarded buffers, LDS bytes read / written, 32x32x2 and 16x16x4 matrix instructions
// since the last call (counted while HIP_EMU_TRAFFIC is set or after hip_emu_count_traffic(1))
void hip_emu_count_traffic(int on) { hip_emu::g_traffic_on = on != 0; }
// out[2] = bytes the kernels read from / wrote to the guarded buffer that starts at p, since it was allocated
void hip_emu_buffer_traffic(void* p, uint64_t* out) {
  std::lock_guard<std::mutex> lock(hip_emu::g_buf_mu);
  const auto it = hip_emu::g_buf_traffic.find(reinterpret_cast<uintptr_t>(p));
  out[0] = it == hip_emu::g_buf_traffic.end() ? 0 : it->second.first;
  out[1] = it == hip_emu::g_buf_traffic.end() ? 0 : it->second.second;
}
void hip_emu_traffic(uint64_t* out) {
  for (int i = 0; i < 6; ++i) out[i] = hip_emu::g_traffic[i].exchange(0);
}
void hip_emu_free(void* p, size_t bytes) {
  {
    std::lock_guard<std::mutex> lock(hip_emu::g_buf_mu);
    hip_emu::g_buffers.erase(reinterpret_cast<uintptr_t>(p));
    hip_emu::g_buf_traffic.erase(reinterpret_cast<uintptr_t>(p));
  }
  const size_t pg = hip_emu::page();
  const size_t rounded = (bytes + 15) / 16 * 16;
  const size_t body = (rounded + pg - 1) / pg * pg;
  char* m = static_cast<char*>(p) + rounded - body - pg;
  munmap(m, body + 2 * pg);
}
}
