#include "bde_common.hpp"
extern "C" int bde_version(void) { return 406; /* 0.4.5: bde_svgd_step streams at every size (the small-model kernel is the caller's explicit choice), bde_mean_scalars multiplies by fl(1/divisor) like torch's GPU division, bde_conv_lrt_bwd_weight takes ws_bytes (0.4.4: + bde_conv_lrt_gvar_bias (g_var and the bias gradients in one pass) (0.4.3: + conv tuning hooks (bde_conv_lrt_pass_geos / _candidates / _set_tiling, _wgrad_candidates / _wgrad_set_tiling) (0.4.2: + bde_sum_scalars; conv weight gradient: one instantiation per column-tile count, 512-slot shares, tree reduction (0.4.1: + bde_conv_lrt_prep_strided / bde_conv_lrt_bwd_data_phases (0.4.0: contiguous SWAG rows again, LDS-DMA batched sampler, bde_conv_lrt_*, single-launch protocol gone))))) */ }
extern "C" const char* bde_arch(void) { return "gfx950"; }

extern "C" {
int bde_internal_load_conv_lrt(void);
int bde_internal_load_conv_lrt_bwd(void);
int bde_internal_load_gauss(void);
int bde_internal_load_ivon(void);
int bde_internal_load_lrt(void);
int bde_internal_load_lrt_bwd(void);
int bde_internal_load_svgd(void);
int bde_internal_load_svgd_fused(void);
int bde_internal_load_svgd_small(void);
int bde_internal_load_swag(void);
int bde_internal_load_swag_batched(void);
}

// Load every code object of the library on the CURRENT device.  HIP defers the upload of a code object to the first
// launch of one of its kernels; a process that is about to start communication threads (torch.distributed) or to share
// the device with other processes calls this first, from one thread, so that no kernel's first launch coincides with
// them (profiles/r03_first_launch_*.txt).  Idempotent, cheap after the first call; needs a visible device.
//
// The translation units whose kernels have been green on an MI355X must load: their failure is bde_init()'s return code.
// The ones that have NOT (the small-model SVGD kernel, the fused convolution kernels -- no default call launches them,
// device_verified.py / conv_profit.py) are uploaded as well, but a failure there is only RECORDED
// (bde_init_optional_failures): one bad code object among the never-run kernels must not take every verified kernel, the
// tests, smoke() and the bench down at construction.
#include <atomic>
static std::atomic<unsigned> g_optional_failures{0};

extern "C" int bde_init(void) {
  int (*const required[])(void) = {bde_internal_load_gauss, bde_internal_load_ivon,       bde_internal_load_lrt,  bde_internal_load_lrt_bwd,
                                   bde_internal_load_svgd,  bde_internal_load_svgd_fused, bde_internal_load_swag, bde_internal_load_swag_batched};
  for (auto load : required) {
    const int rc = load();
    if (rc) return rc;
  }
  int (*const optional[])(void) = {bde_internal_load_svgd_small, bde_internal_load_conv_lrt, bde_internal_load_conv_lrt_bwd};
  unsigned failed = 0;
  for (unsigned i = 0; i < 3; ++i)
    if (optional[i]() != 0) failed |= 1u << i;
  g_optional_failures.store(failed, std::memory_order_relaxed);
  return 0;
}

// Bit mask of the device-unverified translation units whose code object did not load in the last bde_init():
// 1 svgd_small.hip, 2 conv_lrt.hip, 4 conv_lrt_bwd.hip; 0 = everything is resident.
extern "C" int bde_init_optional_failures(void) { return static_cast<int>(g_optional_failures.load(std::memory_order_relaxed)); }
