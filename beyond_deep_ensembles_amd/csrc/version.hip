#include "bde_common.hpp"
extern "C" int bde_version(void) { return 201; /* 0.2.1: + bde_lrt_linear_bwd */ }
extern "C" const char* bde_arch(void) { return "gfx950"; }
