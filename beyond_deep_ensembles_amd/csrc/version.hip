#include "bde_common.hpp"
extern "C" int bde_version(void) { return 100; /* 0.1.0 */ }
extern "C" const char* bde_arch(void) { return "gfx950"; }
