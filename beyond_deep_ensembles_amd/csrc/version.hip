#include "bde_common.hpp"
extern "C" int bde_version(void) { return 200; /* 0.2.0 */ }
extern "C" const char* bde_arch(void) { return "gfx950"; }
