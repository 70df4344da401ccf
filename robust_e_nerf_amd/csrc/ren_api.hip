// Library identification (C ABI, include/ren_amd.h).
#include "ren_common.h"
extern "C" int ren_abi_version(void) { return 17; }
extern "C" const char *ren_build_info(void) { return "ren_amd gfx950 (CDNA4) hipcc " __VERSION__; }
