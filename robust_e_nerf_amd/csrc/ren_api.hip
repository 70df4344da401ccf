// Library identification (C ABI, include/ren_amd.h).
#include "ren_common.h"
extern "C" int ren_abi_version(void) { return 25; }
extern "C" const char *ren_build_info(void) { return "ren_amd gfx950 (CDNA4) hipcc " __VERSION__; }

extern "C" int ren_set_knob(int32_t knob, int32_t value) {
    if (knob < 0 || knob >= REN_KNOB_COUNT) return REN_ERR_BAD_ARG;
    ren_knob_store()[knob].store(value, std::memory_order_relaxed);
    return REN_OK;
}
extern "C" int ren_get_knob(int32_t knob) { return knob < 0 || knob >= REN_KNOB_COUNT ? REN_ERR_BAD_ARG : ren_knob(knob); }
