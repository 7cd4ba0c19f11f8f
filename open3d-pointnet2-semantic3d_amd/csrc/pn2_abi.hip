// pn2_abi.hip -- version / error-string entry points of libpn2_hip.so.
#include "pn2_common.h"

extern "C" int pn2_abi_version(void) { return PN2_ABI_VERSION; }

extern "C" const char* pn2_build_info(void) {
    return "libpn2_hip gfx950 (CDNA4, wave64) -ffp-contract=off; fp32 MFMA 32x32x2; built " __DATE__;
}

extern "C" const char* pn2_strerror(int code) {
    switch (code) {
        case PN2_OK: return "ok";
        case PN2_EINVAL: return "PN2_EINVAL: non-positive dimension or bad attribute";
        case PN2_ENULL: return "PN2_ENULL: required pointer is NULL";
        case PN2_ERANGE: return "PN2_ERANGE: dimension exceeds kernel limits";
        case PN2_EUNSUP: return "PN2_EUNSUP: unsupported configuration";
        default: break;
    }
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "unknown pn2 error";
}
