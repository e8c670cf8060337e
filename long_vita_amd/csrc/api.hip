// ABI bookkeeping for libvita_hip.so.
#include "vita_common.h"

extern "C" int vita_abi_version(void) { return 18; }

extern "C" const char* vita_error_string(int code) {
  switch (code) {
    case VITA_OK: return "ok";
    case VITA_ERR_INVALID_ARG: return "invalid argument";
    case VITA_ERR_UNSUPPORTED: return "shape or layout not supported by the gfx950 kernels";
    case VITA_ERR_LAUNCH: return "HIP kernel launch failed";
    default: return "unknown error";
  }
}
