// ABI identification and error strings of libdvmvs_hip.so.
#include "dvmvs_device.h"

extern "C" int dvmvs_abi_version(void) { return DVMVS_ABI_VERSION; }

extern "C" const char* dvmvs_build_arch(void) { return "gfx950"; }

extern "C" const char* dvmvs_error_string(int code) {
  if (code == 0) return "success";
  if (code == DVMVS_EINVAL) return "dvmvs: invalid argument (null pointer, non-positive dimension or bad count)";
  if (code == DVMVS_EUNSUPPORTED) return "dvmvs: shape or mode not supported by the gfx950 kernels";
  if (code == DVMVS_ELIBRARY) return "dvmvs: a MIOpen call failed";
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "dvmvs: unknown error";
}
