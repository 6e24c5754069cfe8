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

// A kernel that does nothing, with a name nothing else in a process has: bench.py --mark-region launches it right before and right
// after its timed loop so that tools/summarize_trace.py can cut the loop out of a rocprofv3 kernel trace.
namespace dvmvs {
__global__ void trace_marker_kernel() {}
}  // namespace dvmvs

extern "C" int dvmvs_trace_marker(dvmvs_stream_t stream) {
  hipLaunchKernelGGL(dvmvs::trace_marker_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream));
  return dvmvs::launch_status();
}
