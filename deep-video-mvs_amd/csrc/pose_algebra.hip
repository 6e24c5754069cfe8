// Opt-in "exact" pose algebra on the device (fp64, rounded once), gfx950.
//
// By default the Python surface computes the sweep constants and the two relative poses with the reference's own fp32 torch
// expressions (/root/reference/dvmvs/utils.py:51-56, :121; dvmvs/convlstm.py:30) and hands them to the kernels as device arrays
// (include/dvmvs_hip.h, "Small pose algebra").  These two launches are the alternative for callers that want the matrices
// closer to the real-number result than fp32 LAPACK gets them, or no host involvement at all.
#include "plane_sweep.h"

namespace dvmvs {

struct SweepMatrixArgs {
  const float* pose1;
  const float* pose2[DVMVS_MAX_MEASUREMENTS];
  const float* K;
  float* Hm;
  float* kt;
  int B, M;
};

__global__ void sweep_matrices_kernel(SweepMatrixArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (b, m)
  if (i >= a.B * a.M) return;
  const int b = i / a.M, m = i - b * a.M;
  float Hm[9], kt[3];
  sweep_matrices(a.pose1 + b * 16, a.pose2[m] + b * 16, a.K + b * 9, Hm, kt);
#pragma unroll
  for (int k = 0; k < 9; ++k) a.Hm[i * 9 + k] = Hm[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) a.kt[i * 3 + k] = kt[k];
}

__global__ void relative_pose_kernel(const float* __restrict__ a, const float* __restrict__ c, float* __restrict__ out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double r[16];
  relative_pose_f64(a + b * 16, c + b * 16, r);
#pragma unroll
  for (int i = 0; i < 16; ++i) out[b * 16 + i] = static_cast<float>(r[i]);
}

}  // namespace dvmvs

extern "C" int dvmvs_sweep_matrices(const float* pose1, const float* const* pose2s, const float* K, float* Hm, float* kt,
                                    int B, int M, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!pose1 || !pose2s || !K || !Hm || !kt || B <= 0 || M <= 0) return DVMVS_EINVAL;
  if (M > DVMVS_MAX_MEASUREMENTS) return DVMVS_EUNSUPPORTED;
  SweepMatrixArgs a;
  a.pose1 = pose1; a.K = K; a.Hm = Hm; a.kt = kt; a.B = B; a.M = M;
  for (int m = 0; m < DVMVS_MAX_MEASUREMENTS; ++m) {
    if (m < M && !pose2s[m]) return DVMVS_EINVAL;
    a.pose2[m] = m < M ? pose2s[m] : nullptr;
  }
  hipLaunchKernelGGL(sweep_matrices_kernel, dim3((B * M + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), a);
  return launch_status();
}

extern "C" int dvmvs_relative_pose(const float* a, const float* c, float* out, int B, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!a || !c || !out || B <= 0) return DVMVS_EINVAL;
  hipLaunchKernelGGL(relative_pose_kernel, dim3((B + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), a, c, out, B);
  return launch_status();
}
