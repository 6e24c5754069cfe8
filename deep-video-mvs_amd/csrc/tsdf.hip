// TSDF fusion of one RGB-D frame into a voxel volume (gfx950): the `integrate` kernel of the reference's reconstruction
// script, /root/reference/sample-data/run-tsdf-reconstruction.py:79-152 (a CUDA source string compiled with pycuda there).
//
// HBM-bound streaming update: per visible voxel 3 floats read + 3 written (tsdf, weight, folded colour) plus two image
// gathers; the voxel index has z fastest, so consecutive lanes walk consecutive z (coalesced 256-byte wave accesses).
// The reference launches a 3-D grid several times ("gpu loops") with float-encoded scalars; here one grid-stride launch
// with 64-bit voxel indices covers any volume, scalars travel by value, and intrinsics / pose are read through scalar loads.
// Arithmetic order is the reference's, statement by statement, with floating-point contraction off so that the CPU
// restatement (oracle/tsdf_oracle.py) can be compared without tolerance games.
#include "dvmvs_device.h"

namespace dvmvs {

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void tsdf_integrate_kernel(float* __restrict__ tsdf_vol, float* __restrict__ weight_vol,
                                                              float* __restrict__ color_vol, int dim_x, int dim_y, int dim_z,
                                                              float origin_x, float origin_y, float origin_z, float voxel_size,
                                                              const float* __restrict__ cam_intr, const float* __restrict__ cam_pose,
                                                              const float* __restrict__ color_im, const float* __restrict__ depth_im,
                                                              int im_h, int im_w, float trunc_margin, float obs_weight) {
  const long long total = static_cast<long long>(dim_x) * dim_y * dim_z;
  const long long plane = static_cast<long long>(dim_y) * dim_z;
  const float fx = cam_intr[0], cx = cam_intr[2], fy = cam_intr[4], cy = cam_intr[5];
  float R[9], t[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) R[r * 3 + c] = cam_pose[r * 4 + c];
    t[r] = cam_pose[r * 4 + 3];
  }
  for (long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int vx = static_cast<int>(idx / plane);
    const int rem = static_cast<int>(idx - vx * plane);
    const int vy = rem / dim_z, vz = rem - vy * dim_z;
    // voxel grid -> world -> camera (rotation transposed: cam_pose is camera-to-world)
    const float tx = (origin_x + static_cast<float>(vx) * voxel_size) - t[0];
    const float ty = (origin_y + static_cast<float>(vy) * voxel_size) - t[1];
    const float tz = (origin_z + static_cast<float>(vz) * voxel_size) - t[2];
    const float cam_x = R[0] * tx + R[3] * ty + R[6] * tz;
    const float cam_y = R[1] * tx + R[4] * ty + R[7] * tz;
    const float cam_z = R[2] * tx + R[5] * ty + R[8] * tz;
    const float u = roundf(fx * (cam_x / cam_z) + cx), v = roundf(fy * (cam_y / cam_z) + cy);
    // outside the view frustum (NaN / Inf from cam_z == 0 fail the comparisons the same way), behind the camera
    if (!(u >= 0.0f && u < static_cast<float>(im_w) && v >= 0.0f && v < static_cast<float>(im_h)) || cam_z < 0.0f) continue;
    const int pixel = static_cast<int>(v) * im_w + static_cast<int>(u);
    const float depth_value = depth_im[pixel];
    if (depth_value == 0.0f) continue;
    const float depth_diff = depth_value - cam_z;
    if (depth_diff < -trunc_margin) continue;
    const float dist = fminf(1.0f, depth_diff / trunc_margin);
    const float w_old = weight_vol[idx];
    const float w_new = w_old + obs_weight;
    weight_vol[idx] = w_new;
    tsdf_vol[idx] = (tsdf_vol[idx] * w_old + obs_weight * dist) / w_new;
    // colour: b * 65536 + g * 256 + r in one float
    const float old_color = color_vol[idx];
    const float old_b = floorf(old_color / 65536.0f);
    const float old_g = floorf((old_color - old_b * 65536.0f) / 256.0f);
    const float old_r = old_color - old_b * 65536.0f - old_g * 256.0f;
    const float new_color = color_im[pixel];
    float new_b = floorf(new_color / 65536.0f);
    float new_g = floorf((new_color - new_b * 65536.0f) / 256.0f);
    float new_r = new_color - new_b * 65536.0f - new_g * 256.0f;
    new_b = fminf(roundf((old_b * w_old + obs_weight * new_b) / w_new), 255.0f);
    new_g = fminf(roundf((old_g * w_old + obs_weight * new_g) / w_new), 255.0f);
    new_r = fminf(roundf((old_r * w_old + obs_weight * new_r) / w_new), 255.0f);
    color_vol[idx] = new_b * 65536.0f + new_g * 256.0f + new_r;
  }
}
#pragma clang fp contract(fast)

}  // namespace dvmvs

extern "C" int dvmvs_tsdf_integrate(float* tsdf_vol, float* weight_vol, float* color_vol, int dim_x, int dim_y, int dim_z,
                                    float origin_x, float origin_y, float origin_z, float voxel_size, const float* cam_intr,
                                    const float* cam_pose, const float* color_im, const float* depth_im, int im_h, int im_w,
                                    float trunc_margin, float obs_weight, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!tsdf_vol || !weight_vol || !color_vol || !cam_intr || !cam_pose || !color_im || !depth_im) return DVMVS_EINVAL;
  if (dim_x <= 0 || dim_y <= 0 || dim_z <= 0 || im_h <= 0 || im_w <= 0) return DVMVS_EINVAL;
  if (!(voxel_size > 0.0f) || !(trunc_margin > 0.0f)) return DVMVS_EINVAL;
  if (static_cast<long long>(dim_y) * dim_z >= (1LL << 31) || static_cast<long long>(im_h) * im_w >= (1LL << 31)) return DVMVS_EUNSUPPORTED;
  const long long total = static_cast<long long>(dim_x) * dim_y * dim_z;
  // enough workgroups to fill 256 CUs several times over, grid-stride beyond that
  const long long wanted = (total + 255) / 256;
  const unsigned int grid = static_cast<unsigned int>(wanted < 16384 ? wanted : 16384);
  hipLaunchKernelGGL(tsdf_integrate_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), tsdf_vol, weight_vol, color_vol,
                     dim_x, dim_y, dim_z, origin_x, origin_y, origin_z, voxel_size, cam_intr, cam_pose, color_im, depth_im, im_h, im_w,
                     trunc_margin, obs_weight);
  return launch_status();
}
