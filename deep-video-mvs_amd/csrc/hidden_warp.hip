// Depth-conditioned inverse warp of the ConvLSTM hidden state (forward + gradient w.r.t. the source), gfx950.
//
// Per destination pixel: un-project with the destination depth, move into the source camera, clamp z at 0,
// project, bilinear-sample the source map (zeros padding).  The caller's "zero where depth <= 0.01" mask is
// fused into the forward.  Semantics: /root/reference/dvmvs/utils.py:205-258 + dvmvs/convlstm.py:29-41; the
// pin-hole helpers are kornia 0.3.2's (restated in oracle/dvmvs_oracle.py).
//
// The maps are tiny (512 x 8 x 10): the kernel is latency bound, so one thread per output element, geometry
// recomputed per thread (about 60 flops) rather than staged through LDS with a barrier.
#include "dvmvs_device.h"

namespace dvmvs {

constexpr float kHomogeneousEps = 1e-8f;

// Source-image sample position for destination pixel (x, y).  The op order mirrors the torch expression graph
// (separately rounded multiply / add), hence contraction is switched off inside.
__device__ inline void hidden_warp_position(const float* T, const float* Km, float depth, int x, int y, int W, int H,
                                            float* ix, float* iy) {
#pragma clang fp contract(off)
  const float fx = Km[0], cx = Km[2], fy = Km[4], cy = Km[5];
  // kornia.depth_to_3d: ((u - cx) / fx, (v - cy) / fy, 1) * depth
  const float px = ((static_cast<float>(x) - cx) / fx) * depth;
  const float py = ((static_cast<float>(y) - cy) / fy) * depth;
  const float pz = depth;
  // kornia.transform_points: homogeneous 4x4 multiply (k-ordered accumulation), then de-homogenise
  float q[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) q[r] = ((T[r * 4 + 0] * px + T[r * 4 + 1] * py) + T[r * 4 + 2] * pz) + T[r * 4 + 3];
  const float sw = fabsf(q[3]) > kHomogeneousEps ? 1.0f / q[3] : 1.0f;
  const float X = sw * q[0], Y = sw * q[1];
  const float Z = fmaxf(sw * q[2], 0.0f);  // relu on z (utils.py:248)
  // kornia.project_points: de-homogenise (|z| > eps ? 1/z : 1), then fx * x + cx
  const float sz = fabsf(Z) > kHomogeneousEps ? 1.0f / Z : 1.0f;
  const float u = (X * sz) * fx + cx;
  const float v = (Y * sz) * fy + cy;
  // kornia.normalize_pixel_coordinates: 2 / max(size - 1, eps) * p - 1, then grid_sample's align_corners map
  const float nx = 2.0f / fmaxf(static_cast<float>(W - 1), kHomogeneousEps);
  const float ny = 2.0f / fmaxf(static_cast<float>(H - 1), kHomogeneousEps);
  *ix = unnormalize_ac(nx * u - 1.0f, W);
  *iy = unnormalize_ac(ny * v - 1.0f, H);
}

__global__ __launch_bounds__(256) void hidden_warp_fwd_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                                              const float* __restrict__ T, const float* __restrict__ Km,
                                                              float* __restrict__ out, int B, int C, int H, int W,
                                                              int zero_invalid) {
  const int HW = H * W;
  const long long total = static_cast<long long>(B) * C * HW;
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int pix = static_cast<int>(i % HW);
    const long long bc = i / HW;
    const int b = static_cast<int>(bc / C);
    const int y = pix / W, x = pix - y * W;
    const float d = depth[static_cast<long long>(b) * HW + pix];
    float ix, iy;
    hidden_warp_position(T + b * 16, Km + b * 9, d, x, y, W, H, &ix, &iy);
    const BilinearTaps t = make_taps(ix, iy, W, H);
    const float* plane = src + bc * HW;
    float s = 0.0f;
    if (t.in_x0 && t.in_y0) s += plane[t.y0 * W + t.x0] * t.w_nw;
    if (t.in_x1 && t.in_y0) s += plane[t.y0 * W + t.x0 + 1] * t.w_ne;
    if (t.in_x0 && t.in_y1) s += plane[(t.y0 + 1) * W + t.x0] * t.w_sw;
    if (t.in_x1 && t.in_y1) s += plane[(t.y0 + 1) * W + t.x0 + 1] * t.w_se;
    if (zero_invalid && d <= 0.01f) s = 0.0f;
    out[i] = s;
  }
}

// Gradient w.r.t. the source map: scatter of grad_out through the same taps (no mask, see header).
__global__ __launch_bounds__(256) void hidden_warp_bwd_kernel(const float* __restrict__ grad_out, const float* __restrict__ depth,
                                                              const float* __restrict__ T, const float* __restrict__ Km,
                                                              float* __restrict__ grad_src, int B, int C, int H, int W) {
  const int HW = H * W;
  const long long total = static_cast<long long>(B) * C * HW;
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int pix = static_cast<int>(i % HW);
    const long long bc = i / HW;
    const int b = static_cast<int>(bc / C);
    const int y = pix / W, x = pix - y * W;
    float ix, iy;
    hidden_warp_position(T + b * 16, Km + b * 9, depth[static_cast<long long>(b) * HW + pix], x, y, W, H, &ix, &iy);
    const BilinearTaps t = make_taps(ix, iy, W, H);
    const float g = grad_out[i];
    float* plane = grad_src + bc * HW;
    if (t.in_x0 && t.in_y0) atomicAdd(plane + t.y0 * W + t.x0, g * t.w_nw);
    if (t.in_x1 && t.in_y0) atomicAdd(plane + t.y0 * W + t.x0 + 1, g * t.w_ne);
    if (t.in_x0 && t.in_y1) atomicAdd(plane + (t.y0 + 1) * W + t.x0, g * t.w_sw);
    if (t.in_x1 && t.in_y1) atomicAdd(plane + (t.y0 + 1) * W + t.x0 + 1, g * t.w_se);
  }
}

inline int elementwise_grid(long long total, int block) {
  long long g = (total + block - 1) / block;
  const long long cap = 256LL * 8;  // CUs x resident workgroups; grid-stride beyond
  return static_cast<int>(g < cap ? (g > 0 ? g : 1) : cap);
}

}  // namespace dvmvs

extern "C" int dvmvs_hidden_warp_fwd(const float* image_src, const float* depth_dst, const float* src_trans_dst,
                                     const float* camera_matrix, float* out, int B, int C, int H, int W,
                                     int zero_invalid, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!image_src || !depth_dst || !src_trans_dst || !camera_matrix || !out) return DVMVS_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return DVMVS_EINVAL;
  const long long total = static_cast<long long>(B) * C * H * W;
  hipLaunchKernelGGL(hidden_warp_fwd_kernel, dim3(elementwise_grid(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), image_src, depth_dst, src_trans_dst, camera_matrix, out, B, C, H, W,
                     zero_invalid);
  return launch_status();
}

extern "C" int dvmvs_hidden_warp_bwd(const float* grad_out, const float* depth_dst, const float* src_trans_dst,
                                     const float* camera_matrix, float* grad_src, int B, int C, int H, int W,
                                     dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!grad_out || !depth_dst || !src_trans_dst || !camera_matrix || !grad_src) return DVMVS_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return DVMVS_EINVAL;
  const long long total = static_cast<long long>(B) * C * H * W;
  hipLaunchKernelGGL(hidden_warp_bwd_kernel, dim3(elementwise_grid(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), grad_out, depth_dst, src_trans_dst, camera_matrix, grad_src, B, C, H, W);
  return launch_status();
}
