// Gradient of the fused plane-sweep cost volume (dot-product mode) w.r.t. both feature maps, gfx950.
//
// cost[b,d,p] = 1/(M*C) * sum_m sum_c f1[c,p] * sum_t w_t(m,d,p) * f2_m[c, q_t(m,d,p)]
//   d f1[c,p]   = 1/(M*C) * sum_m sum_d g[d,p] * warped_m[c,d,p]                       -> gather kernel, no atomics
//   d f2_m[c,q] = 1/(M*C) * sum_d sum_p g[d,p] * f1[c,p] * w_t   for taps with q_t = q -> scatter kernel, atomics
// The sampling positions depend only on poses / intrinsics, which are data (no gradient), as in autograd through
// /root/reference/dvmvs/utils.py:75-82.
#include "plane_sweep.h"

namespace dvmvs {

struct CostVolumeBwdArgs {
  CostVolumeArgs fwd;  // image1, image2[], poses, K, shapes, plane spacing; fwd.out unused
  const float* grad_cost;
  float* grad_image1;
  float* grad_image2[DVMVS_MAX_MEASUREMENTS];
};

constexpr int kBwdChannelChunk = 8;

// grad wrt the reference features: workgroup = 64 pixels x 4 channel chunks of 8; every thread walks all planes.
__global__ __launch_bounds__(256) void cost_volume_bwd_ref_kernel(CostVolumeBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const CostVolumeArgs& f = a.fwd;
  float* s_H = smem;
  float* s_kt = s_H + DVMVS_MAX_MEASUREMENTS * 9;
  float* s_ktd = s_kt + DVMVS_MAX_MEASUREMENTS * 3;  // [M][D][3]
  const int b = blockIdx.z;
  const int tid = threadIdx.y * kWave + threadIdx.x;
  sweep_setup(f, b, 0, f.D, tid, 256, s_H, s_kt, s_ktd);

  const int HW = f.H * f.W;
  const int pix = blockIdx.x * kWave + threadIdx.x;
  const int c0 = (blockIdx.y * 4 + threadIdx.y) * kBwdChannelChunk;
  if (pix >= HW || c0 >= f.C) return;
  const int y = pix / f.W, x = pix - y * f.W;
  const float xf = static_cast<float>(x), yf = static_cast<float>(y);
  const int nch = min(kBwdChannelChunk, f.C - c0);

  float acc[kBwdChannelChunk];
#pragma unroll
  for (int k = 0; k < kBwdChannelChunk; ++k) acc[k] = 0.0f;
  const float* g = a.grad_cost + static_cast<size_t>(b) * f.D * HW + pix;
  for (int m = 0; m < f.M; ++m) {
    const float* meas = f.image2[m] + (static_cast<size_t>(b) * f.C + c0) * HW;
    for (int d = 0; d < f.D; ++d) {
      float ix, iy;
      sweep_position(s_H + m * 9, s_ktd + (m * f.D + d) * 3, xf, yf, f.W, f.H, &ix, &iy);
      const BilinearTaps t = make_taps(ix, iy, f.W, f.H);
      const int xa = t.in_x0 ? t.x0 : 0, xb = t.in_x1 ? t.x0 + 1 : 0;
      const int ya = t.in_y0 ? t.y0 : 0, yb = t.in_y1 ? t.y0 + 1 : 0;
      const float gd = g[static_cast<size_t>(d) * HW];
      const float w0 = (t.in_x0 && t.in_y0) ? t.w_nw * gd : 0.0f;
      const float w1 = (t.in_x1 && t.in_y0) ? t.w_ne * gd : 0.0f;
      const float w2 = (t.in_x0 && t.in_y1) ? t.w_sw * gd : 0.0f;
      const float w3 = (t.in_x1 && t.in_y1) ? t.w_se * gd : 0.0f;
      const int o0 = ya * f.W + xa, o1 = ya * f.W + xb, o2 = yb * f.W + xa, o3 = yb * f.W + xb;
#pragma unroll
      for (int k = 0; k < kBwdChannelChunk; ++k) {
        if (k < nch) {
          const float* plane = meas + static_cast<size_t>(k) * HW;
          acc[k] += plane[o0] * w0 + plane[o1] * w1 + plane[o2] * w2 + plane[o3] * w3;
        }
      }
    }
  }
  const float scale = 1.0f / (static_cast<float>(f.M) * static_cast<float>(f.C));
  float* out = a.grad_image1 + (static_cast<size_t>(b) * f.C + c0) * HW + pix;
#pragma unroll
  for (int k = 0; k < kBwdChannelChunk; ++k)
    if (k < nch) out[static_cast<size_t>(k) * HW] = acc[k] * scale;
}

// grad wrt the measurement features: same thread layout as the forward generic kernel, scatter with atomics.
constexpr int kBwdPlaneGroups = 4;
constexpr int kBwdPPT = 4;

__global__ __launch_bounds__(256) void cost_volume_bwd_meas_kernel(CostVolumeBwdArgs a) {
  constexpr int kPlanesPerBlock = kBwdPlaneGroups * kBwdPPT;
  __shared__ float s_H[DVMVS_MAX_MEASUREMENTS * 9];
  __shared__ float s_kt[DVMVS_MAX_MEASUREMENTS * 3];
  __shared__ float s_ktd[DVMVS_MAX_MEASUREMENTS * kPlanesPerBlock * 3];
  const CostVolumeArgs& f = a.fwd;
  const int b = blockIdx.z;
  const int d_block = blockIdx.y * kPlanesPerBlock;
  const int tid = threadIdx.y * kWave + threadIdx.x;
  sweep_setup(f, b, d_block, kPlanesPerBlock, tid, 256, s_H, s_kt, s_ktd);

  const int HW = f.H * f.W;
  const int pix = blockIdx.x * kWave + threadIdx.x;
  if (pix >= HW) return;
  const int y = pix / f.W, x = pix - y * f.W;
  const float xf = static_cast<float>(x), yf = static_cast<float>(y);
  const int dl0 = threadIdx.y * kBwdPPT;
  const float scale = 1.0f / (static_cast<float>(f.M) * static_cast<float>(f.C));
  const float* ref = f.image1 + static_cast<size_t>(b) * f.C * HW + pix;
  const float* g = a.grad_cost + (static_cast<size_t>(b) * f.D + d_block + dl0) * HW + pix;

  for (int m = 0; m < f.M; ++m) {
    float* gmeas = a.grad_image2[m];
    if (!gmeas) continue;
    gmeas += static_cast<size_t>(b) * f.C * HW;
#pragma unroll
    for (int j = 0; j < kBwdPPT; ++j) {
      const int d = d_block + dl0 + j;
      if (d >= f.D) break;
      float ix, iy;
      sweep_position(s_H + m * 9, s_ktd + (m * kPlanesPerBlock + dl0 + j) * 3, xf, yf, f.W, f.H, &ix, &iy);
      const BilinearTaps t = make_taps(ix, iy, f.W, f.H);
      const float gd = g[static_cast<size_t>(j) * HW] * scale;
      const bool v0 = t.in_x0 && t.in_y0, v1 = t.in_x1 && t.in_y0, v2 = t.in_x0 && t.in_y1, v3 = t.in_x1 && t.in_y1;
      if (!(v0 || v1 || v2 || v3) || gd == 0.0f) continue;
      const int o0 = t.y0 * f.W + t.x0;
      const float w0 = t.w_nw * gd, w1 = t.w_ne * gd, w2 = t.w_sw * gd, w3 = t.w_se * gd;
      for (int c = 0; c < f.C; ++c) {
        const float r = ref[static_cast<size_t>(c) * HW];
        float* plane = gmeas + static_cast<size_t>(c) * HW;
        if (v0) atomicAdd(plane + o0, r * w0);
        if (v1) atomicAdd(plane + o0 + 1, r * w1);
        if (v2) atomicAdd(plane + o0 + f.W, r * w2);
        if (v3) atomicAdd(plane + o0 + f.W + 1, r * w3);
      }
    }
  }
}

}  // namespace dvmvs

extern "C" int dvmvs_cost_volume_bwd(const float* grad_cost, const float* image1, const float* const* image2s,
                                     const float* pose1, const float* const* pose2s, const float* K,
                                     float* grad_image1, float* const* grad_image2s,
                                     int B, int M, int C, int H, int W, int D,
                                     double min_depth, double max_depth, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!grad_cost || !grad_image1 || !grad_image2s) return DVMVS_EINVAL;
  CostVolumeBwdArgs a;
  int rc = fill_sweep_args(&a.fwd, image1, image2s, pose1, pose2s, K, /*out=*/nullptr, B, M, C, H, W, D, min_depth, max_depth,
                           /*need_out=*/false);
  if (rc != 0) return rc;
  a.grad_cost = grad_cost;
  a.grad_image1 = grad_image1;
  bool any_meas = false;
  for (int m = 0; m < DVMVS_MAX_MEASUREMENTS; ++m) {
    a.grad_image2[m] = m < M ? grad_image2s[m] : nullptr;
    any_meas |= a.grad_image2[m] != nullptr;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int HW = H * W;
  {
    const size_t smem = sizeof(float) * (DVMVS_MAX_MEASUREMENTS * 12 + static_cast<size_t>(M) * D * 3);
    dim3 block(kWave, 4), grid((HW + kWave - 1) / kWave, (C + 4 * kBwdChannelChunk - 1) / (4 * kBwdChannelChunk), B);
    hipLaunchKernelGGL(cost_volume_bwd_ref_kernel, grid, block, smem, s, a);
    rc = launch_status();
    if (rc != 0) return rc;
  }
  if (any_meas) {
    constexpr int kPlanesPerBlock = kBwdPlaneGroups * kBwdPPT;
    dim3 block(kWave, kBwdPlaneGroups), grid((HW + kWave - 1) / kWave, (D + kPlanesPerBlock - 1) / kPlanesPerBlock, B);
    hipLaunchKernelGGL(cost_volume_bwd_meas_kernel, grid, block, 0, s, a);
    rc = launch_status();
  }
  return rc;
}
