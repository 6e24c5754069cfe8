// Plane-sweep geometry shared by the forward and backward cost-volume kernels (gfx950).
// Semantics: /root/reference/dvmvs/utils.py:51-73.
#pragma once

#include "dvmvs_device.h"

namespace dvmvs {

struct CostVolumeArgs {
  const float* image1;
  const float* image2[DVMVS_MAX_MEASUREMENTS];
  const float* Hm;   // [B,M,9]  K R K^-1, computed by the caller (reference rounding) or by dvmvs_sweep_matrices (fp64)
  const float* kt;   // [B,M,3]  K t
  float* out;
  int B, M, C, H, W, D;
  double inv_depth_base, inv_depth_step;
  int image2_nhwc;     // measurement maps are channels-last ([B,H,W,C]); reference map and output stay NCHW
  unsigned int* spill;   // optional spill workspace of the two-pass tiled sweep (layout: sweep_tiled.hip); nullptr = gather inline
  const unsigned int* items;   // optional work list of the tiled sweep (dvmvs_sweep_work_list; layout: sweep_tiled.hip); nullptr = static numbering
};

// Per-(batch, measurement) sweep constants in fp64, rounded once (dvmvs_sweep_matrices: the opt-in "exact" pose algebra).
//   Hm = K R K^-1 (row-major 3x3), kt = K t   with [R|t] = inverse(pose2) * pose1     (utils.py:51-56)
__device__ inline void sweep_matrices(const float* pose1, const float* pose2, const float* K, float* Hm, float* kt) {
  double E[16];
  relative_pose_f64(pose2, pose1, E);
  double Kd[9], Kinv[9], R[9], KR[9], KRKinv[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Kd[i] = static_cast<double>(K[i]);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) R[r * 3 + c] = E[r * 4 + c];
  inverse3(Kd, Kinv);
  matmul3(Kd, R, KR);
  matmul3(KR, Kinv, KRKinv);
#pragma unroll
  for (int i = 0; i < 9; ++i) Hm[i] = static_cast<float>(KRKinv[i]);
#pragma unroll
  for (int r = 0; r < 3; ++r)
    kt[r] = static_cast<float>(Kd[r * 3 + 0] * E[0 * 4 + 3] + Kd[r * 3 + 1] * E[1 * 4 + 3] + Kd[r * 3 + 2] * E[2 * 4 + 3]);
}

// depth of sweep plane d as the reference's python-double expression, rounded to fp32 where it meets the tensor
__device__ inline float plane_depth(double inv_base, double inv_step, int d) {
  return static_cast<float>(1.0 / (inv_base + static_cast<double>(d) * inv_step));
}

// Fills s_H[M][9], s_kt[M][3] (copies of the caller's matrices for batch item b) and s_ktd[M][planes][3] (= kt / depth_d for d
// in [d_begin, d_begin+planes): the reference's `Kt / this_depth`, utils.py:66-68, an IEEE fp32 division by the fp32-rounded depth).
__device__ inline void sweep_setup(const CostVolumeArgs& a, int b, int d_begin, int planes, int tid, int nthreads,
                                   float* s_H, float* s_kt, float* s_ktd) {
  const float* Hm = a.Hm + static_cast<size_t>(b) * a.M * 9;
  const float* kt = a.kt + static_cast<size_t>(b) * a.M * 3;
  for (int i = tid; i < a.M * 9; i += nthreads) s_H[i] = Hm[i];
  for (int i = tid; i < a.M * 3; i += nthreads) s_kt[i] = kt[i];
  for (int i = tid; i < a.M * planes * 3; i += nthreads) {
    const int k = i % 3;
    const int dl = (i / 3) % planes;
    const int m = i / (3 * planes);
    const int d = d_begin + dl;
    s_ktd[i] = (d < a.D) ? kt[m * 3 + k] / plane_depth(a.inv_depth_base, a.inv_depth_step, d) : 0.0f;
  }
  __syncthreads();
}

// Sample position of reference pixel (x, y) on one plane of one measurement frame, in measurement-image pixels.
// Op order follows utils.py:68-73 and ATen's align_corners un-normalisation.
__device__ inline void sweep_position(const float* Hm, const float* ktd, float xf, float yf, int W, int H, float* ix, float* iy,
                                      float* z_out = nullptr) {
  const float X = fmaf(Hm[2], 1.0f, fmaf(Hm[1], yf, Hm[0] * xf)) + ktd[0];
  const float Y = fmaf(Hm[5], 1.0f, fmaf(Hm[4], yf, Hm[3] * xf)) + ktd[1];
  const float Z = fmaf(Hm[8], 1.0f, fmaf(Hm[7], yf, Hm[6] * xf)) + ktd[2];
  const float denom = Z + 1e-8f;
  if (z_out) *z_out = denom;
  const float u = X / denom;
  const float v = Y / denom;
  const float wn = static_cast<float>(W) * 0.5f;
  const float hn = static_cast<float>(H) * 0.5f;
  *ix = unnormalize_ac((u - wn) / wn, W);
  *iy = unnormalize_ac((v - hn) / hn, H);
}

// Validates the shared arguments of the forward / backward entry points and fills the kernel argument block.
inline int fill_sweep_args(CostVolumeArgs* a, const float* image1, const float* const* image2s, const float* Hm, const float* kt,
                           float* out, int B, int M, int C, int H, int W, int D, double min_depth, double max_depth, bool need_out) {
  if (!image1 || !image2s || !Hm || !kt || (need_out && !out)) return DVMVS_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || D <= 0 || M <= 0) return DVMVS_EINVAL;
  if (M > DVMVS_MAX_MEASUREMENTS || D > DVMVS_MAX_DEPTH_LEVELS || B > 65535) return DVMVS_EUNSUPPORTED;
  if (static_cast<long long>(C) * H * W >= (1LL << 31)) return DVMVS_EUNSUPPORTED;
  if (!(min_depth > 0.0) || !(max_depth > 0.0)) return DVMVS_EINVAL;
  a->image1 = image1;
  a->Hm = Hm;
  a->kt = kt;
  a->out = out;
  for (int m = 0; m < DVMVS_MAX_MEASUREMENTS; ++m) {
    if (m < M && !image2s[m]) return DVMVS_EINVAL;
    a->image2[m] = m < M ? image2s[m] : nullptr;
  }
  a->B = B; a->M = M; a->C = C; a->H = H; a->W = W; a->D = D;
  // utils.py:59-60, python doubles
  a->inv_depth_base = 1.0 / max_depth;
  a->inv_depth_step = D > 1 ? (1.0 / min_depth - 1.0 / max_depth) / (D - 1) : 0.0;
  a->spill = nullptr;
  a->items = nullptr;
  a->image2_nhwc = 0;
  return 0;
}

// Bounding box of the tile's sample positions over planes [j_lo, j_hi] of measurement frame m, evaluated by the first 8
// lanes (tile corner x extreme plane) and published through s_box: {x_lo, y_lo, RW, RH, state}.
//   state 1: box staged through LDS; 2: box entirely outside the image (zeros); 0: does not fit / not well defined.
template <int TW, int TH, int DP, int CAP>
__device__ inline void publish_sample_box(const CostVolumeArgs& a, const float* Hm, const float* ktd_m, int tile_x, int tile_y,
                                          int j_lo, int j_hi, int tid, int* s_box) {
  if (tid < 64) {
    float ix = 0.0f, iy = 0.0f, z = 1.0f;
    if (tid < 8) {
      const int cx = (tid & 1) ? min(tile_x * TW + TW - 1, a.W - 1) : tile_x * TW;
      const int cy = (tid & 2) ? min(tile_y * TH + TH - 1, a.H - 1) : tile_y * TH;
      const int dl = (tid & 4) ? j_hi : j_lo;
      sweep_position(Hm, ktd_m + dl * 3, static_cast<float>(cx), static_cast<float>(cy), a.W, a.H, &ix, &iy, &z);
    }
    float lo_x = ix, hi_x = ix, lo_y = iy, hi_y = iy, lo_z = z;
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) {
      lo_x = fminf(lo_x, __shfl_xor(lo_x, off, 8));
      hi_x = fmaxf(hi_x, __shfl_xor(hi_x, off, 8));
      lo_y = fminf(lo_y, __shfl_xor(lo_y, off, 8));
      hi_y = fmaxf(hi_y, __shfl_xor(hi_y, off, 8));
      lo_z = fminf(lo_z, __shfl_xor(lo_z, off, 8));
    }
    if (tid == 0) {
      // NaN-safe: every comparison below is false for NaN, which leaves state == 0
      const bool finite = (lo_x > -1e6f) && (hi_x < 1e6f) && (lo_y > -1e6f) && (hi_y < 1e6f) && (lo_z > 1e-6f);
      int state = 0, x_lo = 0, y_lo = 0, RW = 0, RH = 0;
      if (finite) {
        // 0.05 px of slack for round-off between the corner samples and interior pixels; one apron pixel outside the
        // image is enough, everything further out is zero as well
        x_lo = max(-1, static_cast<int>(floorf(lo_x - 0.05f)));
        y_lo = max(-1, static_cast<int>(floorf(lo_y - 0.05f)));
        const int x_hi = min(a.W, static_cast<int>(floorf(hi_x + 0.05f)) + 1);
        const int y_hi = min(a.H, static_cast<int>(floorf(hi_y + 0.05f)) + 1);
        RW = x_hi - x_lo + 1;
        RH = y_hi - y_lo + 1;
        if (RW <= 0 || RH <= 0) state = 2;
        else if (RW * RH <= CAP) state = 1;
      }
      s_box[0] = x_lo; s_box[1] = y_lo; s_box[2] = RW; s_box[3] = RH; s_box[4] = state;
    }
  }
  __syncthreads();
}

}  // namespace dvmvs
