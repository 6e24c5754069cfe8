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

// grad wrt the measurement features.  The adjoint of the bilinear gather is a scatter; done naively it is four fp32
// atomics per (pixel, plane, channel) into global memory -- 3.7 G device-scope atomics for one training step at
// 4 x 7 x 256^2 -- which a multi-XCD part serialises at the memory side.  The scatter is therefore PRIVATISED in LDS with
// the forward kernel's geometry: a workgroup owns a 32x8 reference tile and DP planes; all of its targets lie inside the
// bounding box of 8 sample positions (see publish_sample_box), so it accumulates into a zero-initialised LDS image of that
// box with ds_add_f32 (record stride CCH+1 floats: consecutive box positions fall on different banks) and flushes each
// box element once, coalesced, with a single global atomic.  Global atomics drop by the tile's reuse factor (~15x) and
// the rest stay on chip.  Segments whose box does not fit scatter straight to global memory as before.
constexpr int kBwdPlaneGroups = 4;   // generic fallback geometry (also used when H*W is tiny)
constexpr int kBwdPPT = 4;

template <int TW, int TH, int DP, int CCH, int CAP>
__global__ __launch_bounds__(TW* TH) void cost_volume_bwd_meas_tiled_kernel(CostVolumeBwdArgs a) {
  constexpr int NT = TW * TH;
  constexpr int REC = CCH + 1;
  extern __shared__ __attribute__((aligned(16))) float s_acc[];  // [CAP][REC]
  __shared__ float s_H[DVMVS_MAX_MEASUREMENTS * 9];
  __shared__ float s_kt[DVMVS_MAX_MEASUREMENTS * 3];
  __shared__ float s_ktd[DVMVS_MAX_MEASUREMENTS * DP * 3];
  __shared__ int s_box[5];
  const CostVolumeArgs& f = a.fwd;

  const int tiles_x = (f.W + TW - 1) / TW;
  const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
  const int d_block = (gridDim.y - 1 - blockIdx.y) * DP;
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  sweep_setup(f, b, d_block, DP, tid, NT, s_H, s_kt, s_ktd);

  const int HW = f.H * f.W;
  const int x = tile_x * TW + tid % TW, y = tile_y * TH + tid / TW;
  const bool live = x < f.W && y < f.H;
  const float xf = static_cast<float>(x), yf = static_cast<float>(y);
  const int pix = live ? y * f.W + x : 0;
  const int planes = min(DP, f.D - d_block);
  const float scale = 1.0f / (static_cast<float>(f.M) * static_cast<float>(f.C));
  const float* ref = f.image1 + static_cast<size_t>(b) * f.C * HW + pix;
  const float* g = a.grad_cost + (static_cast<size_t>(b) * f.D + d_block) * HW + pix;

  float gd[DP];   // upstream gradient of this pixel's planes, pre-scaled
#pragma unroll
  for (int j = 0; j < DP; ++j) gd[j] = (live && j < planes) ? g[static_cast<size_t>(j) * HW] * scale : 0.0f;

  for (int m = 0; m < f.M; ++m) {
    float* gmeas = a.grad_image2[m];
    if (!gmeas) continue;                       // workgroup-uniform
    gmeas += static_cast<size_t>(b) * f.C * HW;
    const float* Hm = s_H + m * 9;
    const float* ktd_m = s_ktd + m * DP * 3;
    int seg_lo = 0;
    while (seg_lo < planes) {
      int seg_len = planes - seg_lo;
      int state;
      int base[DP];
      float w[DP][4];
      for (;;) {
        publish_sample_box<TW, TH, DP, CAP>(f, Hm, ktd_m, tile_x, tile_y, seg_lo, seg_lo + seg_len - 1, tid, s_box);
        state = s_box[4];
        int violation = 0;
        if (state == 1) {
          const int x_lo = s_box[0], y_lo = s_box[1], RW = s_box[2], RH = s_box[3];
#pragma unroll
          for (int j = 0; j < DP; ++j) {
            base[j] = 0;
            w[j][0] = w[j][1] = w[j][2] = w[j][3] = 0.0f;
            if (j >= seg_lo && j < seg_lo + seg_len && live) {
              float ix, iy;
              sweep_position(Hm, ktd_m + j * 3, xf, yf, f.W, f.H, &ix, &iy);
              const BilinearTaps t = make_taps(ix, iy, f.W, f.H);
              const bool dead = (t.x0 < -1) || (t.x0 > f.W - 1) || (t.y0 < -1) || (t.y0 > f.H - 1);
              const int rx = t.x0 - x_lo, ry = t.y0 - y_lo;
              const bool inside = (rx >= 0) && (rx + 1 < RW) && (ry >= 0) && (ry + 1 < RH);
              if (!dead && !inside) violation = 1;
              if (!dead && inside) {
                base[j] = ry * RW + rx;
                // taps outside the image land in the apron of the box and are dropped by the flush
                w[j][0] = t.w_nw * gd[j]; w[j][1] = t.w_ne * gd[j]; w[j][2] = t.w_sw * gd[j]; w[j][3] = t.w_se * gd[j];
              }
            }
          }
        }
        if (__syncthreads_or(violation)) state = 0;
        if (state != 0 || seg_len <= 4) break;
        seg_len = max((seg_len + 1) / 2, 4);
      }
      const int seg_hi = seg_lo + seg_len;

      if (state == 1) {
        const int x_lo = s_box[0], y_lo = s_box[1], RW = s_box[2], RH = s_box[3];
        const int RS = RW * RH;
        for (int c0 = 0; c0 < f.C; c0 += CCH) {
          const int nch = min(CCH, f.C - c0);
          for (int i = tid; i < RS * REC; i += NT) s_acc[i] = 0.0f;
          float rv[CCH];
#pragma unroll
          for (int c = 0; c < CCH; ++c) rv[c] = (live && c < nch) ? ref[static_cast<size_t>(c0 + c) * HW] : 0.0f;
          __syncthreads();
          if (live) {
#pragma unroll
            for (int j = 0; j < DP; ++j) {
              if (j >= seg_lo && j < seg_hi && (w[j][0] != 0.0f || w[j][1] != 0.0f || w[j][2] != 0.0f || w[j][3] != 0.0f)) {
                float* r0 = s_acc + base[j] * REC;
                float* r1 = r0 + RW * REC;
#pragma unroll
                for (int c = 0; c < CCH; ++c) {
                  atomicAdd(r0 + c, rv[c] * w[j][0]);
                  atomicAdd(r0 + REC + c, rv[c] * w[j][1]);
                  atomicAdd(r1 + c, rv[c] * w[j][2]);
                  atomicAdd(r1 + REC + c, rv[c] * w[j][3]);
                }
              }
            }
          }
          __syncthreads();
          // flush: channel-major so that consecutive threads hit consecutive x of one channel plane
          for (int i = tid; i < RS * nch; i += NT) {
            const int c = i / RS, r = i - c * RS;
            const int ry = r / RW, rx = r - ry * RW;
            const int gx = x_lo + rx, gy = y_lo + ry;
            const float v = s_acc[r * REC + c];
            if (v != 0.0f && gx >= 0 && gx < f.W && gy >= 0 && gy < f.H)
              atomicAdd(gmeas + static_cast<size_t>(c0 + c) * HW + gy * f.W + gx, v);
          }
          __syncthreads();
        }
      } else if (state == 0 && live) {
        // box does not fit: scatter straight to global memory
        for (int j = seg_lo; j < seg_hi; ++j) {
          float gj = 0.0f;
#pragma unroll
          for (int jj = 0; jj < DP; ++jj)
            if (jj == j) gj = gd[jj];
          if (gj == 0.0f) continue;
          float ix, iy;
          sweep_position(Hm, ktd_m + j * 3, xf, yf, f.W, f.H, &ix, &iy);
          const BilinearTaps t = make_taps(ix, iy, f.W, f.H);
          const bool v0 = t.in_x0 && t.in_y0, v1 = t.in_x1 && t.in_y0, v2 = t.in_x0 && t.in_y1, v3 = t.in_x1 && t.in_y1;
          if (!(v0 || v1 || v2 || v3)) continue;
          const int o0 = t.y0 * f.W + t.x0;
          const float w0 = t.w_nw * gj, w1 = t.w_ne * gj, w2 = t.w_sw * gj, w3 = t.w_se * gj;
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
      seg_lo = seg_hi;
    }
  }
}

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
                                     const float* Hm, const float* kt, float* grad_image1, float* const* grad_image2s,
                                     int B, int M, int C, int H, int W, int D,
                                     double min_depth, double max_depth, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!grad_cost || !grad_image1 || !grad_image2s) return DVMVS_EINVAL;
  CostVolumeBwdArgs a;
  int rc = fill_sweep_args(&a.fwd, image1, image2s, Hm, kt, /*out=*/nullptr, B, M, C, H, W, D, min_depth, max_depth,
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
  if (any_meas && HW >= 64 * 64) {
    constexpr int TW = 32, TH = 8, DP = 8, CCH = 16, CAP = 768;
    constexpr size_t kLds = sizeof(float) * CAP * (CCH + 1);   // 51 KB
    auto kernel = cost_volume_bwd_meas_tiled_kernel<TW, TH, DP, CCH, CAP>;
    // the dynamic-LDS limit is a per-device function attribute; setting it is idempotent, racing threads write the same value
    static bool configured[64] = {};
    int device = 0;
    DVMVS_RETURN_IF_HIP(hipGetDevice(&device));
    const bool tracked = device >= 0 && device < 64;
    if (!tracked || !configured[device]) {
      DVMVS_RETURN_IF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(kLds)));
      if (tracked) configured[device] = true;
    }
    dim3 block(TW * TH), grid(((W + TW - 1) / TW) * ((H + TH - 1) / TH), (D + DP - 1) / DP, B);
    hipLaunchKernelGGL(kernel, grid, block, kLds, s, a);
    rc = launch_status();
  } else if (any_meas) {
    constexpr int kPlanesPerBlock = kBwdPlaneGroups * kBwdPPT;
    dim3 block(kWave, kBwdPlaneGroups), grid((HW + kWave - 1) / kWave, (D + kPlanesPerBlock - 1) / kPlanesPerBlock, B);
    hipLaunchKernelGGL(cost_volume_bwd_meas_kernel, grid, block, 0, s, a);
    rc = launch_status();
  }
  return rc;
}
