// Fused plane-sweep warp + feature correlation (forward) for gfx950.
//
// One launch produces the whole [B,D,H,W] cost volume for all M measurement frames: the per-plane homography,
// the bilinear gather of the measurement features, the channel reduction and the mean over measurement frames
// are fused, so the volume is written exactly once and no warped temporary ever exists in HBM.
// Semantics: /root/reference/dvmvs/utils.py:45-107 (see oracle/dvmvs_oracle.py for the CPU restatement).
#include "plane_sweep.h"

namespace dvmvs {

// ----------------------------------------------------------------------------------------------------------------
// Generic kernel: any C, dot or SAD, arithmetic in the reference's order (interpolate, then reduce channels).
// Workgroup = 64 consecutive pixels x 4 plane sub-groups; each thread owns one pixel and PPT consecutive planes,
// so a wave's tap loads for one channel hit a few consecutive cache lines of the NCHW measurement map.
// ----------------------------------------------------------------------------------------------------------------
constexpr int kGenericPlaneGroups = 4;

template <bool DOT, int PPT>
__global__ __launch_bounds__(kWave* kGenericPlaneGroups) void cost_volume_generic_kernel(CostVolumeArgs a) {
  constexpr int kPlanesPerBlock = kGenericPlaneGroups * PPT;
  __shared__ float s_H[DVMVS_MAX_MEASUREMENTS * 9];
  __shared__ float s_kt[DVMVS_MAX_MEASUREMENTS * 3];
  __shared__ float s_ktd[DVMVS_MAX_MEASUREMENTS * kPlanesPerBlock * 3];

  const int b = blockIdx.z;
  const int d_block = blockIdx.y * kPlanesPerBlock;
  const int tid = threadIdx.y * kWave + threadIdx.x;
  sweep_setup(a, b, d_block, kPlanesPerBlock, tid, kWave * kGenericPlaneGroups, s_H, s_kt, s_ktd);

  const int HW = a.H * a.W;
  const int pix = blockIdx.x * kWave + threadIdx.x;
  if (pix >= HW) return;
  const int y = pix / a.W;
  const int x = pix - y * a.W;
  const float xf = static_cast<float>(x), yf = static_cast<float>(y);
  const int dl0 = threadIdx.y * PPT;

  const float* ref = a.image1 + static_cast<size_t>(b) * a.C * HW + pix;
  float fused[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) fused[j] = 0.0f;

  for (int m = 0; m < a.M; ++m) {
    int off[PPT][4];
    float wgt[PPT][4];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      float ix, iy;
      sweep_position(s_H + m * 9, s_ktd + (m * kPlanesPerBlock + dl0 + j) * 3, xf, yf, a.W, a.H, &ix, &iy);
      const BilinearTaps t = make_taps(ix, iy, a.W, a.H);
      const int xa = t.in_x0 ? t.x0 : 0, xb = t.in_x1 ? t.x0 + 1 : 0;
      const int ya = t.in_y0 ? t.y0 : 0, yb = t.in_y1 ? t.y0 + 1 : 0;
      off[j][0] = ya * a.W + xa;
      off[j][1] = ya * a.W + xb;
      off[j][2] = yb * a.W + xa;
      off[j][3] = yb * a.W + xb;
      wgt[j][0] = (t.in_x0 && t.in_y0) ? t.w_nw : 0.0f;
      wgt[j][1] = (t.in_x1 && t.in_y0) ? t.w_ne : 0.0f;
      wgt[j][2] = (t.in_x0 && t.in_y1) ? t.w_sw : 0.0f;
      wgt[j][3] = (t.in_x1 && t.in_y1) ? t.w_se : 0.0f;
    }
    float acc[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) acc[j] = 0.0f;
    const float* meas = a.image2[m] + static_cast<size_t>(b) * a.C * HW;
    for (int c = 0; c < a.C; ++c) {
      const float r = ref[static_cast<size_t>(c) * HW];
      const float* plane = meas + static_cast<size_t>(c) * HW;
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        float s = plane[off[j][0]] * wgt[j][0];
        s += plane[off[j][1]] * wgt[j][1];
        s += plane[off[j][2]] * wgt[j][2];
        s += plane[off[j][3]] * wgt[j][3];
        if (DOT) acc[j] += r * s;
        else acc[j] += fabsf(r - s);
      }
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) fused[j] += DOT ? acc[j] / static_cast<float>(a.C) : acc[j];
  }

  float* out = a.out + (static_cast<size_t>(b) * a.D + d_block + dl0) * HW + pix;
#pragma unroll
  for (int j = 0; j < PPT; ++j)
    if (d_block + dl0 + j < a.D) out[static_cast<size_t>(j) * HW] = fused[j] / static_cast<float>(a.M);
}

int launch_cost_volume_generic(const CostVolumeArgs& a, bool dot, hipStream_t stream) {
  constexpr int PPT = 4;
  const int HW = a.H * a.W;
  dim3 block(kWave, kGenericPlaneGroups);
  dim3 grid((HW + kWave - 1) / kWave, (a.D + kGenericPlaneGroups * PPT - 1) / (kGenericPlaneGroups * PPT), a.B);
  if (dot) hipLaunchKernelGGL((cost_volume_generic_kernel<true, PPT>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((cost_volume_generic_kernel<false, PPT>), grid, block, 0, stream, a);
  return launch_status();
}


// ----------------------------------------------------------------------------------------------------------------
// Tiled kernel (dot product): the measurement-image footprint of a reference tile is staged through LDS.
//
// A workgroup owns a TW x TH tile of reference pixels and DP consecutive sweep planes.  For one measurement frame the
// samples of the whole tile over those planes fall inside the bounding box of 8 points (4 tile corners x first / last
// plane): a plane-induced homography maps the tile to a convex quadrilateral and the position is monotone in inverse
// depth as long as Z stays positive.  That box (plus a one-pixel zero apron that implements the zeros padding) is
// copied once, coalesced, from the NCHW measurement map into LDS, CCH channels at a time, and TRANSPOSED on the way to
// a channel-interleaved image: one record of CCH floats (+4 floats of padding) per box position.  A tap is then CCH/4
// ds_read_b128 instead of CCH ds_read_b32 -- 256 B/clk instead of 128 B/clk of LDS bandwidth, which is what bounds this
// kernel -- and the (CCH+4)*4-byte record stride (80 B for CCH = 16: an odd multiple of 16 B) puts the 16 lanes of every
// ds_read_b128 service group on 16 different 16-byte bank slots, so neighbouring pixels reading neighbouring records
// do not conflict.  If the box does not fit (large parallax, Z <= 0, non-finite positions) the workgroup falls back to
// the global-memory path for that measurement frame, so the result never depends on the staging succeeding.
// ----------------------------------------------------------------------------------------------------------------
template <int TW, int TH, int DP, int CCH, int CAP>
struct TiledConfig {
  static constexpr int kThreads = TW * TH;
  static constexpr int kSlots = (CAP + kThreads - 1) / kThreads;  // box positions staged per thread
  static constexpr int kRec = CCH + 4;                              // floats per LDS record (payload + bank-spreading pad)
  static constexpr size_t kLdsBytes = sizeof(float) * (static_cast<size_t>(kRec) * CAP);
  static_assert(CCH % 4 == 0 && ((kRec / 4) % 2) == 1, "record stride must be an odd number of 16-byte slots");
};

template <int TW, int TH, int DP, int CCH, int CAP>
__global__ __launch_bounds__(TW* TH) void cost_volume_tiled_kernel(CostVolumeArgs a) {
  using Cfg = TiledConfig<TW, TH, DP, CCH, CAP>;
  constexpr int NT = Cfg::kThreads;
  extern __shared__ __attribute__((aligned(16))) float s_tile[];  // [CAP][kRec]
  __shared__ float s_H[DVMVS_MAX_MEASUREMENTS * 9];
  __shared__ float s_kt[DVMVS_MAX_MEASUREMENTS * 3];
  __shared__ float s_ktd[DVMVS_MAX_MEASUREMENTS * DP * 3];
  __shared__ int s_box[5];  // x_lo, y_lo, RW, RH, usable

  const int tiles_x = (a.W + TW - 1) / TW;
  const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
  const int d_block = blockIdx.y * DP;
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  sweep_setup(a, b, d_block, DP, tid, NT, s_H, s_kt, s_ktd);

  const int HW = a.H * a.W;
  const int x = tile_x * TW + tid % TW, y = tile_y * TH + tid / TW;
  const bool live = x < a.W && y < a.H;
  const float xf = static_cast<float>(x), yf = static_cast<float>(y);
  const int pix = live ? y * a.W + x : 0;
  const int planes = min(DP, a.D - d_block);

  float fused[DP];
#pragma unroll
  for (int j = 0; j < DP; ++j) fused[j] = 0.0f;

  for (int m = 0; m < a.M; ++m) {
    const float* Hm = s_H + m * 9;
    // ---- bounding box of the tile's samples over this workgroup's planes (threads 0..7: corner x extreme plane) ----
    if (tid < 64) {
      float ix = 0.0f, iy = 0.0f, z = 1.0f;
      if (tid < 8) {
        const int cx = (tid & 1) ? min(tile_x * TW + TW - 1, a.W - 1) : tile_x * TW;
        const int cy = (tid & 2) ? min(tile_y * TH + TH - 1, a.H - 1) : tile_y * TH;
        const int dl = (tid & 4) ? planes - 1 : 0;
        sweep_position(Hm, s_ktd + (m * DP + dl) * 3, static_cast<float>(cx), static_cast<float>(cy), a.W, a.H, &ix, &iy, &z);
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
        // NaN-safe: every comparison below is false for NaN, which leaves usable == 0
        const bool finite = (lo_x > -1e6f) && (hi_x < 1e6f) && (lo_y > -1e6f) && (hi_y < 1e6f) && (lo_z > 1e-6f);
        int usable = 0, x_lo = 0, y_lo = 0, RW = 0, RH = 0;
        if (finite) {
          // 0.05 px of slack for round-off between the corner samples and interior pixels; one apron pixel outside
          // the image is enough, everything further out is zero as well
          x_lo = max(-1, static_cast<int>(floorf(lo_x - 0.05f)));
          y_lo = max(-1, static_cast<int>(floorf(lo_y - 0.05f)));
          const int x_hi = min(a.W, static_cast<int>(floorf(hi_x + 0.05f)) + 1);
          const int y_hi = min(a.H, static_cast<int>(floorf(hi_y + 0.05f)) + 1);
          RW = x_hi - x_lo + 1;
          RH = y_hi - y_lo + 1;
          if (RW <= 0 || RH <= 0) {
            usable = 2;  // the whole footprint lies outside the image: this frame contributes zeros
          } else if (RW * RH <= CAP) {
            usable = 1;
          }
        }
        s_box[0] = x_lo; s_box[1] = y_lo; s_box[2] = RW; s_box[3] = RH; s_box[4] = usable;
      }
    }
    __syncthreads();
    const int x_lo = s_box[0], y_lo = s_box[1], RW = s_box[2], RH = s_box[3];
    int usable = s_box[4];

    // ---- this thread's taps: region-relative base offsets and weights, checked against the box ----
    int base[DP];
    float w_nw[DP], w_ne[DP], w_sw[DP], w_se[DP];
    int violation = 0;
    if (usable == 1) {
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        float ix, iy;
        sweep_position(Hm, s_ktd + (m * DP + j) * 3, xf, yf, a.W, a.H, &ix, &iy);
        const BilinearTaps t = make_taps(ix, iy, a.W, a.H);
        // taps entirely outside [-1, W] x [-1, H] see only zeros; everything else must lie inside the staged box
        const bool dead = (t.x0 < -1) || (t.x0 > a.W - 1) || (t.y0 < -1) || (t.y0 > a.H - 1) || (j >= planes) || !live;
        const int rx = t.x0 - x_lo, ry = t.y0 - y_lo;
        const bool inside = (rx >= 0) && (rx + 1 < RW) && (ry >= 0) && (ry + 1 < RH);
        if (!dead && !inside) violation = 1;
        const bool use = !dead && inside;
        base[j] = use ? ry * RW + rx : 0;
        w_nw[j] = use ? t.w_nw : 0.0f;
        w_ne[j] = use ? t.w_ne : 0.0f;
        w_sw[j] = use ? t.w_sw : 0.0f;
        w_se[j] = use ? t.w_se : 0.0f;
      }
    }
    if (__syncthreads_or(violation)) usable = 0;

    if (usable == 2) continue;

    float acc[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) acc[j] = 0.0f;
    const float* meas = a.image2[m] + static_cast<size_t>(b) * a.C * HW;
    const float* ref = a.image1 + static_cast<size_t>(b) * a.C * HW + pix;

    if (usable == 1) {
      // ---- staging plan: each thread copies up to kSlots region elements per channel ----
      const int RS = RW * RH;
      int goff[Cfg::kSlots];
      bool gin[Cfg::kSlots];
#pragma unroll
      for (int k = 0; k < Cfg::kSlots; ++k) {
        const int r = tid + k * NT;
        const int ry = r / RW, rx = r - ry * RW;
        const int gx = x_lo + rx, gy = y_lo + ry;
        gin[k] = (r < RS) && (gx >= 0) && (gx < a.W) && (gy >= 0) && (gy < a.H);
        goff[k] = gin[k] ? gy * a.W + gx : 0;
      }
      constexpr int REC = Cfg::kRec;
      typedef float float4v __attribute__((ext_vector_type(4)));
      for (int c0 = 0; c0 < a.C; c0 += CCH) {
        const int nch = min(CCH, a.C - c0);
        // reference features of this pass: issued before the staging loads so their latency overlaps the copy
        float rv[CCH];
#pragma unroll
        for (int c = 0; c < CCH; ++c) rv[c] = (c < nch) ? ref[static_cast<size_t>(c0 + c) * HW] : 0.0f;
        // global (NCHW, coalesced along x) -> registers -> LDS records (transposed): all loads of a position are in
        // flight before its CCH/4 ds_write_b128
#pragma unroll
        for (int k = 0; k < Cfg::kSlots; ++k) {
          const int r = tid + k * NT;
          float4v v[CCH / 4];
#pragma unroll
          for (int c = 0; c < CCH; ++c) {
            const float* plane = meas + static_cast<size_t>(c0 + min(c, nch - 1)) * HW;
            v[c / 4][c % 4] = (gin[k] && c < nch) ? plane[goff[k]] : 0.0f;
          }
          if (r < RS) {
#pragma unroll
            for (int q = 0; q < CCH / 4; ++q) *reinterpret_cast<float4v*>(s_tile + r * REC + q * 4) = v[q];
          }
        }
        __syncthreads();
        if (live) {
#pragma unroll
          for (int j = 0; j < DP; ++j) {
            const float* row0 = s_tile + base[j] * REC;
            const float* row1 = row0 + RW * REC;
            float sum = 0.0f;
#pragma unroll
            for (int q = 0; q < CCH / 4; ++q) {
              const float4v nw = *reinterpret_cast<const float4v*>(row0 + q * 4);
              const float4v ne = *reinterpret_cast<const float4v*>(row0 + REC + q * 4);
              const float4v sw = *reinterpret_cast<const float4v*>(row1 + q * 4);
              const float4v se = *reinterpret_cast<const float4v*>(row1 + REC + q * 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float t = nw[e] * w_nw[j];
                t += ne[e] * w_ne[j];
                t += sw[e] * w_sw[j];
                t += se[e] * w_se[j];
                sum += rv[q * 4 + e] * t;   // rv == 0 for channels beyond nch
              }
            }
            acc[j] += sum;
          }
        }
        __syncthreads();
      }
    } else if (live) {
      // ---- fallback: taps straight from global memory (same arithmetic as the generic kernel) ----
      for (int j = 0; j < planes; ++j) {
        float ix, iy;
        sweep_position(Hm, s_ktd + (m * DP + j) * 3, xf, yf, a.W, a.H, &ix, &iy);
        const BilinearTaps t = make_taps(ix, iy, a.W, a.H);
        const int xa = t.in_x0 ? t.x0 : 0, xb = t.in_x1 ? t.x0 + 1 : 0;
        const int ya = t.in_y0 ? t.y0 : 0, yb = t.in_y1 ? t.y0 + 1 : 0;
        const float g0 = (t.in_x0 && t.in_y0) ? t.w_nw : 0.0f, g1 = (t.in_x1 && t.in_y0) ? t.w_ne : 0.0f;
        const float g2 = (t.in_x0 && t.in_y1) ? t.w_sw : 0.0f, g3 = (t.in_x1 && t.in_y1) ? t.w_se : 0.0f;
        float sum = 0.0f;
        for (int c = 0; c < a.C; ++c) {
          const float* plane = meas + static_cast<size_t>(c) * HW;
          float s = plane[ya * a.W + xa] * g0;
          s += plane[ya * a.W + xb] * g1;
          s += plane[yb * a.W + xa] * g2;
          s += plane[yb * a.W + xb] * g3;
          sum += ref[static_cast<size_t>(c) * HW] * s;
        }
        // acc[] is indexed with a compile-time constant below, so select instead of indexing dynamically
#pragma unroll
        for (int jj = 0; jj < DP; ++jj)
          if (jj == j) acc[jj] = sum;
      }
    }
#pragma unroll
    for (int j = 0; j < DP; ++j) fused[j] += acc[j] / static_cast<float>(a.C);
  }

  if (live) {
    float* out = a.out + (static_cast<size_t>(b) * a.D + d_block) * HW + pix;
#pragma unroll
    for (int j = 0; j < DP; ++j)
      if (j < planes) out[static_cast<size_t>(j) * HW] = fused[j] / static_cast<float>(a.M);
  }
}

template <int TW, int TH, int DP, int CCH, int CAP>
int launch_cost_volume_tiled(const CostVolumeArgs& a, hipStream_t stream) {
  using Cfg = TiledConfig<TW, TH, DP, CCH, CAP>;
  auto kernel = cost_volume_tiled_kernel<TW, TH, DP, CCH, CAP>;
  static bool configured = false;  // raising the dynamic-LDS limit is idempotent; racing threads set the same value
  if (!configured) {
    DVMVS_RETURN_IF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            static_cast<int>(Cfg::kLdsBytes)));
    configured = true;
  }
  const int tiles = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
  dim3 grid(tiles, (a.D + DP - 1) / DP, a.B), block(Cfg::kThreads);
  hipLaunchKernelGGL(kernel, grid, block, Cfg::kLdsBytes, stream, a);
  return launch_status();
}

// One thread per (batch, measurement frame): the matrices above into the caller's workspace, so that the sweep
// kernels (hundreds of workgroups) do not each repeat the fp64 inverse.
__global__ void sweep_setup_kernel(CostVolumeArgs a, float* setup) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.B * a.M) return;
  const int b = i / a.M, m = i - b * a.M;
  float Hm[9], kt[3];
  sweep_matrices(a.pose1 + b * 16, a.pose2[m] + b * 16, a.K + b * 9, Hm, kt);
  float* out = setup + static_cast<size_t>(i) * kSetupFloats;
#pragma unroll
  for (int k = 0; k < 9; ++k) out[k] = Hm[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) out[9 + k] = kt[k];
}


}  // namespace dvmvs

extern "C" size_t dvmvs_cost_volume_workspace_bytes(int B, int M) {
  if (B <= 0 || M <= 0) return 0;
  return sizeof(float) * static_cast<size_t>(B) * M * dvmvs::kSetupFloats;
}

extern "C" int dvmvs_cost_volume_fwd(const float* image1, const float* const* image2s, const float* pose1,
                                     const float* const* pose2s, const float* K, float* cost_volume,
                                     int B, int M, int C, int H, int W, int D,
                                     double min_depth, double max_depth, int dot_product, int variant,
                                     float* workspace, size_t workspace_bytes, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (variant < 0 || (variant > 2 && variant < 16) || variant > 31) return DVMVS_EINVAL;
  if (variant == 2 && !dot_product) return DVMVS_EUNSUPPORTED;
  CostVolumeArgs a;
  const int rc = fill_sweep_args(&a, image1, image2s, pose1, pose2s, K, cost_volume, B, M, C, H, W, D, min_depth, max_depth, true);
  if (rc != 0) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (workspace != nullptr) {
    if (workspace_bytes < dvmvs_cost_volume_workspace_bytes(B, M)) return DVMVS_EINVAL;
    hipLaunchKernelGGL(sweep_setup_kernel, dim3((B * M + 63) / 64), dim3(64), 0, s, a, workspace);
    const int src = launch_status();
    if (src != 0) return src;
    a.setup = workspace;
  }
  if (variant >= 16) {
    // tuning configurations for tools/cv_microbench.py (TW, TH, DP, CCH, CAP); not part of the stable interface
    if (!dot_product) return DVMVS_EUNSUPPORTED;
    switch (variant - 16) {
      case 0: return launch_cost_volume_tiled<32, 8, 8, 16, 640>(a, s);    // 50 KB LDS, 3 workgroups / CU
      case 1: return launch_cost_volume_tiled<64, 4, 8, 16, 640>(a, s);
      case 2: return launch_cost_volume_tiled<32, 8, 8, 16, 768>(a, s);    // 60 KB, 2 / CU
      case 3: return launch_cost_volume_tiled<64, 4, 8, 16, 768>(a, s);
      case 4: return launch_cost_volume_tiled<32, 8, 8, 8, 1024>(a, s);    // 48 KB, 48-byte records
      case 5: return launch_cost_volume_tiled<32, 8, 16, 8, 1024>(a, s);
      case 6: return launch_cost_volume_tiled<32, 4, 8, 16, 512>(a, s);    // 128-thread workgroups, 40 KB
      case 7: return launch_cost_volume_tiled<64, 2, 8, 16, 512>(a, s);
      case 8: return launch_cost_volume_tiled<32, 8, 4, 16, 512>(a, s);    // 4 planes / workgroup: 1280 workgroups
      case 9: return launch_cost_volume_tiled<64, 4, 4, 16, 512>(a, s);
      case 10: return launch_cost_volume_tiled<32, 8, 16, 16, 960>(a, s);  // 75 KB, 2 / CU, 16 planes
      case 11: return launch_cost_volume_tiled<64, 4, 16, 16, 960>(a, s);
      default: return DVMVS_EINVAL;
    }
  }
  const bool tiled = dot_product && (variant == 2 || (variant == 0 && H * W >= 64 * 64));
  if (tiled) return launch_cost_volume_tiled<32, 8, 8, 16, 640>(a, s);
  return launch_cost_volume_generic(a, dot_product != 0, s);
}
