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
  const int d_block = (gridDim.y - 1 - blockIdx.y) * kPlanesPerBlock;   // near (scattered) planes first, see the tiled kernel
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

// NHWC: the measurement maps are channels-last ([B,H,W,C] in memory).  A box position's CCH channels are then 4*CCH
// contiguous bytes (staging = plain 16-byte copies, no transposition through registers) and, more importantly, a gather
// tap of the spill path is one cache line for all 32 channels instead of 32 lines.
template <int TW, int TH, int DP, int CCH, int CAP, bool NHWC>
__global__ __launch_bounds__(TW* TH) void cost_volume_tiled_kernel(CostVolumeArgs a) {
  using Cfg = TiledConfig<TW, TH, DP, CCH, CAP>;
  constexpr int NT = Cfg::kThreads;
  constexpr int REC = Cfg::kRec;
  typedef float float4v __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float s_tile[];  // [CAP][kRec]
  __shared__ float s_H[DVMVS_MAX_MEASUREMENTS * 9];
  __shared__ float s_kt[DVMVS_MAX_MEASUREMENTS * 3];
  __shared__ float s_ktd[DVMVS_MAX_MEASUREMENTS * DP * 3];
  __shared__ int s_box[5];

  const int tiles_x = (a.W + TW - 1) / TW;
  const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
  // Near planes first: their footprints are the large ones (parallax and magnification grow with inverse depth), so the
  // workgroups most likely to spill to the gather path are dispatched first and the cheap far planes fill the tail.
  const int d_block = (gridDim.y - 1 - blockIdx.y) * DP;
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  sweep_setup(a, b, d_block, DP, tid, NT, s_H, s_kt, s_ktd);

  const int HW = a.H * a.W;
  const int x = tile_x * TW + tid % TW, y = tile_y * TH + tid / TW;
  const bool live = x < a.W && y < a.H;
  const float xf = static_cast<float>(x), yf = static_cast<float>(y);
  const int pix = live ? y * a.W + x : 0;
  const int planes = min(DP, a.D - d_block);
  const float* ref = a.image1 + static_cast<size_t>(b) * a.C * HW + pix;

  float fused[DP];
#pragma unroll
  for (int j = 0; j < DP; ++j) fused[j] = 0.0f;

  for (int m = 0; m < a.M; ++m) {
    const float* Hm = s_H + m * 9;
    const float* ktd_m = s_ktd + m * DP * 3;
    const float* meas = a.image2[m] + static_cast<size_t>(b) * a.C * HW;
    float acc[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) acc[j] = 0.0f;

    // The planes of this workgroup are processed in segments [seg_lo, seg_hi): normally one segment with all of them.
    // When their common footprint does not fit the LDS budget (strong parallax: near planes under forward motion) the
    // segment is halved, down to kMinStagedPlanes (4, which fits for 97 % of the workgroups on the reference's sample
    // scene); a segment that still does not fit has no locality worth staging and takes the global-memory path with all
    // of its planes in flight (16 gathers per channel, the generic kernel's structure).
    constexpr int kMinStagedPlanes = DP < 4 ? DP : 4;
    int seg_hint = DP;   // planes per segment that fitted last time: parallax per plane is uniform along the sweep
    int seg_lo = 0;
    while (seg_lo < planes) {
      int seg_len = min(planes - seg_lo, seg_hint);
      int state;
      int base[DP];
      float w_nw[DP], w_ne[DP], w_sw[DP], w_se[DP];
      for (;;) {
        publish_sample_box<TW, TH, DP, CAP>(a, Hm, ktd_m, tile_x, tile_y, seg_lo, seg_lo + seg_len - 1, tid, s_box);
        state = s_box[4];
        // this thread's taps: box-relative base offsets and weights, checked against the box
        int violation = 0;
        if (state == 1) {
          const int x_lo = s_box[0], y_lo = s_box[1], RW = s_box[2], RH = s_box[3];
#pragma unroll
          for (int j = 0; j < DP; ++j) {
            base[j] = 0;
            w_nw[j] = w_ne[j] = w_sw[j] = w_se[j] = 0.0f;
            if (j >= seg_lo && j < seg_lo + seg_len && live) {
              float ix, iy;
              sweep_position(Hm, ktd_m + j * 3, xf, yf, a.W, a.H, &ix, &iy);
              const BilinearTaps t = make_taps(ix, iy, a.W, a.H);
              // taps entirely outside [-1, W] x [-1, H] see only zeros; everything else must lie inside the staged box
              const bool dead = (t.x0 < -1) || (t.x0 > a.W - 1) || (t.y0 < -1) || (t.y0 > a.H - 1);
              const int rx = t.x0 - x_lo, ry = t.y0 - y_lo;
              const bool inside = (rx >= 0) && (rx + 1 < RW) && (ry >= 0) && (ry + 1 < RH);
              if (!dead && !inside) violation = 1;
              if (!dead && inside) {
                base[j] = ry * RW + rx;
                w_nw[j] = t.w_nw; w_ne[j] = t.w_ne; w_sw[j] = t.w_sw; w_se[j] = t.w_se;
              }
            }
          }
        }
        // (the barrier inside __syncthreads_or also orders this round's s_box reads before the next round's write)
        if (__syncthreads_or(violation)) state = 0;
        if (state != 0 || seg_len <= kMinStagedPlanes) break;
        seg_len = max((seg_len + 1) / 2, kMinStagedPlanes);
      }
      seg_hint = max(seg_len, kMinStagedPlanes);
      const int seg_hi = seg_lo + seg_len;

      if (state == 1) {
        const int x_lo = s_box[0], y_lo = s_box[1], RW = s_box[2], RH = s_box[3];
        const int RS = RW * RH;
        // staging plan.  NCHW: each thread copies up to kSlots box positions per pass (CCH dword loads each, transposed
        // into the record).  NHWC: each thread copies up to kItems 16-byte pieces (position, channel quad) per pass.
        constexpr int kItems = (CAP * (CCH / 4) + NT - 1) / NT;
        constexpr int kPlan = NHWC ? kItems : Cfg::kSlots;
        int goff[kPlan];   // element offset into the measurement map, -1 = outside the image (zero apron) or past the box
#pragma unroll
        for (int k = 0; k < kPlan; ++k) {
          const int item = tid + k * NT;
          const int r = NHWC ? item / (CCH / 4) : item;
          const int ry = r / RW, rx = r - ry * RW;
          const int gx = x_lo + rx, gy = y_lo + ry;
          const bool in = (r < RS) && (gx >= 0) && (gx < a.W) && (gy >= 0) && (gy < a.H);
          goff[k] = in ? (NHWC ? (gy * a.W + gx) * a.C + (item % (CCH / 4)) * 4 : gy * a.W + gx) : -1;
        }
        for (int c0 = 0; c0 < a.C; c0 += CCH) {
          const int nch = min(CCH, a.C - c0);
          // reference features of this pass: issued before the staging loads so their latency overlaps the copy
          float rv[CCH];
#pragma unroll
          for (int c = 0; c < CCH; ++c) rv[c] = (c < nch) ? ref[static_cast<size_t>(c0 + c) * HW] : 0.0f;
          if (NHWC) {
            // kStageBatch 16-byte loads in flight, then their ds_write_b128s (channels beyond C are never read: rv == 0)
            constexpr int kStageBatch = 4;
#pragma unroll
            for (int k0 = 0; k0 < kPlan; k0 += kStageBatch) {
              float4v v[kStageBatch];
#pragma unroll
              for (int kk = 0; kk < kStageBatch; ++kk) {
                const int k = k0 + kk;
                v[kk] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
                if (k < kPlan) {
                  const int item = tid + k * NT;
                  if (goff[k < kPlan ? k : 0] >= 0 && c0 + (item % (CCH / 4)) * 4 < a.C)
                    v[kk] = *reinterpret_cast<const float4v*>(meas + goff[k < kPlan ? k : 0] + c0);
                }
              }
#pragma unroll
              for (int kk = 0; kk < kStageBatch; ++kk) {
                const int item = tid + (k0 + kk) * NT;
                if (k0 + kk < kPlan && item < RS * (CCH / 4))
                  *reinterpret_cast<float4v*>(s_tile + (item / (CCH / 4)) * REC + (item % (CCH / 4)) * 4) = v[kk];
              }
            }
          } else {
            // global (NCHW, coalesced along x) -> registers -> LDS records (transposed): all loads of a position are in
            // flight before its CCH/4 ds_write_b128
#pragma unroll
            for (int k = 0; k < kPlan; ++k) {
              const int r = tid + k * NT;
              float4v v[CCH / 4];
#pragma unroll
              for (int c = 0; c < CCH; ++c) {
                const float* plane = meas + static_cast<size_t>(c0 + min(c, nch - 1)) * HW;
                v[c / 4][c % 4] = (goff[k] >= 0 && c < nch) ? plane[goff[k]] : 0.0f;
              }
              if (r < RS) {
#pragma unroll
                for (int q = 0; q < CCH / 4; ++q) *reinterpret_cast<float4v*>(s_tile + r * REC + q * 4) = v[q];
              }
            }
          }
          __syncthreads();
          if (live) {
#pragma unroll
            for (int j = 0; j < DP; ++j) {
              if (j >= seg_lo && j < seg_hi) {   // workgroup-uniform
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
          }
          __syncthreads();
        }
      } else if (state == 0 && a.spill != nullptr) {
        // Footprint cannot be staged and the caller provided a spill list: hand the segment to the second pass
        // (cost_volume_spill_kernel), which spreads such segments over the whole chip instead of leaving a few
        // workgroups with a long tail.  This workgroup contributes nothing for these planes of frame m.
        if (tid == 0) {
          const unsigned int slot = atomicAdd(a.spill, 1u);
          a.spill[4 + 2 * slot] = (static_cast<unsigned int>(b) << 16) | static_cast<unsigned int>(blockIdx.x);
          a.spill[5 + 2 * slot] = (static_cast<unsigned int>(d_block / DP) << 16) | (static_cast<unsigned int>(m) << 10) |
                                  (static_cast<unsigned int>(seg_lo) << 5) | static_cast<unsigned int>(seg_len);
        }
      } else if (state == 0 && live) {
        // Footprint cannot be staged: taps straight from global memory.  The gathers are scattered (a different cache
        // line per lane and channel), so what matters is memory-level parallelism: two planes x four taps x eight
        // channels = 64 independent loads are issued before the first use, which cuts the dependent-latency chain of a
        // segment to (planes / 2) * (C / 8) steps.
        constexpr int kPair = 2, kChan = 4;
        for (int j0 = seg_lo; j0 < seg_hi; j0 += kPair) {
          int off[kPair][4];
          float wgt[kPair][4], part[kPair];
#pragma unroll
          for (int u = 0; u < kPair; ++u) {
            const int j = min(j0 + u, seg_hi - 1);
            float ix, iy;
            sweep_position(Hm, ktd_m + j * 3, xf, yf, a.W, a.H, &ix, &iy);
            const BilinearTaps t = make_taps(ix, iy, a.W, a.H);
            const int xa = t.in_x0 ? t.x0 : 0, xb = t.in_x1 ? t.x0 + 1 : 0;
            const int ya = t.in_y0 ? t.y0 : 0, yb = t.in_y1 ? t.y0 + 1 : 0;
            const int es = NHWC ? a.C : 1;   // elements per pixel step
            off[u][0] = (ya * a.W + xa) * es; off[u][1] = (ya * a.W + xb) * es;
            off[u][2] = (yb * a.W + xa) * es; off[u][3] = (yb * a.W + xb) * es;
            const bool on = j0 + u < seg_hi;
            wgt[u][0] = (on && t.in_x0 && t.in_y0) ? t.w_nw : 0.0f;
            wgt[u][1] = (on && t.in_x1 && t.in_y0) ? t.w_ne : 0.0f;
            wgt[u][2] = (on && t.in_x0 && t.in_y1) ? t.w_sw : 0.0f;
            wgt[u][3] = (on && t.in_x1 && t.in_y1) ? t.w_se : 0.0f;
            part[u] = 0.0f;
          }
          // under strong magnification most pixels sample outside the image: a wave whose 64 pixels are all dead for
          // this plane pair skips its channel loop (wave-uniform branch)
          float any_w = 0.0f;
#pragma unroll
          for (int u = 0; u < kPair; ++u) any_w += wgt[u][0] + wgt[u][1] + wgt[u][2] + wgt[u][3];
          if (!__any(any_w != 0.0f)) {
            // contributes zeros; acc[] entries stay 0
          } else if (NHWC) {
            // one 16-byte load per (plane, tap, channel quad): 8 loads (32 values) in flight per step; the eight quads of
            // a tap share one 128-byte line, so only the first step of a plane pair misses
            for (int cq = 0; cq < a.C; cq += 4) {
              float4v v[kPair][4];
              float r[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) r[e] = ref[static_cast<size_t>(cq + e) * HW];
#pragma unroll
              for (int u = 0; u < kPair; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) v[u][t] = *reinterpret_cast<const float4v*>(meas + off[u][t] + cq);
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < kPair; ++u) {
                  float t = v[u][0][e] * wgt[u][0];
                  t += v[u][1][e] * wgt[u][1];
                  t += v[u][2][e] * wgt[u][2];
                  t += v[u][3][e] * wgt[u][3];
                  part[u] += r[e] * t;
                }
            }
          } else {
            for (int c0 = 0; c0 < a.C; c0 += kChan) {
              float v[kChan][kPair][4], r[kChan];
#pragma unroll
              for (int cc = 0; cc < kChan; ++cc) {
                const int c = min(c0 + cc, a.C - 1);
                const float* plane = meas + static_cast<size_t>(c) * HW;
                r[cc] = (c0 + cc < a.C) ? ref[static_cast<size_t>(c) * HW] : 0.0f;
#pragma unroll
                for (int u = 0; u < kPair; ++u)
#pragma unroll
                  for (int t = 0; t < 4; ++t) v[cc][u][t] = plane[off[u][t]];
              }
#pragma unroll
              for (int cc = 0; cc < kChan; ++cc)
#pragma unroll
                for (int u = 0; u < kPair; ++u) {
                  float t = v[cc][u][0] * wgt[u][0];
                  t += v[cc][u][1] * wgt[u][1];
                  t += v[cc][u][2] * wgt[u][2];
                  t += v[cc][u][3] * wgt[u][3];
                  part[u] += r[cc] * t;
                }
            }
          }
#pragma unroll
          for (int jj = 0; jj < DP; ++jj)   // compile-time indexed select keeps acc[] in registers
#pragma unroll
            for (int u = 0; u < kPair; ++u)
              if (jj == j0 + u && jj < seg_hi) acc[jj] = part[u];
        }
      }
      // state == 2: the whole footprint of the segment lies outside the image -> zeros
      seg_lo = seg_hi;
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

template <int TW, int TH, int DP, int CCH, int CAP, bool NHWC>
int launch_cost_volume_tiled_layout(const CostVolumeArgs& a, hipStream_t stream) {
  using Cfg = TiledConfig<TW, TH, DP, CCH, CAP>;
  auto kernel = cost_volume_tiled_kernel<TW, TH, DP, CCH, CAP, NHWC>;
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

template <int TW, int TH, int DP>
__global__ void cost_volume_spill_kernel(CostVolumeArgs a);

template <int TW, int TH, int DP, int CCH, int CAP>
int launch_cost_volume_tiled(const CostVolumeArgs& a, hipStream_t stream) {
  if (a.image2_nhwc) return launch_cost_volume_tiled_layout<TW, TH, DP, CCH, CAP, true>(a, stream);
  const int rc = launch_cost_volume_tiled_layout<TW, TH, DP, CCH, CAP, false>(a, stream);
  if (rc != 0 || a.spill == nullptr) return rc;
  static_assert(DP <= 31 && TW * TH <= 1024, "spill item encoding");
  hipLaunchKernelGGL((cost_volume_spill_kernel<TW, TH, DP>), dim3(1024), dim3(TW * TH), 0, stream, a);
  return launch_status();
}

// One thread per (batch, measurement frame): the matrices above into the caller's workspace, so that the sweep
// kernels (hundreds of workgroups) do not each repeat the fp64 inverse.
__global__ void sweep_setup_kernel(CostVolumeArgs a, float* setup, unsigned int* spill) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && spill != nullptr) spill[0] = 0u;   // empty spill list for the sweep launch that follows on the stream
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

// Second pass of the tiled sweep: the (tile, measurement frame, plane segment) items whose footprint did not fit in LDS.
// One workgroup per item (grid-stride over the list), 256 threads = the tile's pixels, gathers straight from global
// memory with 32 loads in flight, result ADDED to the volume the first pass already wrote (atomicAdd: two measurement
// frames may spill the same pixel and plane).  Arithmetic per sample is the generic kernel's.
template <int TW, int TH, int DP>
__global__ __launch_bounds__(TW* TH) void cost_volume_spill_kernel(CostVolumeArgs a) {
  const unsigned int count = a.spill[0];
  const int tid = threadIdx.x;
  const int HW = a.H * a.W;
  const int tiles_x = (a.W + TW - 1) / TW;
  for (unsigned int it = blockIdx.x; it < count; it += gridDim.x) {
    const unsigned int w0 = a.spill[4 + 2 * it], w1 = a.spill[5 + 2 * it];
    const int b = static_cast<int>(w0 >> 16), tile = static_cast<int>(w0 & 0xffffu);
    const int chunk = static_cast<int>(w1 >> 16), m = static_cast<int>((w1 >> 10) & 0x3fu);
    const int seg_lo = static_cast<int>((w1 >> 5) & 0x1fu), seg_len = static_cast<int>(w1 & 0x1fu);
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int x = tile_x * TW + tid % TW, y = tile_y * TH + tid / TW;
    if (x >= a.W || y >= a.H) continue;
    const float xf = static_cast<float>(x), yf = static_cast<float>(y);
    const int pix = y * a.W + x;
    const float* setup = a.setup + (static_cast<size_t>(b) * a.M + m) * kSetupFloats;   // Hm (9) + kt (3)
    float Hm[9], kt[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Hm[k] = setup[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) kt[k] = setup[9 + k];
    const float* meas = a.image2[m] + static_cast<size_t>(b) * a.C * HW;
    const float* ref = a.image1 + static_cast<size_t>(b) * a.C * HW + pix;
    const float norm = 1.0f / (static_cast<float>(a.C) * static_cast<float>(a.M));
    constexpr int kPair = 2, kChan = 4;
    for (int j0 = seg_lo; j0 < seg_lo + seg_len; j0 += kPair) {
      int off[kPair][4];
      float wgt[kPair][4], part[kPair];
#pragma unroll
      for (int u = 0; u < kPair; ++u) {
        const int j = min(j0 + u, seg_lo + seg_len - 1);
        const float depth = plane_depth(a.inv_depth_base, a.inv_depth_step, chunk * DP + j);
        const float ktd[3] = {kt[0] / depth, kt[1] / depth, kt[2] / depth};   // same expression as sweep_setup's table
        float ix, iy;
        sweep_position(Hm, ktd, xf, yf, a.W, a.H, &ix, &iy);
        const BilinearTaps t = make_taps(ix, iy, a.W, a.H);
        const int xa = t.in_x0 ? t.x0 : 0, xb = t.in_x1 ? t.x0 + 1 : 0;
        const int ya = t.in_y0 ? t.y0 : 0, yb = t.in_y1 ? t.y0 + 1 : 0;
        off[u][0] = ya * a.W + xa; off[u][1] = ya * a.W + xb; off[u][2] = yb * a.W + xa; off[u][3] = yb * a.W + xb;
        const bool on = j0 + u < seg_lo + seg_len;
        wgt[u][0] = (on && t.in_x0 && t.in_y0) ? t.w_nw : 0.0f;
        wgt[u][1] = (on && t.in_x1 && t.in_y0) ? t.w_ne : 0.0f;
        wgt[u][2] = (on && t.in_x0 && t.in_y1) ? t.w_sw : 0.0f;
        wgt[u][3] = (on && t.in_x1 && t.in_y1) ? t.w_se : 0.0f;
        part[u] = 0.0f;
      }
      float any_w = 0.0f;
#pragma unroll
      for (int u = 0; u < kPair; ++u) any_w += wgt[u][0] + wgt[u][1] + wgt[u][2] + wgt[u][3];
      if (!__any(any_w != 0.0f)) continue;   // the whole wave samples outside the image for this plane pair
      for (int c0 = 0; c0 < a.C; c0 += kChan) {
        float v[kChan][kPair][4], r[kChan];
#pragma unroll
        for (int cc = 0; cc < kChan; ++cc) {
          const int c = min(c0 + cc, a.C - 1);
          const float* plane = meas + static_cast<size_t>(c) * HW;
          r[cc] = (c0 + cc < a.C) ? ref[static_cast<size_t>(c) * HW] : 0.0f;
#pragma unroll
          for (int u = 0; u < kPair; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) v[cc][u][t] = plane[off[u][t]];
        }
#pragma unroll
        for (int cc = 0; cc < kChan; ++cc)
#pragma unroll
          for (int u = 0; u < kPair; ++u) {
            float t = v[cc][u][0] * wgt[u][0];
            t += v[cc][u][1] * wgt[u][1];
            t += v[cc][u][2] * wgt[u][2];
            t += v[cc][u][3] * wgt[u][3];
            part[u] += r[cc] * t;
          }
      }
#pragma unroll
      for (int u = 0; u < kPair; ++u)
        if (j0 + u < seg_lo + seg_len && part[u] != 0.0f)
          atomicAdd(a.out + (static_cast<size_t>(b) * a.D + chunk * DP + j0 + u) * HW + pix, part[u] * norm);
    }
  }
}

// sweep_tiled.hip
size_t sweep_spill_words(int B, int M, int H, int W, int D);
int launch_sweep_default(const CostVolumeArgs& a, hipStream_t stream);
int launch_sweep_tuning(int which, const CostVolumeArgs& a, hipStream_t stream);

}  // namespace dvmvs

extern "C" size_t dvmvs_cost_volume_workspace_bytes(int B, int M) {
  if (B <= 0 || M <= 0) return 0;
  return sizeof(float) * static_cast<size_t>(B) * M * dvmvs::kSetupFloats;
}

// Workspace that additionally holds the spill list of the two-pass tiled sweep: set-up block (rounded to 16 bytes),
// 4 header words, then two words per possible item (every (batch, tile, plane chunk, frame) may spill up to 8 segments).
extern "C" size_t dvmvs_cost_volume_workspace_bytes_two_pass(int B, int M, int H, int W, int D) {
  if (B <= 0 || M <= 0 || H <= 0 || W <= 0 || D <= 0) return 0;
  const size_t setup = (dvmvs_cost_volume_workspace_bytes(B, M) + 15) / 16 * 16;
  const size_t tiles = static_cast<size_t>((W + 31) / 32) * ((H + 7) / 8);
  const size_t items = static_cast<size_t>(B) * tiles * ((D + 7) / 8) * M * 8;
  const size_t legacy_words = 4 + 2 * items;
  const size_t words = dvmvs::sweep_spill_words(B, M, H, W, D);
  return setup + sizeof(unsigned int) * (words > legacy_words ? words : legacy_words);
}

extern "C" int dvmvs_cost_volume_fwd(const float* image1, const float* const* image2s, const float* pose1,
                                     const float* const* pose2s, const float* K, float* cost_volume,
                                     int B, int M, int C, int H, int W, int D,
                                     double min_depth, double max_depth, int dot_product, int variant, int image2_layout,
                                     float* workspace, size_t workspace_bytes, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (image2_layout != DVMVS_LAYOUT_NCHW && image2_layout != DVMVS_LAYOUT_NHWC) return DVMVS_EINVAL;
  if (variant < 0 || (variant > 3 && variant < 16) || variant > 63) return DVMVS_EINVAL;
  if ((variant == 2 || variant == 3) && !dot_product) return DVMVS_EUNSUPPORTED;
  CostVolumeArgs a;
  const int rc = fill_sweep_args(&a, image1, image2s, pose1, pose2s, K, cost_volume, B, M, C, H, W, D, min_depth, max_depth, true);
  if (rc != 0) return rc;
  a.image2_nhwc = image2_layout == DVMVS_LAYOUT_NHWC ? 1 : 0;
  // channels-last measurement maps are understood by the LDS-tiled dot-product kernel only (16-byte channel quads)
  if (a.image2_nhwc && (!dot_product || variant == 1 || C % 4 != 0 || H * W < 64 * 64)) return DVMVS_EUNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (workspace != nullptr) {
    if (workspace_bytes < dvmvs_cost_volume_workspace_bytes(B, M)) return DVMVS_EINVAL;
    unsigned int* spill = nullptr;
    // a workspace large enough for the spill list switches the tiled sweep to its two-pass form (NCHW maps, tiles indexable
    // in 16 bits, batch < 65536)
    const bool legacy = variant == 3 || (variant >= 16 && variant < 32);
    if (workspace_bytes >= dvmvs_cost_volume_workspace_bytes_two_pass(B, M, H, W, D) &&
        (!legacy || (!a.image2_nhwc && static_cast<size_t>((W + 31) / 32) * ((H + 7) / 8) <= 65535)))
      spill = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(workspace) + (dvmvs_cost_volume_workspace_bytes(B, M) + 15) / 16 * 16);
    hipLaunchKernelGGL(sweep_setup_kernel, dim3((B * M + 63) / 64), dim3(64), 0, s, a, workspace, spill);
    a.spill = spill;
    const int src = launch_status();
    if (src != 0) return src;
    a.setup = workspace;
  }
  if (variant >= 32) {
    if (!dot_product) return DVMVS_EUNSUPPORTED;
    return launch_sweep_tuning(variant - 32, a, s);
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
  if (variant == 3) return launch_cost_volume_tiled<32, 8, 8, 16, 640>(a, s);   // round-1 kernel, kept for A/B timing
  const bool tiled = dot_product && (variant == 2 || a.image2_nhwc || (variant == 0 && H * W >= 64 * 64));
  if (tiled) return launch_sweep_default(a, s);
  return launch_cost_volume_generic(a, dot_product != 0, s);
}
