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

// grad wrt the measurement features, as a SCATTER (rounds 1-3; since the gather kernel further down the product launches this
// only through the tools-only build, where it is the comparison of tools/cv_bwd_microbench.py, and the plain form below it for
// degenerate image sizes).  Done naively the adjoint of the bilinear gather is four fp32 atomics per (pixel, plane, channel) into
// global memory -- 3.7 G device-scope atomics for one training step at 4 x 7 x 256^2 -- which a multi-XCD part serialises at the
// memory side.  The tiled kernel PRIVATISES the scatter in LDS with the forward kernel's geometry: a workgroup owns a 32x8
// reference tile and DP planes, accumulates into a zero-initialised LDS image of the bounding box of its sample positions with
// ds_add_f32 (record stride CCH+1 floats), walking boxes larger than the image in windows of whole rows, and flushes each element
// once with a single global atomic.  2.67 ms per training call whatever the channel chunk, window size or residency
// (profiles/r03_other_experiments.md): the LDS float atomic itself is the limit, which is why the product gathers instead.
constexpr int kBwdPlaneGroups = 4;   // generic fallback geometry (also used when H*W is tiny)
constexpr int kBwdPPT = 4;

// One (pixel, plane) straight to global memory: 4 taps x C channels of device-scope atomics.
__device__ inline void scatter_sample_global(const CostVolumeArgs& f, const float* Hm, const float* ktd, float xf, float yf, float gj,
                                             const float* ref, float* gmeas, int HW) {
  float ix, iy;
  sweep_position(Hm, ktd, xf, yf, f.W, f.H, &ix, &iy);
  const BilinearTaps t = make_taps(ix, iy, f.W, f.H);
  const bool v0 = t.in_x0 && t.in_y0, v1 = t.in_x1 && t.in_y0, v2 = t.in_x0 && t.in_y1, v3 = t.in_x1 && t.in_y1;
  if (!(v0 || v1 || v2 || v3)) return;
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

#ifdef DVMVS_SWEEP_TUNING
template <int TW, int TH, int DP, int CCH, int CAP>
__global__ __launch_bounds__(TW* TH) void cost_volume_bwd_meas_tiled_kernel(CostVolumeBwdArgs a) {
  constexpr int NT = TW * TH;
  constexpr int REC = CCH + 1;
  constexpr int RPT = (CAP + NT - 1) / NT;   // box positions a thread flushes
  static_assert((CAP * REC) % 4 == 0, "the LDS image is cleared in 16-byte stores");
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
    // box of all the planes of the chunk, no capacity test (state 1: well defined, 2: entirely outside the image, 0: not defined)
    publish_sample_box<TW, TH, DP, (1 << 30)>(f, Hm, ktd_m, tile_x, tile_y, 0, planes - 1, tid, s_box);
    const int x_lo = s_box[0], y_lo = s_box[1], RW = s_box[2], RH = s_box[3];
    const int state = (s_box[4] == 1 && RW > CAP) ? 0 : s_box[4];
    __syncthreads();                            // s_box is rewritten for the next frame

    unsigned direct = 0;                        // planes of this pixel that go straight to global memory
    if (state == 1) {
      int pos[DP];                              // (box row << 16) | box column of the north-west tap
      float w[DP][4];
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        pos[j] = 0;
        w[j][0] = w[j][1] = w[j][2] = w[j][3] = 0.0f;
        if (gd[j] != 0.0f) {                    // implies live && j < planes
          float ix, iy;
          sweep_position(Hm, ktd_m + j * 3, xf, yf, f.W, f.H, &ix, &iy);
          const BilinearTaps t = make_taps(ix, iy, f.W, f.H);
          const bool dead = (t.x0 < -1) || (t.x0 > f.W - 1) || (t.y0 < -1) || (t.y0 > f.H - 1);
          const int rx = t.x0 - x_lo, ry = t.y0 - y_lo;
          const bool inside = (rx >= 0) && (rx + 1 < RW) && (ry >= 0) && (ry + 1 < RH);
          if (!dead && inside) {
            pos[j] = (ry << 16) | rx;
            // taps outside the image land in the apron of the box and are dropped by the flush
            w[j][0] = t.w_nw * gd[j]; w[j][1] = t.w_ne * gd[j]; w[j][2] = t.w_sw * gd[j]; w[j][3] = t.w_se * gd[j];
          } else if (!dead) {
            direct |= 1u << j;
          }
        }
      }

      const int WH = min(RH, CAP / RW);         // rows per window (>= 1: RW <= CAP)
      for (int wy0 = 0; wy0 < RH; wy0 += WH) {
        const int rows = min(WH, RH - wy0);
        // anything of this pixel in rows [wy0, wy0 + rows)?
        bool mine = false;
#pragma unroll
        for (int j = 0; j < DP; ++j) {
          const int ry = (pos[j] >> 16) - wy0;
          const bool any = (w[j][0] != 0.0f) || (w[j][1] != 0.0f) || (w[j][2] != 0.0f) || (w[j][3] != 0.0f);
          mine |= any && (ry + 1 >= 0) && (ry < rows);
        }
        if (!__syncthreads_or(mine ? 1 : 0)) continue;   // also orders the previous window's flush before this one's clear

        const int RS = rows * RW;
        int goff[RPT];                          // image offset of the box positions this thread flushes (-1: none / apron)
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          const int r = tid + k * NT;
          const int ry = r / RW, rx = r - ry * RW;
          const int gx = x_lo + rx, gy = y_lo + wy0 + ry;
          goff[k] = (r < RS && gx >= 0 && gx < f.W && gy >= 0 && gy < f.H) ? gy * f.W + gx : -1;
        }
        for (int c0 = 0; c0 < f.C; c0 += CCH) {
          const int nch = min(CCH, f.C - c0);
          {
            float4* z = reinterpret_cast<float4*>(s_acc);
            const int n4 = (RS * REC + 3) >> 2;
            for (int i = tid; i < n4; i += NT) z[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          }
          float rv[CCH];
#pragma unroll
          for (int c = 0; c < CCH; ++c) rv[c] = (mine && c < nch) ? ref[static_cast<size_t>(c0 + c) * HW] : 0.0f;
          __syncthreads();
          if (mine) {
#pragma unroll
            for (int j = 0; j < DP; ++j) {
              const int ry = (pos[j] >> 16) - wy0, rx = pos[j] & 0xffff;
              const bool any = (w[j][0] != 0.0f) || (w[j][1] != 0.0f) || (w[j][2] != 0.0f) || (w[j][3] != 0.0f);
              if (any && (ry + 1 >= 0) && (ry < rows)) {
                float* r0 = s_acc + (ry * RW + rx) * REC;
                if (ry >= 0) {
#pragma unroll
                  for (int c = 0; c < CCH; ++c) {
                    atomicAdd(r0 + c, rv[c] * w[j][0]);
                    atomicAdd(r0 + REC + c, rv[c] * w[j][1]);
                  }
                }
                if (ry + 1 < rows) {
                  float* r1 = r0 + RW * REC;
#pragma unroll
                  for (int c = 0; c < CCH; ++c) {
                    atomicAdd(r1 + c, rv[c] * w[j][2]);
                    atomicAdd(r1 + REC + c, rv[c] * w[j][3]);
                  }
                }
              }
            }
          }
          __syncthreads();
          // flush: for one channel a wave's lanes hit consecutive x of one gradient plane; LDS reads are REC words apart (odd)
#pragma unroll
          for (int k = 0; k < RPT; ++k) {
            if (goff[k] >= 0) {
              const float* rec = s_acc + (tid + k * NT) * REC;
              float* dst = gmeas + static_cast<size_t>(c0) * HW + goff[k];
#pragma unroll
              for (int c = 0; c < CCH; ++c) {
                if (c < nch) {
                  const float v = rec[c];
                  if (v != 0.0f) atomicAdd(dst + static_cast<size_t>(c) * HW, v);
                }
              }
            }
          }
          __syncthreads();
        }
      }
    } else if (state == 0) {
#pragma unroll
      for (int j = 0; j < DP; ++j)
        if (gd[j] != 0.0f) direct |= 1u << j;
    }

    // the rare samples that have no place in the box
    while (direct) {
      const int j = __ffs(direct) - 1;
      direct &= direct - 1;
      float gj = 0.0f;
#pragma unroll
      for (int jj = 0; jj < DP; ++jj)
        if (jj == j) gj = gd[jj];
      scatter_sample_global(f, Hm, ktd_m + j * 3, xf, yf, gj, ref, gmeas, HW);
    }
  }
}

#endif  // DVMVS_SWEEP_TUNING

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

// grad wrt the measurement features as a GATHER (no atomics, deterministic).  A thread owns one measurement pixel q and walks
// the planes; on plane d the sample map p -> pos_d(p) is a homography, so the reference pixels whose 2x2 tap footprint
// contains q are the integer points of pos_d^-1((q-1, q+1)^2): a convex quadrilateral whose corners are four evaluations of
// the inverse homography.  The kernel visits the integer points of that quadrilateral's bounding box (a 3x3 block at unit
// scale), evaluates for each of them the FORWARD position with exactly the forward kernel's arithmetic (sweep_position: same
// floor, same weights as the reference's taps), keeps those that really have a tap on q and accumulates
//   g[d,p] * f1[c,p] * w_tap      for its 32 channels in registers.
// The inverse is only a search window: where it is not trustworthy (the 2x2 footprint straddles the plane's vanishing line,
// or the plane's matrix is singular) the window is the whole reference image -- slow and exact.
// Workgroup = 64 measurement pixels x 4 plane groups (planes ty, ty+4, ...); the four partial sums meet in LDS in a fixed
// order, so the result is bit-reproducible; every output element is written by exactly one thread (added to the caller's
// zero-filled buffer, as the contract of the scatter kernels has it).
// CH channels per thread, PG plane groups, CG channel groups: a workgroup is 64 pixels x PG x CG waves and covers CH * CG
// channels per pass over the planes.  WAVES = waves per SIMD the register allocation is held to.
template <int CH, int PG, int CG, int WAVES>
__global__ __launch_bounds__(kWave* PG* CG) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void cost_volume_bwd_meas_gather_kernel(CostVolumeBwdArgs a) {
  constexpr int NT = kWave * PG * CG;
  constexpr int CPP = CH * CG;                  // channels per pass
  static_assert(CH % PG == 0, "the final sum is split over the plane groups");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const CostVolumeArgs& f = a.fwd;
  const int m = blockIdx.y, b = blockIdx.z;
  float* gmeas = a.grad_image2[m];
  if (!gmeas) return;                           // workgroup-uniform
  float* s_H = smem;                            // [M][9]
  float* s_kt = s_H + DVMVS_MAX_MEASUREMENTS * 9;
  float* s_ktd = s_kt + DVMVS_MAX_MEASUREMENTS * 3;          // [M][D][3]
  float* s_inv = s_ktd + DVMVS_MAX_MEASUREMENTS * f.D * 3;   // [D][9]  reference pixel ~ s_inv * (qx, qy, 1), frame m
  float* s_red = s_inv + f.D * 9;                            // [PG][CPP][64]
  // blockDim.x is the wave width: threadIdx.y is wave-uniform, and telling the compiler so keeps every base pointer scalar
  const int lane = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.y));
  const int pg = wave % PG, cg = wave / PG;
  const int tid = threadIdx.y * kWave + lane;
  sweep_setup(f, b, 0, f.D, tid, NT, s_H, s_kt, s_ktd);
  const float* Hm = s_H + m * 9;
  for (int d = tid; d < f.D; d += NT) {
    // sample position = diag(sx, sy) * (A p) / (A p)_z with A = Hm + (kt / depth_d) e3^T, sx = (W-1)/W, sy = (H-1)/H
    // (utils.py:66-76 and grid_sample's align_corners convention)  =>  p ~ A^-1 diag(1/sx, 1/sy, 1) q
    const float* ktd = s_ktd + (m * f.D + d) * 3;
    double A[9], inv[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = static_cast<double>(Hm[i]) + ((i % 3 == 2) ? static_cast<double>(ktd[i / 3]) : 0.0);
    inverse3(A, inv);                           // non-finite for a singular A: the window test below then fails
    const double ux = static_cast<double>(f.W) / static_cast<double>(f.W - 1), uy = static_cast<double>(f.H) / static_cast<double>(f.H - 1);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      s_inv[d * 9 + r * 3 + 0] = static_cast<float>(inv[r * 3 + 0] * ux);
      s_inv[d * 9 + r * 3 + 1] = static_cast<float>(inv[r * 3 + 1] * uy);
      s_inv[d * 9 + r * 3 + 2] = static_cast<float>(inv[r * 3 + 2]);
    }
  }
  __syncthreads();

  const int HW = f.H * f.W;
  const int q = blockIdx.x * kWave + lane;
  const bool live = q < HW;
  const int qy = live ? q / f.W : 0, qx = live ? q - qy * f.W : 0;
  const float scale = 1.0f / (static_cast<float>(f.M) * static_cast<float>(f.C));
  const float* g = a.grad_cost + static_cast<size_t>(b) * f.D * HW;
  gmeas += static_cast<size_t>(b) * f.C * HW;

  for (int c0 = 0; c0 < f.C; c0 += CPP) {
    const int cbase = c0 + cg * CH;             // this thread's first channel
    const int nch = max(0, min(CH, f.C - cbase));
    const float* ref = f.image1 + (static_cast<size_t>(b) * f.C + min(cbase, f.C - 1)) * HW;
    float acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0.0f;
    if (live && nch > 0) {
      for (int d = pg; d < f.D; d += PG) {
        const float* iv = s_inv + d * 9;
        const float* ktd = s_ktd + (m * f.D + d) * 3;
        // inverse images of the corners of (q - 1.01, q + 1.01)^2 (a search window: the hardware reciprocal is good enough)
        float lo_x = 3.0e38f, hi_x = -3.0e38f, lo_y = 3.0e38f, hi_y = -3.0e38f, lo_w = 3.0e38f, hi_w = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float tx = static_cast<float>(qx) + ((k & 1) ? 1.01f : -1.01f), tyv = static_cast<float>(qy) + ((k & 2) ? 1.01f : -1.01f);
          const float pw = fmaf(iv[6], tx, fmaf(iv[7], tyv, iv[8]));
          const float rw = __builtin_amdgcn_rcpf(pw);
          const float px = fmaf(iv[0], tx, fmaf(iv[1], tyv, iv[2])) * rw, py = fmaf(iv[3], tx, fmaf(iv[4], tyv, iv[5])) * rw;
          lo_x = fminf(lo_x, px); hi_x = fmaxf(hi_x, px); lo_y = fminf(lo_y, py); hi_y = fmaxf(hi_y, py);
          lo_w = fminf(lo_w, pw); hi_w = fmaxf(hi_w, pw);
        }
        // trustworthy: finite, and the homogeneous coordinate keeps its sign with a margin over the footprint (every comparison
        // is false for NaN, which selects the whole image)
        const bool one_sign = (lo_w > 0.0f && lo_w > 1e-3f * hi_w) || (hi_w < 0.0f && hi_w < 1e-3f * lo_w);
        const bool bounded = (lo_x > -1e7f) && (hi_x < 1e7f) && (lo_y > -1e7f) && (hi_y < 1e7f);
        int x0 = 0, x1 = f.W - 1, y0 = 0, y1 = f.H - 1;
        if (one_sign && bounded) {
          // integer points of [lo - margin, hi + margin]
          x0 = max(0, static_cast<int>(ceilf(lo_x - 0.05f)));
          x1 = min(f.W - 1, static_cast<int>(floorf(hi_x + 0.05f)));
          y0 = max(0, static_cast<int>(ceilf(lo_y - 0.05f)));
          y1 = min(f.H - 1, static_cast<int>(floorf(hi_y + 0.05f)));
        }
        const float* gplane = g + static_cast<size_t>(d) * HW;
        for (int py = y0; py <= y1; ++py) {
          for (int px = x0; px <= x1; ++px) {
            float ix, iy;
            sweep_position(Hm, ktd, static_cast<float>(px), static_cast<float>(py), f.W, f.H, &ix, &iy);
            // make_taps' own test and arithmetic (dvmvs_device.h): positions this far out have no tap anywhere
            if (!((ix > -2.0f) && (ix < static_cast<float>(f.W) + 1.0f) && (iy > -2.0f) && (iy < static_cast<float>(f.H) + 1.0f))) continue;
            const float fx = floorf(ix), fy = floorf(iy);
            const int dx = qx - static_cast<int>(fx), dy = qy - static_cast<int>(fy);
            if (dx < 0 || dx > 1 || dy < 0 || dy > 1) continue;
            const float wxv = dx ? ix - fx : (fx + 1.0f) - ix;
            const float wyv = dy ? iy - fy : (fy + 1.0f) - iy;
            const float wgt = wxv * wyv;
            if (wgt == 0.0f) continue;
            // the upstream gradient and the reference features are loaded together (no branch on the former)
            const int p = py * f.W + px;
            const float coef = wgt * (gplane[p] * scale);
#pragma unroll
            for (int c = 0; c < CH; ++c)
              if (c < nch) acc[c] = fmaf(ref[static_cast<size_t>(c) * HW + p], coef, acc[c]);
          }
        }
      }
    }
    // the plane groups meet in a fixed order
    __syncthreads();                            // previous pass's readers are done
#pragma unroll
    for (int c = 0; c < CH; ++c) s_red[(pg * CPP + cg * CH + c) * kWave + lane] = acc[c];
    __syncthreads();
    if (live) {
      constexpr int kPer = CH / PG;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const int cl = cg * CH + pg * kPer + k;  // channel within the pass
        if (c0 + cl < f.C) {
          float v = 0.0f;
#pragma unroll
          for (int gidx = 0; gidx < PG; ++gidx) v += s_red[(gidx * CPP + cl) * kWave + lane];
          float* dst = gmeas + static_cast<size_t>(c0 + cl) * HW + q;
          *dst += v;
        }
      }
    }
  }
}

template <int CH, int PG, int CG, int WAVES>
static int launch_bwd_meas_gather(const CostVolumeBwdArgs& a, int B, int M, int H, int W, int D, hipStream_t s) {
  auto lds_floats = [](int planes) {
    return static_cast<size_t>(DVMVS_MAX_MEASUREMENTS) * 12 + static_cast<size_t>(DVMVS_MAX_MEASUREMENTS) * planes * 3 + static_cast<size_t>(planes) * 9 +
           static_cast<size_t>(PG) * CH * CG * kWave;
  };
  auto kernel = cost_volume_bwd_meas_gather_kernel<CH, PG, CG, WAVES>;
  static bool configured[64] = {};
  int device = 0;
  DVMVS_RETURN_IF_HIP(hipGetDevice(&device));
  const bool tracked = device >= 0 && device < 64;
  if (!tracked || !configured[device]) {
    DVMVS_RETURN_IF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            static_cast<int>(sizeof(float) * lds_floats(DVMVS_MAX_DEPTH_LEVELS))));
    if (tracked) configured[device] = true;
  }
  const int HW = H * W;
  dim3 block(kWave, PG * CG), grid((HW + kWave - 1) / kWave, M, B);
  hipLaunchKernelGGL(kernel, grid, block, sizeof(float) * lds_floats(D), s, a);
  return launch_status();
}

#ifdef DVMVS_SWEEP_TUNING
// Launch of the LDS-privatised scatter: 32x8-pixel tiles x 8 planes per workgroup, CCH channels per pass, CAP box positions in LDS.
template <int CCH, int CAP>
static int launch_bwd_meas_tiled(const CostVolumeBwdArgs& a, int B, int H, int W, int D, hipStream_t s) {
  constexpr int TW = 32, TH = 8, DP = 8;
  constexpr size_t kLds = sizeof(float) * CAP * (CCH + 1);
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
  return launch_status();
}

static int g_bwd_tuning_config = -1;   // -1: what the product does
#endif

}  // namespace dvmvs

#ifdef DVMVS_SWEEP_TUNING
// tools-only library (make tuning): which instantiation dvmvs_cost_volume_bwd launches (tools/cv_bwd_microbench.py)
extern "C" void dvmvs_tuning_set_bwd_config(int config) { dvmvs::g_bwd_tuning_config = config; }
#endif

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
  if (!any_meas) return rc;
  int config = (W > 1 && H > 1) ? 5 : 9;        // the gather needs (W-1)/W and (H-1)/H to be invertible
#ifdef DVMVS_SWEEP_TUNING
  if (g_bwd_tuning_config >= 0 && (config == 5 || g_bwd_tuning_config == 9)) config = g_bwd_tuning_config;
  switch (config) {
    case 0: return launch_bwd_meas_tiled<16, 768>(a, B, H, W, D, s);    // LDS-privatised scatter, 51 KB: three workgroups per CU
    case 1: return launch_bwd_meas_tiled<8, 1536>(a, B, H, W, D, s);    // 4 channel passes, 54 KB
    case 2: return launch_bwd_meas_tiled<32, 384>(a, B, H, W, D, s);    // 1 channel pass, 50 KB
    case 3: return launch_bwd_meas_tiled<16, 1152>(a, B, H, W, D, s);   // 77 KB: two workgroups per CU
    case 4: return launch_bwd_meas_tiled<16, 576>(a, B, H, W, D, s);    // 38 KB
    // the gather with other splits of a workgroup (channels per thread, plane groups, channel groups, waves per SIMD); measured
    // 824 / 684 / 1436 / 1481 / 913 us against the product's 668 us on the training geometry (profiles/r03_bwd_microbench.txt)
    case 6: return launch_bwd_meas_gather<16, 4, 2, 5>(a, B, M, H, W, D, s);
    case 7: return launch_bwd_meas_gather<8, 4, 4, 8>(a, B, M, H, W, D, s);
    case 8: return launch_bwd_meas_gather<32, 4, 1, 3>(a, B, M, H, W, D, s);
    case 10: return launch_bwd_meas_gather<32, 4, 1, 4>(a, B, M, H, W, D, s);
    case 11: return launch_bwd_meas_gather<32, 8, 1, 4>(a, B, M, H, W, D, s);
    default: break;
  }
#endif
  // 16 channels per thread, 8 plane groups x 2 channel groups = 16 waves per 64 pixels: the kernel is bound by the length of a
  // wave's chain of dependent gathers, so the planes are spread over as many waves as a workgroup holds
  if (config == 5) return launch_bwd_meas_gather<16, 8, 2, 5>(a, B, M, H, W, D, s);
  constexpr int kPlanesPerBlock = kBwdPlaneGroups * kBwdPPT;     // the plain global-atomic scatter
  dim3 block(kWave, kBwdPlaneGroups), grid((HW + kWave - 1) / kWave, (D + kPlanesPerBlock - 1) / kPlanesPerBlock, B);
  hipLaunchKernelGGL(cost_volume_bwd_meas_kernel, grid, block, 0, s, a);
  return launch_status();
}
