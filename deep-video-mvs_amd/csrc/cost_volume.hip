// Fused plane-sweep warp + feature correlation (forward) for gfx950.
//
// One launch produces the whole [B,D,H,W] cost volume for all M measurement frames: the per-plane homography,
// the bilinear gather of the measurement features, the channel reduction and the mean over measurement frames
// are fused, so the volume is written exactly once and no warped temporary ever exists in HBM.
// Semantics: /root/reference/dvmvs/utils.py:45-107 (see oracle/dvmvs_oracle.py for the CPU restatement).
#include "plane_sweep.h"

// coefficients of dvmvs::sweep_model_us (us), least squares over the 285 keyframe pairs of the sample scene, launches with the
// host-planned work list (tools/sweep_select_fit.py on tools/cv_microbench.py --lines all --variants 2,3 --work-list;
// profiles/r04_sweep_select_fit.md): rms residual 3.2 us (default) / 2.4 us (wide)
#define SWEEP_MODEL_DEFAULT {22.7412, 1.2906, 10.4202, 0.8341, 0.02711, 2.140e-05}
#define SWEEP_MODEL_WIDE {24.9001, 3.8563, 8.7854, 1.1609, 0.02185, 1.917e-05}

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


// sweep_tiled.hip
size_t sweep_spill_words(int B, int M, int H, int W, int D);
int launch_sweep_default(const CostVolumeArgs& a, hipStream_t stream);
int launch_sweep_wide(const CostVolumeArgs& a, hipStream_t stream);
size_t sweep_work_list_words(int B, int H, int W, int D);
int sweep_work_list_host(int configuration, const float* Hm, const float* kt, int B, int M, int H, int W, int D, double inv_base, double inv_step,
                         unsigned int* items, size_t capacity_words, long long* stats);
void sweep_plan_stats_host(int configuration, const float* Hm, const float* kt, int B, int M, int H, int W, int D, double inv_base, double inv_step,
                           long long* stats);
int launch_sweep_tuning(int which, const CostVolumeArgs& a, hipStream_t stream);
// sweep_mfma.hip
bool sweep_mfma_supports(const CostVolumeArgs& a);
int launch_sweep_mfma(const CostVolumeArgs& a, hipStream_t stream, bool allow_persistent);
int launch_sweep_mfma_tuning(int which, const CostVolumeArgs& a, hipStream_t stream);
void sweep_mfma_estimate_host(const float* Hm, const float* kt, int M, int H, int W, int D, double inv_base, double inv_step, double* stats);

// Predicted duration (us) of the sweep + second pass in one configuration from its plan statistics (dvmvs_sweep_plan_stats):
// base + staged runs of the longest work item (the work list cuts chains to <= 3) + a fixed cost when the second pass is not empty
// (its chain of dependent loads and gathers: ~10 us however few units) + queued planes of the worst workgroup + all queued planes + all staged
// records.  Fitted on the 128x160x64 shape; it only ranks the two configurations, so other shapes reuse it as it is.
inline double sweep_model_us(int configuration, const long long* st, int B, int H, int W, int D) {
  static const double kCoef[2][6] = {SWEEP_MODEL_DEFAULT, SWEEP_MODEL_WIDE};
  const double* c = kCoef[configuration];
  (void)B; (void)H; (void)W; (void)D;
  const double longest = st[6] < 3 ? static_cast<double>(st[6]) : 3.0;
  return c[0] + c[1] * longest + (st[4] > 0 ? c[2] : 0.0) + c[3] * static_cast<double>(st[7]) + c[4] * static_cast<double>(st[4]) +
         c[5] * static_cast<double>(st[1]);
}

}  // namespace dvmvs

// Workspace of the two-pass LDS-tiled sweep: spill header (4 words), group list, one slot per workgroup (sweep_tiled.hip).
extern "C" size_t dvmvs_cost_volume_workspace_bytes(int B, int M, int H, int W, int D) {
  if (B <= 0 || M <= 0 || H <= 0 || W <= 0 || D <= 0) return 0;
  return sizeof(unsigned int) * dvmvs::sweep_spill_words(B, M, H, W, D);
}

extern "C" int dvmvs_cost_volume_fwd(const float* image1, const float* const* image2s, const float* Hm, const float* kt,
                                     float* cost_volume, int B, int M, int C, int H, int W, int D,
                                     double min_depth, double max_depth, int dot_product, int variant, int image2_layout,
                                     float* workspace, size_t workspace_bytes, dvmvs_stream_t stream) {
  return dvmvs_cost_volume_planned_fwd(image1, image2s, Hm, kt, cost_volume, B, M, C, H, W, D, min_depth, max_depth, dot_product, variant,
                                       image2_layout, workspace, workspace_bytes, nullptr, stream);
}

extern "C" size_t dvmvs_sweep_work_list_bytes(int B, int H, int W, int D) {
  if (B <= 0 || H <= 0 || W <= 0 || D <= 0) return 0;
  return sizeof(unsigned int) * dvmvs::sweep_work_list_words(B, H, W, D);
}

extern "C" int dvmvs_sweep_work_list(const float* Hm_host, const float* kt_host, int B, int M, int H, int W, int D,
                                     double min_depth, double max_depth, int configuration, unsigned int* work_list_host, size_t work_list_bytes) {
  if (!Hm_host || !kt_host || !work_list_host || B <= 0 || M <= 0 || H <= 0 || W <= 0 || D <= 0) return DVMVS_EINVAL;
  if (M > DVMVS_MAX_MEASUREMENTS || D > DVMVS_MAX_DEPTH_LEVELS) return DVMVS_EUNSUPPORTED;
  if (!(min_depth > 0.0) || !(max_depth > 0.0) || (configuration != 0 && configuration != 1)) return DVMVS_EINVAL;
  const double inv_base = 1.0 / max_depth, inv_step = D > 1 ? (1.0 / min_depth - 1.0 / max_depth) / (D - 1) : 0.0;
  return dvmvs::sweep_work_list_host(configuration, Hm_host, kt_host, B, M, H, W, D, inv_base, inv_step, work_list_host, work_list_bytes / sizeof(unsigned int),
                                     nullptr);
}

extern "C" int dvmvs_cost_volume_planned_fwd(const float* image1, const float* const* image2s, const float* Hm, const float* kt,
                                             float* cost_volume, int B, int M, int C, int H, int W, int D,
                                             double min_depth, double max_depth, int dot_product, int variant, int image2_layout,
                                             float* workspace, size_t workspace_bytes, const unsigned int* work_list, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (image2_layout != DVMVS_LAYOUT_NCHW && image2_layout != DVMVS_LAYOUT_NHWC) return DVMVS_EINVAL;
  if (variant < 0 || (variant > 7 && variant < 32) || variant > 255) return DVMVS_EINVAL;
  if (variant >= 2 && variant <= 7 && !dot_product) return DVMVS_EUNSUPPORTED;
  const bool single_pass = variant == 4 || variant == 5;      // no second launch: the sweep gathers an unstageable run inline
  if (single_pass) variant -= 2;
  CostVolumeArgs a;
  const int rc = fill_sweep_args(&a, image1, image2s, Hm, kt, cost_volume, B, M, C, H, W, D, min_depth, max_depth, true);
  if (rc != 0) return rc;
  a.image2_nhwc = image2_layout == DVMVS_LAYOUT_NHWC ? 1 : 0;
  // channels-last measurement maps are understood by the LDS-tiled dot-product kernel only (16-byte channel quads)
  if (a.image2_nhwc && (!dot_product || variant == 1 || C % 4 != 0 || (H * W < 64 * 64 && variant != 6 && variant != 7))) return DVMVS_EUNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // a workspace large enough for the spill list switches the tiled sweep to its two-pass form
  if (!single_pass && workspace != nullptr && workspace_bytes >= dvmvs_cost_volume_workspace_bytes(B, M, H, W, D))
    a.spill = reinterpret_cast<unsigned int*>(workspace);
  if (variant == 6 || variant == 7) {
    // correlate-then-interpolate sweep on the fp32 matrix cores (sweep_mfma.hip): up to 32 channels, either layout of the measurement
    // maps, any image size, no workspace, no work list; 7 = its one-item-per-workgroup form also where the persistent form is eligible
    if (!sweep_mfma_supports(a)) return DVMVS_EUNSUPPORTED;
    return launch_sweep_mfma(a, s, variant == 6);
  }
  if ((variant >= 96 && variant < 128) || variant >= 224) {      // tuning configurations of the MFMA sweep: rounds 5 (96 + k) and 6 (224 + k - 32)
    if (!dot_product) return DVMVS_EUNSUPPORTED;
    return launch_sweep_mfma_tuning(variant >= 224 ? variant - 224 + 32 : variant - 96, a, s);
  }
  if (variant >= 32) {
    // tuning configurations for tools/cv_microbench.py; not part of the stable interface
    if (!dot_product) return DVMVS_EUNSUPPORTED;
    if (work_list != nullptr) a.items = work_list;
    return launch_sweep_tuning(variant - 32, a, s);
  }
  // the tiled sweep addresses the maps through 32-bit buffer offsets: one batch item of one map must stay below 2 GiB
  const bool fits = static_cast<long long>(C) * H * W * 4 < (1LL << 31);
  const bool tiled = dot_product && fits && (variant == 2 || variant == 3 || a.image2_nhwc || (variant == 0 && H * W >= 64 * 64));
  if ((variant == 2 || variant == 3) && !fits) return DVMVS_EUNSUPPORTED;
  if (a.image2_nhwc && !fits) return DVMVS_EUNSUPPORTED;
  if (tiled && work_list != nullptr) a.items = work_list;
  if (tiled && variant == 3) return launch_sweep_wide(a, s);
  if (tiled) return launch_sweep_default(a, s);
  return launch_cost_volume_generic(a, dot_product != 0, s);
}

// Host-side model of the LDS-tiled sweep's run plan for HOST copies of the matrices (no HIP call): see sweep_tiled.hip.
extern "C" int dvmvs_sweep_plan_stats(const float* Hm_host, const float* kt_host, int B, int M, int H, int W, int D,
                                      double min_depth, double max_depth, int configuration, long long* stats) {
  if (!Hm_host || !kt_host || !stats || B <= 0 || M <= 0 || H <= 0 || W <= 0 || D <= 0) return DVMVS_EINVAL;
  if (M > DVMVS_MAX_MEASUREMENTS || D > DVMVS_MAX_DEPTH_LEVELS) return DVMVS_EUNSUPPORTED;
  if (!(min_depth > 0.0) || !(max_depth > 0.0) || (configuration != 0 && configuration != 1)) return DVMVS_EINVAL;
  const double inv_base = 1.0 / max_depth, inv_step = D > 1 ? (1.0 / min_depth - 1.0 / max_depth) / (D - 1) : 0.0;
  dvmvs::sweep_plan_stats_host(configuration, Hm_host, kt_host, B, M, H, W, D, inv_base, inv_step, stats);
  return 0;
}

// Which configuration of the LDS-tiled sweep for this keyframe pair.  An easy geometry -- nothing queued for the second pass and no
// workgroup with more than three staged runs in the default configuration, two thirds of the sample scene's pairs -- keeps the default
// one (35 us against 41); otherwise a linear cost model over the plan statistics of both configurations decides, fitted to per-pair
// timings of both on all 285 keyframe pairs of the sample scene (tools/sweep_select_fit.py; profiles/r04_sweep_select_fit.md).
// Deterministic in the matrices (IEEE fp32 / integer arithmetic only).
namespace dvmvs {
inline bool sweep_is_easy(const long long* default_stats) { return default_stats[4] == 0 && default_stats[6] <= 3; }
}

extern "C" int dvmvs_sweep_select_variant(const float* Hm_host, const float* kt_host, int B, int M, int H, int W, int D,
                                          double min_depth, double max_depth) {
  long long d[8], w[8];
  int rc = dvmvs_sweep_plan_stats(Hm_host, kt_host, B, M, H, W, D, min_depth, max_depth, 0, d);
  if (rc != 0) return rc;
  if (dvmvs::sweep_is_easy(d)) return 2;
  rc = dvmvs_sweep_plan_stats(Hm_host, kt_host, B, M, H, W, D, min_depth, max_depth, 1, w);
  if (rc != 0) return rc;
  return dvmvs::sweep_model_us(0, d, B, H, W, D) <= dvmvs::sweep_model_us(1, w, B, H, W, D) ? 2 : 3;
}

// Work estimate of the correlate-then-interpolate sweep (variant 6) for HOST copies of the matrices (batch item 0): see sweep_mfma.hip.
extern "C" int dvmvs_sweep_mfma_estimate(const float* Hm_host, const float* kt_host, int B, int M, int H, int W, int D, double min_depth, double max_depth,
                                         double* stats) {
  if (!Hm_host || !kt_host || !stats || B <= 0 || M <= 0 || H <= 0 || W <= 0 || D <= 0) return DVMVS_EINVAL;
  if (M > DVMVS_MAX_MEASUREMENTS || D > DVMVS_MAX_DEPTH_LEVELS) return DVMVS_EUNSUPPORTED;
  if (!(min_depth > 0.0) || !(max_depth > 0.0)) return DVMVS_EINVAL;
  const double inv_base = 1.0 / max_depth, inv_step = D > 1 ? (1.0 / min_depth - 1.0 / max_depth) / (D - 1) : 0.0;
  dvmvs::sweep_mfma_estimate_host(Hm_host, kt_host, M, H, W, D, inv_base, inv_step, stats);
  return 0;
}

// dvmvs_sweep_select_variant + dvmvs_sweep_work_list in ONE walk over the (tile, chunk) pairs (the per-frame host cost of the sweep
// plan: ~0.15 ms for an easy pair, ~0.5 ms where both configurations have to be planned): decides the configuration (or takes
// `variant` = 2 / 3 as the configuration given; 0 = decide), leaves that configuration's work list in `work_list_host` and returns the
// variant to launch with: 2 / 3 (two passes) or, when the plan queues nothing for the second pass, 4 / 5 (the same configurations as ONE
// launch -- the empty second pass costs 3-4.5 us of every frame it is launched in, five frames of six on the sample scene).
namespace dvmvs {
int sweep_plan_impl(const float* Hm_host, const float* kt_host, int B, int M, int H, int W, int D, double min_depth, double max_depth,
                    int variant, unsigned int* work_list_host, size_t work_list_bytes, bool* easy_out);
}

extern "C" int dvmvs_sweep_plan(const float* Hm_host, const float* kt_host, int B, int M, int H, int W, int D, double min_depth, double max_depth,
                                int variant, unsigned int* work_list_host, size_t work_list_bytes) {
  return dvmvs::sweep_plan_impl(Hm_host, kt_host, B, M, H, W, D, min_depth, max_depth, variant, work_list_host, work_list_bytes, nullptr);
}

extern "C" int dvmvs_sweep_plan6(const float* Hm_host, const float* kt_host, int B, int M, int H, int W, int D, double min_depth, double max_depth,
                                 unsigned int* work_list_host, size_t work_list_bytes) {
  if (!Hm_host || !kt_host || !work_list_host || work_list_bytes < 2 * sizeof(unsigned int)) return DVMVS_EINVAL;
  // Round 6: variant 6 for every single-item launch.  With the persistent form (every SIMD the same mix of work) and gather passes for magnified
  // footprints it is the faster kernel on 255 of the sample scene's 285 keyframe pairs, 34.3 us mean against 41.6 us for the tiled plan, worst pair 74 us
  // against 92, and within 1 - 7 us on the other 30 (profiles/r06_sweep_all_pairs_v6_vs_tiled.json) -- so neither the estimate (25 us) nor the tiled
  // plan's walk over 640 (tile, chunk) pairs (0.17 - 0.5 ms of the planning thread) runs any more.  Round 5 took it below 14 estimated tiles per wave
  // (176 pairs).  Lock-step batches keep the tiled plan (the persistent form takes one batch item).
  if (B == 1 && M <= DVMVS_MAX_MEASUREMENTS && D <= DVMVS_MAX_DEPTH_LEVELS && static_cast<long long>(H) * W >= 64 * 64) {
    work_list_host[0] = 0u;      // (an empty list: a tiled launch on it does nothing)
    work_list_host[1] = 0u;
    return 6;
  }
  return dvmvs::sweep_plan_impl(Hm_host, kt_host, B, M, H, W, D, min_depth, max_depth, 0, work_list_host, work_list_bytes, nullptr);
}

int dvmvs::sweep_plan_impl(const float* Hm_host, const float* kt_host, int B, int M, int H, int W, int D, double min_depth, double max_depth,
                           int variant, unsigned int* work_list_host, size_t work_list_bytes, bool* easy_out) {
  if (!Hm_host || !kt_host || !work_list_host || B <= 0 || M <= 0 || H <= 0 || W <= 0 || D <= 0) return DVMVS_EINVAL;
  if (M > DVMVS_MAX_MEASUREMENTS || D > DVMVS_MAX_DEPTH_LEVELS) return DVMVS_EUNSUPPORTED;
  if (!(min_depth > 0.0) || !(max_depth > 0.0) || (variant != 0 && variant != 2 && variant != 3)) return DVMVS_EINVAL;
  const double inv_base = 1.0 / max_depth, inv_step = D > 1 ? (1.0 / min_depth - 1.0 / max_depth) / (D - 1) : 0.0;
  const size_t words = work_list_bytes / sizeof(unsigned int);
  long long d[8], w[8];
  int rc = 0;
  if (variant != 3) {
    rc = dvmvs::sweep_work_list_host(0, Hm_host, kt_host, B, M, H, W, D, inv_base, inv_step, work_list_host, words, d);
    if (rc < 0) return rc;
    // nothing queued for the second pass in this plan: the single-pass launch (no second kernel; should the kernel's own plan
    // disagree, it gathers that run inline)
    const int chosen = d[3] == 0 ? 4 : 2;
    if (easy_out) *easy_out = dvmvs::sweep_is_easy(d);
    if (variant == 2 || dvmvs::sweep_is_easy(d)) return chosen;
    dvmvs::sweep_plan_stats_host(1, Hm_host, kt_host, B, M, H, W, D, inv_base, inv_step, w);
    if (dvmvs::sweep_model_us(0, d, B, H, W, D) <= dvmvs::sweep_model_us(1, w, B, H, W, D)) return chosen;
  }
  rc = dvmvs::sweep_work_list_host(1, Hm_host, kt_host, B, M, H, W, D, inv_base, inv_step, work_list_host, words, w);
  return rc < 0 ? rc : (w[3] == 0 ? 5 : 3);
}
