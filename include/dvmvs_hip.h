/*
 * dvmvs_hip.h -- C ABI of the MI355X (gfx950) plane-sweep depth kernels.
 *
 * The reference (ardaduz/deep-video-mvs) has no FFI layer: its hot path is a set of Python functions over torch
 * tensors.  This header is the boundary a maintainer binds instead (ctypes stub shown in
 * INTEGRATION.md).  Each entry point names the reference interface it replaces.
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer to contiguous fp32 data unless the comment says "host array" or the parameter is named
 *     *_host (the host-side sweep plan: those functions make no HIP call at all);
 *   - tensors are NCHW, batch-major, exactly as the reference lays them out;
 *   - the caller (PyTorch) allocates and owns every buffer including outputs and workspaces;
 *   - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); no entry point synchronises,
 *     allocates, or keeps global state, so calls are re-entrant and capturable into a hipGraph;
 *   - return value: 0 on success, a positive hipError_t on a HIP failure, a negative DVMVS_E* on a bad argument.
 *     Nothing is enqueued when the return value is negative.
 */
#ifndef DVMVS_HIP_H
#define DVMVS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI 4 (round 4) = ABI 3 + additions, no signature of ABI 3 changed: variant 3 of dvmvs_cost_volume_fwd, the host-side sweep plan
 * (dvmvs_sweep_plan_stats / _select_variant / _work_list / dvmvs_sweep_plan, dvmvs_cost_volume_planned_fwd), the bottleneck convolution
 * (dvmvs_bottleneck_conv_*, dvmvs_partial_sums_bias_act_fwd, dvmvs_lstm_gates_partials_fwd) and two training gradients
 * (dvmvs_upsample2x_bwd, dvmvs_depthwise_conv_bwd).
 * ABI 5 = ABI 4 + the direct convolution of the larger maps and the depth heads (dvmvs_direct_conv_*, dvmvs_conv_head_fwd).
 * ABI 6 (round 5) = ABI 5 + variant 6 of dvmvs_cost_volume_fwd (the correlate-then-interpolate sweep on the fp32 matrix cores) and the
 * one-launch re-projection of the frame path (dvmvs_depth_reproject_estimate_fwd), dvmvs_sweep_plan6 / dvmvs_sweep_mfma_estimate,
 * dvmvs_nchw_to_nhwc, dvmvs_copy_batch; no earlier signature changed.
 * ABI 7 (round 6) = ABI 6 + dvmvs_direct_conv_dual_fwd (the direct convolution with a second, channels-last copy of its output written in the
 * same epilogue: the frame engine's keyframe features reach the MFMA sweep without a transposing launch); no earlier signature changed.
 * Variant 6 of dvmvs_cost_volume_fwd runs as one persistent 16-wave workgroup per CU where the problem allows (csrc/sweep_mfma.hip): same
 * arguments, bit-identical volumes.
 * ABI 8 (round 6) = ABI 7 + the 1x1 convolution with bias, ReLU and the residual add in its store path (dvmvs_pointwise_conv_*); no earlier
 * signature changed; dvmvs_bottleneck_conv_up2x_fwd and the 32x40 stride-2 shape of dvmvs_bottleneck_conv_fwd; dvmvs_host_pointer_device_visible, dvmvs_upsample2x_pair_fwd. */
#define DVMVS_ABI_VERSION 8
#define DVMVS_MAX_MEASUREMENTS 8      /* measurement frames fused per launch */
#define DVMVS_MAX_DEPTH_LEVELS 256    /* sweep planes per launch */

#define DVMVS_LAYOUT_NCHW 0           /* [B,C,H,W] contiguous: the reference's layout */
#define DVMVS_LAYOUT_NHWC 1           /* [B,H,W,C] contiguous ("channels last") */

#define DVMVS_EINVAL (-1)             /* null pointer / non-positive dimension / out-of-range count */
#define DVMVS_EUNSUPPORTED (-2)       /* shape outside what the kernels were built for */
#define DVMVS_ELIBRARY (-3)           /* a call into MIOpen failed (dvmvs_conv_bias_act_fwd) */

typedef void* dvmvs_stream_t;         /* hipStream_t */

/* ABI / build identification. */
int dvmvs_abi_version(void);
const char* dvmvs_build_arch(void);   /* "gfx950" */
const char* dvmvs_error_string(int code);
/* Launches an empty kernel named dvmvs::trace_marker_kernel (profiling: brackets a region of a kernel trace). */
int dvmvs_trace_marker(dvmvs_stream_t stream);

/*
 * Small pose algebra: WHERE IT IS EVALUATED, AND WHY IT IS AN ARGUMENT (ABI 3).
 *
 * The reference derives three tiny matrices per frame from the camera poses with fp32 torch ops on the host side of its
 * hot functions:
 *   sweep constants   E = inverse(pose2) * pose1,  Hm = K R K^-1,  kt = K t        (/root/reference/dvmvs/utils.py:51-56)
 *   splat transform   inverse(reference_pose) * measurement_pose                   (utils.py:121)
 *   warp transform    inverse(previous_pose) * current_pose                        (dvmvs/convlstm.py:30)
 * Their fp32 round-off (cancellation in the relative translation, ~5e-7 m) moves sample positions by up to 3e-4 px on the
 * nearest plane -- more than everything else on the path together -- so "results identical to the reference" requires
 * the very same matrices.  The entry points below therefore take them as DEVICE ARRAYS, computed by the caller exactly as
 * the reference computes them (the Python surface does it with the reference's own torch expressions, see
 * deep-video-mvs_amd/dvmvs/pose_algebra.py), and do no pose algebra themselves.
 *   Hm  [B,M,9]   row-major 3x3 per (batch item, measurement frame)
 *   kt  [B,M,3]
 * For callers that prefer accuracy over bit-parity (or want no host involvement at all) dvmvs_sweep_matrices and
 * dvmvs_relative_pose evaluate the same algebra on the device in fp64 and round once ("exact" mode).
 */

/*
 * Hm[b,m] = K[b] R K[b]^-1, kt[b,m] = K[b] t with [R|t] = inverse(pose2s[m][b]) * pose1[b]; fp64 on the device, rounded once.
 * Opt-in alternative to the reference's fp32 host algebra (utils.py:51-56).
 *   pose1 [B,4,4]   pose2s host array of M device pointers, each [B,4,4]   K [B,3,3]   Hm [B,M,9] out   kt [B,M,3] out
 */
int dvmvs_sweep_matrices(const float* pose1, const float* const* pose2s, const float* K, float* Hm, float* kt,
                         int B, int M, dvmvs_stream_t stream);

/*
 * Fused plane-sweep warp + feature correlation over all planes and all measurement frames, written once.
 * Replaces dvmvs.utils.calculate_cost_volume_by_warping (M == 1) and dvmvs.utils.cost_volume_fusion (M >= 1):
 *   /root/reference/dvmvs/utils.py:45-86 and :89-107 (everything after the small pose algebra of :51-56, see above).
 *
 *   image1      [B,C,H,W]   reference-frame features
 *   image2s     host array of M device pointers, each [B,C,H,W] measurement-frame features
 *   Hm          [B,M,9]     K R K^-1 per (batch item, measurement frame)
 *   kt          [B,M,3]     K t
 *   cost_volume [B,D,H,W]   out; plane 0 = max_depth ... plane D-1 = min_depth, uniform in inverse depth
 *   dot_product 1: sum_c(f1*warp(f2))/C   0: sum_c|f1-warp(f2)|  (utils.py:81-84); result is the mean over M
 *   variant     0 = pick the fastest kernel for the shape; 1 = force the generic reference-order kernel (taps through
 *               the vector L1); 2 = force the LDS-tiled sweep in its default configuration (dot_product only); 3 = the
 *               LDS-tiled sweep in its wide-baseline configuration (72 KB sample boxes, 512-thread workgroups: faster where
 *               the default one has to split or queue runs of planes, slower on easy pairs -- dvmvs_sweep_select_variant
 *               decides from the matrices); 4 / 5 = configurations 2 / 3 as ONE launch: no second pass, a run that cannot be staged is
 *               gathered inline by the sweep kernel (what dvmvs_sweep_plan returns when its plan queues nothing; also what a NULL
 *               workspace gives); 6 = the correlate-then-interpolate sweep on the fp32 matrix cores (csrc/sweep_mfma.hip: the 32-channel dot
 *               product per measurement CELL on v_mfma_f32_16x16x4_f32, four table look-ups per (pixel, plane, frame); up to 32 channels,
 *               either layout, any image size, no workspace, no work list, no host plan; since round 6 one persistent 16-wave workgroup per CU
 *               that gives every SIMD the same mix of work, for one batch item with >= 4096 (pixel group, 16-plane chunk) items whose
 *               K t / depth table fits in LDS, D * M <= 512); 7 = variant 6 as one work item per workgroup for every shape (what 6 falls back
 *               to; bit-identical to 6: the comparison kernel of tests and bench).  Each is bit-reproducible; they differ in fp32
 *               summation order.
 *   image2_layout DVMVS_LAYOUT_NCHW, or DVMVS_LAYOUT_NHWC when the MEASUREMENT maps are stored channels-last (a keyframe's
 *               features are reused as measurement features by later frames, so a runner converts them once per
 *               keyframe).  Supported by the LDS-tiled dot-product kernel (C % 4 == 0, H*W >= 4096); image1 and
 *               cost_volume are always NCHW.
 *   workspace   optional device scratch of dvmvs_cost_volume_workspace_bytes(B, M, H, W, D) bytes for the two-pass form of
 *               the LDS-tiled sweep: runs of planes whose sample footprint does not fit in LDS (magnified or behind-camera
 *               views) are queued there by the sweep launch and finished by a second, finely grained gather launch that
 *               is spread over the whole chip, in a fixed order (the volume is bit-reproducible run to run).
 *               CONTRACT: the first 16 bytes must be zero when a call starts.  Every call leaves them zero again, so the
 *               owner zero-fills the buffer once, when it allocates it; a buffer must not be shared by calls that may
 *               overlap in time (different streams).  With workspace == NULL (or too small) such runs are gathered inline
 *               by the sweep kernel itself (one launch, long tail on wide-baseline / forward-motion pairs).
 */
size_t dvmvs_cost_volume_workspace_bytes(int B, int M, int H, int W, int D);
int dvmvs_cost_volume_fwd(const float* image1, const float* const* image2s, const float* Hm, const float* kt,
                          float* cost_volume, int B, int M, int C, int H, int W, int D,
                          double min_depth, double max_depth, int dot_product, int variant, int image2_layout,
                          float* workspace, size_t workspace_bytes, dvmvs_stream_t stream);

/*
 * HOST-side model of the LDS-tiled sweep's run plan (no HIP call, no device memory): Hm_host / kt_host are HOST copies of the
 * matrices the launch will get -- they are on the host before the launch anyway (see "small matrices" above).
 *   configuration 0 = default (variant 2), 1 = wide-baseline (variant 3)
 *   stats[8] out: staged runs, LDS records of all staged runs, runs entirely outside the image, runs queued for the second
 *                 pass, planes of those runs, workgroups with at least one queued run, the largest number of staged runs of
 *                 one workgroup (all workgroups are resident at once: the longest chain is the launch's span), the largest
 *                 number of queued planes of one workgroup
 * dvmvs_sweep_select_variant returns the variant (2 or 3) the cost model built on those numbers expects to be faster for this
 * keyframe pair (negative DVMVS_E* on bad arguments); a captured frame graph per variant is replayed accordingly.
 */
int dvmvs_sweep_plan_stats(const float* Hm_host, const float* kt_host, int B, int M, int H, int W, int D,
                           double min_depth, double max_depth, int configuration, long long* stats);
int dvmvs_sweep_select_variant(const float* Hm_host, const float* kt_host, int B, int M, int H, int W, int D,
                               double min_depth, double max_depth);

/*
 * Work list of the LDS-tiled sweep, built on the HOST from the host copies of the matrices (no HIP call).  All workgroups of a sweep
 * launch are resident at once, so the launch lasts as long as its slowest workgroup; on wide-baseline / forward-motion pairs a few
 * (tile, 8-plane chunk) pairs need 5-8 staged runs instead of 2.  The list cuts those into plane sub-ranges of at most 3 staged runs
 * that separate workgroups process in parallel.  Upload it (dvmvs_sweep_work_list_bytes(B, H, W, D) bytes; the returned value is the
 * number of 32-bit words actually used, negative on error) and pass the DEVICE copy to dvmvs_cost_volume_planned_fwd, which is
 * dvmvs_cost_volume_fwd with that one extra argument (NULL = static numbering; used only together with a workspace).  `configuration`
 * as for dvmvs_sweep_plan_stats; it must match the variant of the launch (2 -> 0, 3 -> 1; 0 picks the default configuration).
 * Results do not depend on the list beyond fp32 summation order (a cut can turn a queued run into staged ones); with a given list the
 * volume is bit-reproducible.
 */
size_t dvmvs_sweep_work_list_bytes(int B, int H, int W, int D);
/* dvmvs_sweep_select_variant + dvmvs_sweep_work_list in one walk (what a frame loop calls once per keyframe): `variant` 0 = decide,
 * 2 / 3 = that configuration; leaves the chosen configuration's work list in work_list_host and returns the variant to launch with
 * (2 / 3, or 4 / 5 = the same configuration without a second pass when the plan queues nothing for it; negative on error). */
int dvmvs_sweep_plan(const float* Hm_host, const float* kt_host, int B, int M, int H, int W, int D, double min_depth, double max_depth,
                     int variant, unsigned int* work_list_host, size_t work_list_bytes);
/* ABI 6: HOST-side work estimate of variant 6 (the correlate-then-interpolate sweep) for batch item 0 of these matrices, no HIP call:
 * stats[4] = mean 16-cell tiles per wave, mean passes per wave, mean strips beyond the first per wave, fraction of waves whose footprint
 * cannot be bounded from its corners (behind-camera / non-finite); sampled on every sixth pixel group row / column (25 us at 160 x 128 x 64).
 * dvmvs_sweep_plan6 uses it. */
int dvmvs_sweep_mfma_estimate(const float* Hm_host, const float* kt_host, int B, int M, int H, int W, int D, double min_depth, double max_depth,
                              double* stats);
/* dvmvs_sweep_plan with variant 6 as a candidate.  Since round 6 (ABI 7) a single-item launch (B == 1, H * W >= 4096) always gets variant 6: the
 * work list is left EMPTY (header count 0: a tiled launch on it does nothing), neither the estimate nor the tiled plan runs, the call is a few
 * nanoseconds -- variant 6 in its persistent form is the faster kernel on 255 of the sample scene's 285 keyframe pairs and within 1 - 7 us on the
 * rest (profiles/r06_sweep_all_pairs_v6_vs_tiled.json; round 5 took it below 14 estimated tiles per wave: profiles/r05_sweep_selection.md).
 * Lock-step batches (B > 1) get the tiled plan as dvmvs_sweep_plan makes it.  For callers whose measurement maps are channels-last (what variant 6
 * is fast with) and have at most 32 channels. */
int dvmvs_sweep_plan6(const float* Hm_host, const float* kt_host, int B, int M, int H, int W, int D, double min_depth, double max_depth,
                      unsigned int* work_list_host, size_t work_list_bytes);
/* [B,C,H,W] -> [B,H,W,C] (C <= 64, a multiple of 4), one launch: how a keyframe's features enter a channels-last feature cache. */
int dvmvs_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, dvmvs_stream_t stream);
/* n <= 8 contiguous device-to-device copies of n_floats[j] floats (a multiple of 4; 16-byte aligned pointers; host arrays of n entries) in
 * ONE launch -- the small copies a frame step makes in front of its graph (features into the feature cache, measurement maps and the next
 * image into the buffers the graph reads).  Ranges must not overlap each other (the copies run concurrently). */
int dvmvs_copy_batch(const float* const* srcs, float* const* dsts, const long long* n_floats, int n, dvmvs_stream_t stream);
/* (ABI 8) A source of dvmvs_copy_batch may also be PINNED HOST memory that the device sees at the same address (the frame engine's parameter
 * block rides up in the frame's copy batch instead of a blit launch of its own): 1 when p is such memory for the current device, else 0. */
int dvmvs_host_pointer_device_visible(const void* p);
int dvmvs_sweep_work_list(const float* Hm_host, const float* kt_host, int B, int M, int H, int W, int D,
                          double min_depth, double max_depth, int configuration, unsigned int* work_list_host, size_t work_list_bytes);
int dvmvs_cost_volume_planned_fwd(const float* image1, const float* const* image2s, const float* Hm, const float* kt,
                                  float* cost_volume, int B, int M, int C, int H, int W, int D,
                                  double min_depth, double max_depth, int dot_product, int variant, int image2_layout,
                                  float* workspace, size_t workspace_bytes, const unsigned int* work_list, dvmvs_stream_t stream);

/*
 * Gradient of the fused cost volume (dot_product mode) w.r.t. both feature maps; poses/K carry no gradient
 * (autograd through utils.py:75-82; the sampling grid is data).
 *   grad_cost   [B,D,H,W]
 *   Hm, kt      as for the forward
 *   grad_image1 [B,C,H,W]   out (overwritten)
 *   grad_image2s host array of M device pointers, each [B,C,H,W]; MUST be zero-filled by the caller (the gradient is ADDED to
 *               it -- by a gather without atomics, so the result is bit-reproducible); an entry may be NULL to skip that frame.
 */
int dvmvs_cost_volume_bwd(const float* grad_cost, const float* image1, const float* const* image2s,
                          const float* Hm, const float* kt, float* grad_image1, float* const* grad_image2s,
                          int B, int M, int C, int H, int W, int D,
                          double min_depth, double max_depth, dvmvs_stream_t stream);

/*
 * Depth-conditioned inverse warp of the ConvLSTM hidden state.
 * Replaces dvmvs.utils.warp_frame_depth (/root/reference/dvmvs/utils.py:205-258; normalize_points=False,
 * bilinear) and, with zero_invalid != 0, the caller's mask h[depth <= 0.01] = 0 (dvmvs/convlstm.py:32-41).
 *   image_src     [B,C,H,W]
 *   depth_dst     [B,1,H,W]
 *   src_trans_dst [B,4,4]
 *   camera_matrix [B,3,3]
 *   out           [B,C,H,W]
 */
int dvmvs_hidden_warp_fwd(const float* image_src, const float* depth_dst, const float* src_trans_dst,
                          const float* camera_matrix, float* out, int B, int C, int H, int W,
                          int zero_invalid, dvmvs_stream_t stream);

/*
 * Gradient of the hidden-state warp w.r.t. image_src.  As in the reference the validity mask is NOT applied to
 * the gradient (convlstm.py:41 mutates .data outside autograd).  grad_src [B,C,H,W] MUST be zero-filled.
 */
int dvmvs_hidden_warp_bwd(const float* grad_out, const float* depth_dst, const float* src_trans_dst,
                          const float* camera_matrix, float* grad_src, int B, int C, int H, int W,
                          dvmvs_stream_t stream);

/*
 * out[b] = inverse(a[b]) * c[b] for 4x4 matrices (evaluated in fp64, rounded once to fp32): the "exact" alternative to
 * the reference's fp32 torch.bmm(torch.inverse(previous_pose), current_pose) (dvmvs/convlstm.py:30, utils.py:121).
 */
int dvmvs_relative_pose(const float* a, const float* c, float* out, int B, dvmvs_stream_t stream);

/*
 * ConvLSTM gate fusion given the 4*hidden-channel convolution output (split order i,f,o,g):
 *   i,f,o = sigmoid; g = celu(LN_hw(cc_g)); c' = LN_hw(f*c + i*g); h' = o*celu(c').  LN: biased variance,
 *   eps 1e-5, no affine; celu alpha = 1.
 * Replaces /root/reference/dvmvs/convlstm.py:45-59.
 *   combined_conv [B,4*hidden,H,W]   c_cur [B,hidden,H,W]   h_next, c_next [B,hidden,H,W] out; c_next may be c_cur (in-place state update)
 */
int dvmvs_lstm_gates_fwd(const float* combined_conv, const float* c_cur, float* h_next, float* c_next,
                         int B, int hidden, int H, int W, dvmvs_stream_t stream);

/*
 * Backward of the gate fusion: recomputes the gates from (combined_conv, c_cur).
 *   grad_h, grad_c [B,hidden,H,W] (either may be NULL = zero)   grad_cc [B,4*hidden,H,W], grad_c_cur out
 */
int dvmvs_lstm_gates_bwd(const float* grad_h, const float* grad_c, const float* combined_conv,
                         const float* c_cur, float* grad_cc, float* grad_c_cur,
                         int B, int hidden, int H, int W, dvmvs_stream_t stream);

/*
 * Gradients of two frame-path ops for training (BASELINE.json configs[4]); both are gathers / fixed-order reductions without
 * atomics, i.e. bit-reproducible (csrc/train_ops.hip).
 *   dvmvs_upsample2x_bwd      adjoint of dvmvs_upsample2x_fwd (x2 bilinear, align_corners; /root/reference/dvmvs/fusionnet/model.py:59,114):
 *                             grad_out [B,C,2H,2W] -> grad_in [B,C,H,W]
 *   dvmvs_depthwise_conv_bwd  depthwise k x k (3 or 5), padding k/2, stride 1 or 2 (the MnasNet layers): grad_out [B,C,OH,OW], in [B,C,H,W],
 *                             weight [C,1,k,k] -> grad_in [B,C,H,W] and / or grad_weight [C,1,k,k] (either pointer may be NULL);
 *                             workspace: dvmvs_depthwise_conv_bwd_workspace_bytes() bytes of caller-owned scratch for the weight
 *                             gradient's slice sums (0 bytes -> may be NULL)
 */
int dvmvs_upsample2x_bwd(const float* grad_out, float* grad_in, int B, int C, int H, int W, dvmvs_stream_t stream);
size_t dvmvs_depthwise_conv_bwd_workspace_bytes(int B, int C, int H, int W, int kernel_size, int stride);
int dvmvs_depthwise_conv_bwd(const float* grad_out, const float* in, const float* weight, float* grad_in, float* grad_weight,
                             float* workspace, int B, int C, int H, int W, int kernel_size, int stride, dvmvs_stream_t stream);

/*
 * 3x3, padding-1 convolutions on the bottleneck maps of a 320x256 frame (8x10, 16x20 with stride 1 or 2, and -- ABI 8 -- 32x40 with
 * stride 2: the layer that takes the 1/8 map down) as a weight-streaming
 * fp32 MFMA GEMM with a DETERMINISTIC split-K (csrc/bottleneck_conv.hip).  Replaces, for those shapes only, the nn.Conv2d of the
 * ConvLSTM cell (/root/reference/dvmvs/convlstm.py:43-44: 1024 -> 2048 channels) and the 256 / 512-channel layers around it
 * (fusionnet/model.py:167-305), which MIOpen solves with split-K kernels that accumulate with float atomics (results vary from
 * run to run).  Inference only (no gradient); the other convolutions of an inference frame have their own entries below (dvmvs_direct_conv_*,
 * dvmvs_pointwise_conv_*, dvmvs_depthwise_conv_fwd); training convolutions stay on MIOpen.
 *   dvmvs_bottleneck_conv_pack    weight [C_out,C_in,3,3] -> packed (dvmvs_bottleneck_conv_packed_bytes; C_in % 16 == 0), once
 *   dvmvs_bottleneck_conv_splits  number S of partial sums for a problem, DVMVS_EUNSUPPORTED for shapes the kernel does not take
 *   dvmvs_bottleneck_conv_fwd     x [B,C_in,H_in,W_in] -> partials [S][B][C_out][H_out*W_out]: partial s holds the contribution of
 *                                 input channels [s*C_in/S, (s+1)*C_in/S); their sum in ascending s is the convolution
 *   dvmvs_partial_sums_bias_act_fwd  dst[b,c,:] = act(sum_s partials[s,b,c,:] + bias[c]); activation 0 none / 1 ReLU; dst batch
 *                                 item b starts at dst + b*dst_batch_stride (a channel slice of a concatenation buffer)
 *   dvmvs_lstm_gates_partials_fwd dvmvs_lstm_gates_fwd on a convolution output that arrives as n_partials partial sums
 */
size_t dvmvs_bottleneck_conv_packed_bytes(int C_out, int C_in);
int dvmvs_bottleneck_conv_pack(const float* weight, float* packed, int C_out, int C_in, dvmvs_stream_t stream);
int dvmvs_bottleneck_conv_splits(int B, int C_out, int C_in, int H_in, int W_in, int stride);
int dvmvs_bottleneck_conv_fwd(const float* x, const float* packed, float* partials, int B, int C_in, int H_in, int W_in, int C_out,
                              int stride, dvmvs_stream_t stream);
/* (ABI 8) dvmvs_bottleneck_conv_fwd on the 2x bilinear up-sampling (align_corners = True: dvmvs_upsample2x_fwd's values, bit for bit) of
 * x [B,C_in,H_in/2,W_in/2], interpolated while the input is staged -- the decoder's first up-convolution without the up-sampling launch in
 * front of it (/root/reference/dvmvs/fusionnet/model.py UpconvolutionLayer).  H_in x W_in = 16 x 20 (the up-sampled map), stride 1; same
 * split count and partial-sum layout as dvmvs_bottleneck_conv_fwd for that shape. */
int dvmvs_bottleneck_conv_up2x_fwd(const float* x, const float* packed, float* partials, int B, int C_in, int H_in, int W_in, int C_out,
                                   dvmvs_stream_t stream);
int dvmvs_partial_sums_bias_act_fwd(const float* partials, int n_partials, float* dst, long long dst_batch_stride, const float* bias,
                                    int B, int C, int HW, int activation, dvmvs_stream_t stream);
int dvmvs_lstm_gates_partials_fwd(const float* conv_partials, int n_partials, const float* c_cur, float* h_next, float* c_next,
                                  int B, int hidden, int H, int W, dvmvs_stream_t stream);

/*
 * Dense 3x3 / 5x5 convolutions ("same" padding k/2, stride 1 or 2) of the 1/8 ... full-resolution maps of a frame as a direct fp32-MFMA
 * convolution with bias + ReLU in its store path, and the one-output-channel 3x3 depth heads (csrc/direct_conv.hip).  Replace, for the
 * problems dvmvs_direct_conv_tile accepts, the nn.Conv2d + BatchNorm(folded) + ReLU of the cost-volume encoder / decoder, the FPN's
 * smoothing layers and the stem (/root/reference/dvmvs/fusionnet/model.py:167-305, dvmvs/layers.py conv_layer) at inference; no
 * gradient.  Deterministic: input-channel splits are added through LDS in a fixed order.
 *   dvmvs_direct_conv_tile          0 when the kernel does not take the problem (output width not a multiple of 40, C_out % 16 != 0,
 *                                   other kernel sizes / strides: the caller keeps its library convolution); else the number of
 *                                   16-channel output tiles per wave (1 or 2) the weights have to be packed for
 *   dvmvs_direct_conv_pack          weight [C_out,C_in,k,k] -> packed (dvmvs_direct_conv_packed_bytes), once per (layer, n_tile)
 *   dvmvs_direct_conv_fwd           x [B,C_in,H,W] (batch item b at x + b*x_batch_stride, 0 = dense) -> dst [B,C_out,H/stride,W/stride]
 *                                   (batch item b at dst + b*dst_batch_stride, 0 = dense: a channel slice of a concatenation buffer);
 *                                   bias may be NULL; activation 0 none, 1 ReLU; DVMVS_EINVAL when n_tile is not the problem's
 *   dvmvs_direct_conv_dual_fwd      (ABI 7) the same, and when dst_nhwc is not NULL the same values once more as a dense channels-last map
 *                                   [B,H/stride,W/stride,C_out] (DVMVS_LAYOUT_NHWC: what variant 6 of the sweep reads one 128-byte line per cell)
 *   dvmvs_conv_head_fwd             3x3, padding 1, ONE output channel: weight [1,C_in,3,3]; dst [B,1,H,W]; bias may be NULL (raw
 *                                   convolution output when also activation == 0); activation / p0 / p1 as dvmvs_bias_act_fwd
 */
int dvmvs_direct_conv_tile(int B, int C_in, int H, int W, int C_out, int kernel_size, int stride);
size_t dvmvs_direct_conv_packed_bytes(int C_out, int C_in, int kernel_size, int n_tile);
int dvmvs_direct_conv_pack(const float* weight, float* packed, int C_out, int C_in, int kernel_size, int n_tile, dvmvs_stream_t stream);
int dvmvs_direct_conv_fwd(const float* x, long long x_batch_stride, const float* packed, int n_tile, const float* bias, float* dst,
                          long long dst_batch_stride, int B, int C_in, int H, int W, int C_out, int kernel_size, int stride, int activation,
                          dvmvs_stream_t stream);
int dvmvs_direct_conv_dual_fwd(const float* x, long long x_batch_stride, const float* packed, int n_tile, const float* bias, float* dst,
                               long long dst_batch_stride, float* dst_nhwc, int B, int C_in, int H, int W, int C_out, int kernel_size, int stride,
                               int activation, dvmvs_stream_t stream);
int dvmvs_conv_head_fwd(const float* x, long long x_batch_stride, const float* weight, const float* bias, float* dst,
                        long long dst_batch_stride, int B, int C_in, int H, int W, int activation, float p0, float p1, dvmvs_stream_t stream);

/*
 * (ABI 8) 1x1 convolutions (stride 1, no padding, one group) of a frame as an fp32-MFMA GEMM straight from the NCHW map, with the bias add,
 * ReLU and the residual add of dvmvs_bias_act_fwd in its store path (csrc/pointwise_conv.hip).  Replace the nn.Conv2d(k = 1) +
 * BatchNorm(folded) (+ ReLU) (+ residual) of the MnasNet feature extractor's expansion / projection layers and of the feature pyramid's
 * lateral layers (/root/reference/dvmvs/fusionnet/model.py:20-124; torchvision's _InvertedResidual) at inference: one launch instead of a
 * library GEMM + an epilogue launch.  No gradient.  Deterministic: input-channel splits are added through LDS in a fixed order.
 *   dvmvs_pointwise_conv_supported  1 when dvmvs_pointwise_conv_fwd takes the problem (C_in % 4 == 0, H*W % 4 == 0, activation 0 / 1, mode 2 only
 *                                   for even H and W % 4 == 0), else 0: the caller keeps its library convolution
 *   dvmvs_pointwise_conv_pack       weight [C_out,C_in] -> packed (dvmvs_pointwise_conv_packed_bytes), once per layer
 *   dvmvs_pointwise_conv_fwd        x [B,C_in,H,W] (batch item b at x + b*x_batch_stride, 0 = dense) -> dst [B,C_out,H,W] (batch item b at
 *                                   dst + b*dst_batch_stride, 0 = dense: a channel slice of a concatenation buffer; 16-byte aligned):
 *                                   dst = act(conv + bias) + residual.  bias may be NULL; activation 0 none, 1 ReLU; residual_mode 0 none,
 *                                   1 residual [B,C_out,H,W], 2 residual [B,C_out,H/2,W/2] nearest-up-sampled (batch item b at residual +
 *                                   b*residual_batch_stride, 0 = dense) -- the meaning of dvmvs_bias_act_fwd's arguments;
 *                                   splits: input-channel splits per output tile, 1 ... 16, 0 = chosen from the problem size
 */
int dvmvs_pointwise_conv_supported(int B, int C_in, int H, int W, int C_out, int activation, int residual_mode);
size_t dvmvs_pointwise_conv_packed_bytes(int C_out, int C_in);
int dvmvs_pointwise_conv_pack(const float* weight, float* packed, int C_out, int C_in, dvmvs_stream_t stream);
int dvmvs_pointwise_conv_fwd(const float* x, long long x_batch_stride, const float* packed, const float* bias, const float* residual,
                             long long residual_batch_stride, int residual_mode, float* dst, long long dst_batch_stride, int B, int C_in,
                             int H, int W, int C_out, int activation, int splits, dvmvs_stream_t stream);

/*
 * Forward splat (z-buffer, farthest wins) of the previous full-resolution depth into the current view at half
 * resolution; untouched pixels are 0.  No host round trip, order-independent (atomic max on the float bits).
 * Replaces dvmvs.utils.get_non_differentiable_rectangle_depth_estimation (/root/reference/dvmvs/utils.py:110-154).
 *   transformation [B,4,4]  inverse(reference_pose) * measurement_pose (utils.py:121), computed by the caller (see above)
 *   previous_depth [B,1,Hf,Wf]   full_K, half_K [B,3,3]
 *   out [B,1,Hf/2,Wf/2]  (zeroed by this call)
 * If out_lowres != NULL it also receives the nearest-neighbour down-sampling by `lowres_factor` that the call
 * site applies next (fusionnet/run-testing.py:187-189): [B,1,(Hf/2)/f,(Wf/2)/f].
 */
int dvmvs_depth_reproject_fwd(const float* transformation, const float* previous_depth, const float* full_K,
                              const float* half_K, float* out, float* out_lowres, int lowres_factor,
                              int B, int full_height, int full_width, dvmvs_stream_t stream);
/*
 * The frame path's form of the above: only the decimated estimate [B,1,(Hf/2)/f,(Wf/2)/f] is produced.  `zbuffer`
 * [B,Hf/2,Wf/2] is caller-owned scratch that MUST be all-zero when the call starts; the call leaves it all-zero again (the
 * decimation kernel clears it), so the owner zero-fills it once: two launches per frame instead of three.
 */
int dvmvs_depth_reproject_lowres_fwd(const float* transformation, const float* previous_depth, const float* full_K,
                                     const float* half_K, float* zbuffer, float* out_lowres, int lowres_factor,
                                     int B, int full_height, int full_width, dvmvs_stream_t stream);

/*
 * ABI 6.  The frame path in ONE launch: the ConvLSTM reads only rows / columns 0, f, 2f, ... of the estimate
 * (fusionnet/run-testing.py:176-189), so the splat goes straight into `estimate` [B,1,Hf/2/f,Wf/2/f] -- bit-identical to
 * out_lowres of the functions above -- which MUST be all-zero when the call starts.  `estimate_to_clear` (optional, another
 * buffer of that shape) is zero-filled by the same launch: a caller that alternates between two estimate buffers from frame
 * to frame zero-fills them once and never launches a clear.
 */
int dvmvs_depth_reproject_estimate_fwd(const float* transformation, const float* previous_depth, const float* full_K,
                                       const float* half_K, float* estimate, float* estimate_to_clear, int lowres_factor,
                                       int B, int full_height, int full_width, dvmvs_stream_t stream);

/*
 * Frame-path epilogues (not part of the reference's function list; they replace ATen elementwise / copy launches that sit
 * between MIOpen convolutions on the per-frame path, /root/reference/dvmvs/layers.py:39-65 and fusionnet/model.py:57,112,231-232,290-303).
 * At batch 1 a frame is ~250 launches of a few microseconds each, so what these buy is launches, not FLOPs.
 *   dvmvs_bias_act_fwd:     dst[b,c,:,:] = act(x[b,c,:,:] + bias[c]) [+ residual]; x is a dense [B,C,H,W] convolution output, dst may
 *                           be x (in place) or a CHANNEL SLICE of a larger buffer: batch item b starts at dst + b * dst_batch_stride
 *                           (in floats), its C planes are dense -- a torch.cat of convolution outputs is then never copied.
 *                           bias may be NULL; activation 0 none, 1 ReLU, 2 sigmoid, 3 sigmoid then the decoder's depth mapping
 *                           1 / (p0 * s + p1) (fusionnet/model.py:231-232,297-303); residual_mode 0 none, 1 residual [B,C,H,W]
 *                           (MnasNet shortcut), 2 residual [B,C,H/2,W/2] nearest-up-sampled on the fly (FPN top-down sum).
 *   dvmvs_bias_act_inplace: the same with dst = x (activations 0..2).
 *   dvmvs_upsample2x_fwd:   torch.nn.functional.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True);
 *                           in [B,C,H,W] -> out [B,C,2H,2W], batch item b at out + b * out_batch_stride (0 = dense).  With
 *                           pre_activation != 0 (1 ReLU, 2 sigmoid) the input is a raw convolution output and
 *                           act(in + pre_bias[c]) (pre_bias may be NULL) is applied to the taps on the fly.
 */
int dvmvs_bias_act_fwd(const float* x, float* dst, long long dst_batch_stride, const float* bias, const float* residual,
                       int residual_mode, int B, int C, int H, int W, int activation, float p0, float p1, dvmvs_stream_t stream);
int dvmvs_bias_act_inplace(float* x, const float* bias, const float* residual, int residual_mode, int B, int C, int H, int W,
                           int activation, dvmvs_stream_t stream);
int dvmvs_upsample2x_fwd(const float* in, float* out, long long out_batch_stride, const float* pre_bias, int pre_activation,
                         int B, int C, int H, int W, dvmvs_stream_t stream);
/* (ABI 8) Two dvmvs_upsample2x_fwd of maps of the same size [B,C1,H,W] and [B,C2,H,W] in ONE launch: job 1 without pre-activation, job 2 with
 * pre_bias2 / pre_activation2 (a decoder level's feature map and its one-channel depth head, /root/reference/dvmvs/fusionnet/model.py:262-296);
 * the same values bit for bit. */
int dvmvs_upsample2x_pair_fwd(const float* in1, float* out1, long long out1_batch_stride, int C1, const float* in2, float* out2,
                              long long out2_batch_stride, const float* pre_bias2, int pre_activation2, int C2, int B, int H, int W,
                              dvmvs_stream_t stream);
/*
 *   dvmvs_conv_bias_act_fwd: a dense convolution (square kernel K, zero padding, no dilation, one group) WITH its epilogue, as one
 *                           MIOpen fusion plan (convolution + bias [+ ReLU]): out = act(conv(x, weight) + bias[c]).  The convolution
 *                           is MIOpen's in either form; what this saves is the dvmvs_bias_act_fwd launch after it, for the problems
 *                           MIOpen gives to its fp32 Winograd kernel (for the others the plan is slower than convolution + epilogue:
 *                           the caller times both at warm-up, dvmvs/engine.py).  x [B,Cin,H,W], weight [Cout,Cin,K,K], bias [Cout],
 *                           out [B,Cout,Ho,Wo] with batch item b at out + b * out_batch_stride (0 = dense); activation 0 none, 1 ReLU.
 *                           The first call of a problem builds (compiles) its plan -- not inside a stream capture; later calls only
 *                           launch.  DVMVS_EUNSUPPORTED: MIOpen has no fused plan for the problem; DVMVS_ELIBRARY: a MIOpen call failed.
 */
int dvmvs_conv_bias_act_fwd(const float* x, const float* weight, const float* bias, float* out, long long out_batch_stride,
                            int B, int Cin, int H, int W, int Cout, int K, int stride, int padding, int activation, dvmvs_stream_t stream);
/*
 *   dvmvs_depthwise_conv_fwd: depthwise convolution (groups == C, weight [C,1,k,k], k in {3,5}, padding k/2, stride 1|2)
 *                           with bias (may be NULL) and activation fused; in [B,C,H,W] -> out [B,C,OH,OW].  The MnasNet
 *                           depthwise layers of the feature extractor (torchvision mnasnet _InvertedResidual).  With pre_relu != 0
 *                           the input is the RAW output of the preceding 1x1 expansion convolution and relu(in + pre_bias[c])
 *                           (pre_bias may be NULL) is applied to every in-bounds tap on the fly.
 */
int dvmvs_depthwise_conv_fwd(const float* in, const float* weight, const float* bias, const float* pre_bias, int pre_relu, float* out,
                             int B, int C, int H, int W, int kernel_size, int stride, int activation, dvmvs_stream_t stream);

/*
 * TSDF fusion of one RGB-D frame into a voxel volume, in place.  Replaces the `integrate` CUDA kernel the reference's
 * reconstruction script compiles with pycuda (/root/reference/sample-data/run-tsdf-reconstruction.py:79-152) and the
 * per-"gpu loop" launches around it (:240-268): one launch covers the whole volume.
 *   tsdf_vol, weight_vol, color_vol [dim_x,dim_y,dim_z]   (z fastest; colour folded as b * 65536 + g * 256 + r)
 *   origin_*, voxel_size    world position of voxel (0,0,0) and the voxel edge, metres
 *   cam_intr [3,3], cam_pose [4,4] camera-to-world        (device pointers)
 *   color_im, depth_im [im_h,im_w]   folded colour and depth in metres (0 = invalid)
 *   trunc_margin            truncation distance (the reference uses 5 voxels);  obs_weight  weight of this observation
 */
int dvmvs_tsdf_integrate(float* tsdf_vol, float* weight_vol, float* color_vol, int dim_x, int dim_y, int dim_z,
                         float origin_x, float origin_y, float origin_z, float voxel_size, const float* cam_intr,
                         const float* cam_pose, const float* color_im, const float* depth_im, int im_h, int im_w,
                         float trunc_margin, float obs_weight, dvmvs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DVMVS_HIP_H */
