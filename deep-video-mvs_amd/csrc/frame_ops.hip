// Elementwise epilogues of the per-frame network path, gfx950.
//
// MIOpen's fp32 convolutions leave the bias add and the activation to separate ATen kernels, and ATen's bilinear
// up-sampling kernel takes ~80 us on the 512 x 8 x 10 bottleneck map.  At batch 1 a frame is ~370 launches of a few
// microseconds each, so these two ops fuse / replace them:
//   dvmvs_bias_act_inplace : x[b,c,:,:] = act(x[b,c,:,:] + bias[c])      act = none | relu | sigmoid   (one pass)
//   dvmvs_upsample2x_fwd   : F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
// Both are bandwidth / latency trivial; they exist to cut launches and tail latency, not FLOPs.
#include "dvmvs_device.h"

namespace dvmvs {

// ACT 0 none, 1 ReLU, 2 sigmoid, 3 sigmoid followed by the decoder's depth mapping 1 / (p0 * s + p1)
// (/root/reference/dvmvs/fusionnet/model.py:231-232,297-303: inverse_depth_multiplier, inverse_depth_base).
template <int ACT>
__device__ inline float apply_act(float v, float p0 = 0.0f, float p1 = 0.0f) {
  if (ACT == 1) return fmaxf(v, 0.0f);
  if (ACT == 2) return 1.0f / (1.0f + expf(-v));
  if (ACT == 3) {
#pragma clang fp contract(off)
    const float s = 1.0f / (1.0f + expf(-v));
    return 1.0f / (p0 * s + p1);     // multiply, add, reciprocal: the op order of the ATen expression it replaces
  }
  return v;
}

// One workgroup row per (b, c) plane chunk; float4 when the plane size allows it.  Reads the convolution output `x` (dense
// [B,C,H,W]) and writes `dst`, which may be x itself (in place) or a channel slice of a larger buffer -- batch item b of the
// destination starts at dst + b * dst_batch_stride, its C planes are dense -- so that a torch.cat of convolution outputs never
// has to be materialised by a copy kernel.
// RES: 0 none; 1 residual of the same shape is added after the activation; 2 the residual has half the resolution and is
// nearest-up-sampled on the fly (the FPN top-down path: lateral + interpolate(top, "nearest")).
template <int ACT, bool VEC4, int RES>
__global__ __launch_bounds__(256) void bias_act_kernel(const float* x, float* dst, long long dst_batch_stride, const float* __restrict__ bias,
                                                       const float* __restrict__ residual, int C, int HW, int W, float p0, float p1) {
  const int plane = blockIdx.y;  // b * C + c
  const int b = plane / C, c = plane - b * C;
  const float bv = bias ? bias[c] : 0.0f;
  const float* p = x + static_cast<size_t>(plane) * HW;
  float* d = dst + static_cast<size_t>(b) * dst_batch_stride + static_cast<size_t>(c) * HW;
  if (VEC4 && RES != 2) {
    typedef float float4v __attribute__((ext_vector_type(4)));
    const float4v* p4 = reinterpret_cast<const float4v*>(p);
    float4v* d4 = reinterpret_cast<float4v*>(d);
    const float4v* r4 = reinterpret_cast<const float4v*>(residual + (RES == 1 ? static_cast<size_t>(plane) * HW : 0));
    const int n4 = HW / 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
      float4v v = p4[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = apply_act<ACT>(v[e] + bv, p0, p1);
      if (RES == 1) {
        const float4v r = r4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
      }
      d4[i] = v;
    }
  } else {
    const int Wh = W / 2;
    const float* r = RES == 1 ? residual + static_cast<size_t>(plane) * HW
                              : (RES == 2 ? residual + static_cast<size_t>(plane) * (HW / 4) : nullptr);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
      float v = apply_act<ACT>(p[i] + bv, p0, p1);
      if (RES == 1) v += r[i];
      if (RES == 2) {
        const int y = i / W, xx = i - y * W;
        v += r[(y >> 1) * Wh + (xx >> 1)];
      }
      d[i] = v;
    }
  }
}

// 2x bilinear up-sampling, align_corners=True, in ATen's op order:
//   src = dst * (in - 1) / (out - 1);  i0 = int(src);  l1 = src - i0;  l0 = 1 - l1
//   out = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11)
// Batch item b of the destination starts at out + b * out_batch_stride (a channel slice of a concatenation buffer), its C planes are dense.
// PRE: the input is a raw convolution output and act(in + pre_bias[c]) is applied to each of the four taps on the fly (the decoder's
// coarse depth heads: convolution -> bias + sigmoid -> x2 up-sampling into the next block's input, one launch instead of two).
template <int PRE>
__global__ __launch_bounds__(256) void upsample2x_kernel(const float* __restrict__ in, float* __restrict__ out, long long out_batch_stride,
                                                         const float* __restrict__ pre_bias, int planes, int C, int H, int W) {
  // No FMA contraction: the fractional weight must come from the ROUNDED product sh * oy, the same value whose integer part
  // selects the tap (ATen does exactly that); fma(sh, oy, -y0) would mix a rounded index with an unrounded fraction.
#pragma clang fp contract(off)
  const int OH = 2 * H, OW = 2 * W;
  const float sh = OH > 1 ? static_cast<float>(H - 1) / static_cast<float>(OH - 1) : 0.0f;
  const float sw = OW > 1 ? static_cast<float>(W - 1) / static_cast<float>(OW - 1) : 0.0f;
  const long long total = static_cast<long long>(planes) * OH * OW;
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < total;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int ox = static_cast<int>(i % OW);
    const int oy = static_cast<int>((i / OW) % OH);
    const long long pl = i / (static_cast<long long>(OW) * OH);
    const float fy = sh * static_cast<float>(oy), fx = sw * static_cast<float>(ox);
    const int y0 = static_cast<int>(fy), x0 = static_cast<int>(fx);
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float h1 = fy - static_cast<float>(y0), h0 = 1.0f - h1;
    const float w1 = fx - static_cast<float>(x0), w0 = 1.0f - w1;
    const float* p = in + pl * H * W;
    const long long b = pl / C, c = pl - b * C;
    const float bv = (PRE != 0 && pre_bias) ? pre_bias[c] : 0.0f;
    const float v00 = PRE ? apply_act<PRE>(p[y0 * W + x0] + bv) : p[y0 * W + x0], v01 = PRE ? apply_act<PRE>(p[y0 * W + x1] + bv) : p[y0 * W + x1];
    const float v10 = PRE ? apply_act<PRE>(p[y1 * W + x0] + bv) : p[y1 * W + x0], v11 = PRE ? apply_act<PRE>(p[y1 * W + x1] + bv) : p[y1 * W + x1];
    out[b * out_batch_stride + (c * OH + oy) * OW + ox] = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11);
  }
}

// The same values, four horizontally adjacent outputs per thread (round 6): one plane per blockIdx.y, 32-bit index arithmetic (the kernel above
// spends three 64-bit divisions per output: 15 us for the decoder's last 32 x 128 x 160 -> 256 x 320 map), one 16-byte store.  Every output
// is computed by the expression above from the same rounded products, so the two kernels agree bit for bit; taken when the rows allow
// aligned float4 stores (OW % 4 == 0 always; destination and batch stride 16-byte aligned).
template <int PRE>
__device__ inline void upsample2x_quads_plane(const float* __restrict__ in, float* __restrict__ out, long long out_batch_stride, const float* __restrict__ pre_bias,
                                              int pl, int C, int H, int W) {
#pragma clang fp contract(off)
  typedef float float4v __attribute__((ext_vector_type(4)));
  const int OH = 2 * H, OW = 2 * W, QW = OW >> 2;
  const float sh = OH > 1 ? static_cast<float>(H - 1) / static_cast<float>(OH - 1) : 0.0f;
  const float sw = OW > 1 ? static_cast<float>(W - 1) / static_cast<float>(OW - 1) : 0.0f;
  const int b = pl / C, c = pl - b * C;
  const float* p = in + static_cast<size_t>(pl) * H * W;
  const float bv = (PRE != 0 && pre_bias) ? pre_bias[c] : 0.0f;
  float* dst = out + static_cast<size_t>(b) * out_batch_stride + static_cast<size_t>(c) * OH * OW;
  for (int q = blockIdx.x * 256 + threadIdx.x; q < OH * QW; q += gridDim.x * 256) {
    const int oy = q / QW, ox = 4 * (q - oy * QW);
    const float fy = sh * static_cast<float>(oy);
    const int y0 = static_cast<int>(fy);
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0);
    const float h1 = fy - static_cast<float>(y0), h0 = 1.0f - h1;
    const float* r0 = p + y0 * W;
    const float* r1 = p + y1 * W;
    float4v v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float fx = sw * static_cast<float>(ox + e);
      const int x0 = static_cast<int>(fx);
      const int x1 = x0 + (x0 < W - 1 ? 1 : 0);
      const float w1 = fx - static_cast<float>(x0), w0 = 1.0f - w1;
      const float v00 = PRE ? apply_act<PRE>(r0[x0] + bv) : r0[x0], v01 = PRE ? apply_act<PRE>(r0[x1] + bv) : r0[x1];
      const float v10 = PRE ? apply_act<PRE>(r1[x0] + bv) : r1[x0], v11 = PRE ? apply_act<PRE>(r1[x1] + bv) : r1[x1];
      v[e] = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11);
    }
    *reinterpret_cast<float4v*>(dst + oy * OW + ox) = v;
  }
}

template <int PRE>
__global__ __launch_bounds__(256) void upsample2x_quads_kernel(const float* __restrict__ in, float* __restrict__ out, long long out_batch_stride,
                                                               const float* __restrict__ pre_bias, int C, int H, int W) {
  upsample2x_quads_plane<PRE>(in, out, out_batch_stride, pre_bias, blockIdx.y, C, H, W);
}

// Two up-samplings of maps of the same size in ONE launch (round 6): planes [0, B*C1) are job 1 (no pre-activation), the rest job 2 (PRE2) -- a decoder
// level's feature map and its one-channel depth head (raw convolution output -> bias + sigmoid on the taps), which the frame launched back to back.
template <int PRE2>
__global__ __launch_bounds__(256) void upsample2x_pair_quads_kernel(const float* __restrict__ in1, float* __restrict__ out1, long long out1_batch_stride, int C1,
                                                                    const float* __restrict__ in2, float* __restrict__ out2, long long out2_batch_stride,
                                                                    const float* __restrict__ pre_bias2, int C2, int planes1, int H, int W) {
  const int pl = blockIdx.y;
  if (pl < planes1) upsample2x_quads_plane<0>(in1, out1, out1_batch_stride, nullptr, pl, C1, H, W);
  else upsample2x_quads_plane<PRE2>(in2, out2, out2_batch_stride, pre_bias2, pl - planes1, C2, H, W);
}

// Depthwise k x k convolution (groups == channels, "same" padding k/2, stride 1 or 2) with the bias add and activation
// fused.  MIOpen runs these MnasNet layers through its naive reference kernel (naive_conv_ab_nonpacked_fwd_nchw) plus a
// separate bias/activation launch; a direct kernel is one launch and reads each input element from L1 k*k times.
// One thread per output element; the k*k weights of the channel are wave-uniform (scalar loads).
// PRE: the input is the raw output of the preceding 1x1 expansion convolution and its epilogue -- bias add + ReLU -- is applied
// on the fly to every in-bounds tap (the zero padding stays zero, as it would be on the activated map): one launch per inverted
// residual block less than convolution -> epilogue -> depthwise.
template <int K, int ACT, bool PRE>
__global__ __launch_bounds__(256) void depthwise_conv_kernel(const float* __restrict__ in, const float* __restrict__ weight,
                                                             const float* __restrict__ bias, const float* __restrict__ pre_bias,
                                                             float* __restrict__ out, int C, int H, int W, int OH, int OW, int stride) {
  const int c = blockIdx.y, b = blockIdx.z;
  const float* wk = weight + static_cast<size_t>(c) * K * K;
  const float* src = in + (static_cast<size_t>(b) * C + c) * H * W;
  float* dst = out + (static_cast<size_t>(b) * C + c) * OH * OW;
  const float bv = bias ? bias[c] : 0.0f;
  const float pv = (PRE && pre_bias) ? pre_bias[c] : 0.0f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < OH * OW; i += gridDim.x * blockDim.x) {
    const int oy = i / OW, ox = i - oy * OW;
    const int y0 = oy * stride - K / 2, x0 = ox * stride - K / 2;
    float acc = 0.0f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int y = y0 + ky;
      const bool yin = y >= 0 && y < H;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int x = x0 + kx;
        float v = 0.0f;
        if (yin && x >= 0 && x < W) v = PRE ? fmaxf(src[y * W + x] + pv, 0.0f) : src[y * W + x];
        acc = fmaf(v, wk[ky * K + kx], acc);
      }
    }
    dst[i] = apply_act<ACT>(acc + bv);
  }
}

template <int K, int ACT>
int launch_depthwise(const float* in, const float* w, const float* b, const float* pre_bias, bool pre, float* out, int B, int C, int H, int W,
                     int OH, int OW, int stride, hipStream_t s) {
  dim3 grid(max(1, min((OH * OW + 255) / 256, 64)), C, B), block(256);
  if (pre) hipLaunchKernelGGL((depthwise_conv_kernel<K, ACT, true>), grid, block, 0, s, in, w, b, pre_bias, out, C, H, W, OH, OW, stride);
  else hipLaunchKernelGGL((depthwise_conv_kernel<K, ACT, false>), grid, block, 0, s, in, w, b, pre_bias, out, C, H, W, OH, OW, stride);
  return launch_status();
}

template <int ACT>
int launch_bias_act(const float* x, float* dst, long long dst_batch_stride, const float* bias, const float* residual, int residual_mode,
                    int B, int C, int H, int W, float p0, float p1, hipStream_t s) {
  const int HW = H * W;
  const bool aligned = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(residual) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(dst) % 16 == 0) && (dst_batch_stride % 4 == 0);
  const bool vec4 = (HW % 4 == 0) && aligned && residual_mode != 2;
  const int work = vec4 ? HW / 4 : HW;
  dim3 grid(max(1, min((work + 255) / 256, 64)), B * C), block(256);
#define DVMVS_BA(V, R) hipLaunchKernelGGL((bias_act_kernel<ACT, V, R>), grid, block, 0, s, x, dst, dst_batch_stride, bias, residual, C, HW, W, p0, p1)
  if (residual_mode == 0) { if (vec4) DVMVS_BA(true, 0); else DVMVS_BA(false, 0); }
  else if (residual_mode == 1) { if (vec4) DVMVS_BA(true, 1); else DVMVS_BA(false, 1); }
  else DVMVS_BA(false, 2);
#undef DVMVS_BA
  return launch_status();
}

// [B, C, H*W] -> [B, H*W, C] for C <= 64, a multiple of 4: the measurement maps of the correlate-then-interpolate sweep are read one
// 128-byte line per cell, so a keyframe's features are kept channels-last in the engine's feature cache (one such copy per keyframe:
// this kernel instead of the contiguous copy the cache made before).  64 pixels x C channels per workgroup through LDS: reads coalesced
// along the pixels, writes as float4 along the channels.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW) {
  __shared__ float tile[64][65];
  const int b = blockIdx.y, p0 = blockIdx.x * 64;
  const float* s = src + static_cast<size_t>(b) * C * HW;
  float* d = dst + static_cast<size_t>(b) * C * HW;
  for (int i = threadIdx.x; i < C * 64; i += 256) {
    const int c = i >> 6, p = i & 63;
    tile[p][c] = (p0 + p < HW) ? s[static_cast<size_t>(c) * HW + p0 + p] : 0.0f;
  }
  __syncthreads();
  const int quads = C >> 2;
  for (int i = threadIdx.x; i < quads * 64; i += 256) {
    const int p = i / quads, c4 = (i - p * quads) * 4;
    if (p0 + p < HW) {
      float4 v;
      v.x = tile[p][c4]; v.y = tile[p][c4 + 1]; v.z = tile[p][c4 + 2]; v.w = tile[p][c4 + 3];
      *reinterpret_cast<float4*>(d + static_cast<size_t>(p0 + p) * C + c4) = v;
    }
  }
}

// Up to eight contiguous device-to-device copies in ONE launch (blockIdx.y = the copy).  A frame step makes four small copies before its
// graph -- the keyframe's features into the feature cache, the measurement maps into the sweep's buffers, the next image into its home --
// 0.6 - 1 MB each: as four runtime copies they are four 5 us launches in front of the sweep; as one launch they are one.
struct CopyBatchArgs {
  const float4* src[8];
  float4* dst[8];
  unsigned int quads[8];
};

__global__ __launch_bounds__(256) void copy_batch_kernel(CopyBatchArgs a) {
  const int j = blockIdx.y;
  const float4* __restrict__ s = a.src[j];
  float4* __restrict__ d = a.dst[j];
  const unsigned int n = a.quads[j];
  for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) d[i] = s[i];
}

}  // namespace dvmvs

extern "C" int dvmvs_bias_act_fwd(const float* x, float* dst, long long dst_batch_stride, const float* bias, const float* residual,
                                  int residual_mode, int B, int C, int H, int W, int activation, float p0, float p1,
                                  dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!x || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0) return DVMVS_EINVAL;
  if (residual_mode < 0 || residual_mode > 2 || (residual_mode != 0 && !residual)) return DVMVS_EINVAL;
  if (dst_batch_stride < static_cast<long long>(C) * H * W) return DVMVS_EINVAL;
  if (residual_mode == 2 && ((H & 1) || (W & 1))) return DVMVS_EUNSUPPORTED;
  if (static_cast<long long>(B) * C > 65535LL) return DVMVS_EUNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (activation) {
    case 0: return launch_bias_act<0>(x, dst, dst_batch_stride, bias, residual, residual_mode, B, C, H, W, p0, p1, s);
    case 1: return launch_bias_act<1>(x, dst, dst_batch_stride, bias, residual, residual_mode, B, C, H, W, p0, p1, s);
    case 2: return launch_bias_act<2>(x, dst, dst_batch_stride, bias, residual, residual_mode, B, C, H, W, p0, p1, s);
    case 3: return launch_bias_act<3>(x, dst, dst_batch_stride, bias, residual, residual_mode, B, C, H, W, p0, p1, s);
    default: return DVMVS_EINVAL;
  }
}

extern "C" int dvmvs_bias_act_inplace(float* x, const float* bias, const float* residual, int residual_mode, int B, int C, int H,
                                      int W, int activation, dvmvs_stream_t stream) {
  if (activation == 3) return DVMVS_EINVAL;   // the depth mapping has parameters: dvmvs_bias_act_fwd
  return dvmvs_bias_act_fwd(x, x, static_cast<long long>(C) * H * W, bias, residual, residual_mode, B, C, H, W, activation, 0.0f, 0.0f, stream);
}

extern "C" int dvmvs_upsample2x_fwd(const float* in, float* out, long long out_batch_stride, const float* pre_bias, int pre_activation,
                                    int B, int C, int H, int W, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!in || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return DVMVS_EINVAL;
  if (pre_activation < 0 || pre_activation > 2) return DVMVS_EINVAL;
  if (out_batch_stride == 0) out_batch_stride = static_cast<long long>(C) * H * W * 4;
  if (out_batch_stride < static_cast<long long>(C) * H * W * 4) return DVMVS_EINVAL;
  const long long total = static_cast<long long>(B) * C * H * W * 4;
  long long blocks = (total + 255) / 256;
  if (blocks > 256LL * 16) blocks = 256LL * 16;
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (W % 2 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (out_batch_stride & 3) == 0 && static_cast<long long>(B) * C <= 65535 &&
      static_cast<long long>(H) * W < (1LL << 27)) {
    const int quads = 2 * H * (W / 2);
    const int per_plane = max(1, min((quads + 255) / 256, max(1, 4096 / (B * C))));
    const dim3 qgrid(static_cast<unsigned>(per_plane), static_cast<unsigned>(B * C));
    if (pre_activation == 1) hipLaunchKernelGGL(upsample2x_quads_kernel<1>, qgrid, block, 0, s, in, out, out_batch_stride, pre_bias, C, H, W);
    else if (pre_activation == 2) hipLaunchKernelGGL(upsample2x_quads_kernel<2>, qgrid, block, 0, s, in, out, out_batch_stride, pre_bias, C, H, W);
    else hipLaunchKernelGGL(upsample2x_quads_kernel<0>, qgrid, block, 0, s, in, out, out_batch_stride, pre_bias, C, H, W);
    return launch_status();
  }
  if (pre_activation == 1) hipLaunchKernelGGL(upsample2x_kernel<1>, grid, block, 0, s, in, out, out_batch_stride, pre_bias, B * C, C, H, W);
  else if (pre_activation == 2) hipLaunchKernelGGL(upsample2x_kernel<2>, grid, block, 0, s, in, out, out_batch_stride, pre_bias, B * C, C, H, W);
  else hipLaunchKernelGGL(upsample2x_kernel<0>, grid, block, 0, s, in, out, out_batch_stride, pre_bias, B * C, C, H, W);
  return launch_status();
}

extern "C" int dvmvs_upsample2x_pair_fwd(const float* in1, float* out1, long long out1_batch_stride, int C1, const float* in2, float* out2,
                                         long long out2_batch_stride, const float* pre_bias2, int pre_activation2, int C2, int B, int H, int W,
                                         dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!in1 || !out1 || !in2 || !out2 || B <= 0 || C1 <= 0 || C2 <= 0 || H <= 0 || W <= 0) return DVMVS_EINVAL;
  if (pre_activation2 < 0 || pre_activation2 > 2) return DVMVS_EINVAL;
  if (out1_batch_stride == 0) out1_batch_stride = static_cast<long long>(C1) * H * W * 4;
  if (out2_batch_stride == 0) out2_batch_stride = static_cast<long long>(C2) * H * W * 4;
  if (out1_batch_stride < static_cast<long long>(C1) * H * W * 4 || out2_batch_stride < static_cast<long long>(C2) * H * W * 4) return DVMVS_EINVAL;
  const long long planes = static_cast<long long>(B) * (C1 + C2);
  const bool quads = W % 2 == 0 && ((reinterpret_cast<uintptr_t>(out1) | reinterpret_cast<uintptr_t>(out2)) & 15) == 0 && ((out1_batch_stride | out2_batch_stride) & 3) == 0 &&
                     planes <= 65535 && static_cast<long long>(H) * W < (1LL << 27);
  if (!quads) {      // (destinations that do not allow 16-byte stores: the two launches)
    const int rc = dvmvs_upsample2x_fwd(in1, out1, out1_batch_stride, nullptr, 0, B, C1, H, W, stream);
    return rc != 0 ? rc : dvmvs_upsample2x_fwd(in2, out2, out2_batch_stride, pre_bias2, pre_activation2, B, C2, H, W, stream);
  }
  const int n_quads = 2 * H * (W / 2);
  const int per_plane = max(1, min((n_quads + 255) / 256, max(1, 4096 / static_cast<int>(planes))));
  const dim3 grid(static_cast<unsigned>(per_plane), static_cast<unsigned>(planes)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define DVMVS_UP2(P) hipLaunchKernelGGL(upsample2x_pair_quads_kernel<P>, grid, block, 0, s, in1, out1, out1_batch_stride, C1, in2, out2, out2_batch_stride, pre_bias2, C2, B * C1, H, W)
  if (pre_activation2 == 1) DVMVS_UP2(1); else if (pre_activation2 == 2) DVMVS_UP2(2); else DVMVS_UP2(0);
#undef DVMVS_UP2
  return launch_status();
}

extern "C" int dvmvs_depthwise_conv_fwd(const float* in, const float* weight, const float* bias, const float* pre_bias, int pre_relu,
                                        float* out, int B, int C, int H, int W, int kernel_size, int stride, int activation,
                                        dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!in || !weight || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return DVMVS_EINVAL;
  if ((kernel_size != 3 && kernel_size != 5) || (stride != 1 && stride != 2) || activation < 0 || activation > 2) return DVMVS_EUNSUPPORTED;
  if (C > 65535 || B > 65535) return DVMVS_EUNSUPPORTED;
  const int pad = kernel_size / 2;
  const int OH = (H + 2 * pad - kernel_size) / stride + 1, OW = (W + 2 * pad - kernel_size) / stride + 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define DVMVS_DW(K, A) return launch_depthwise<K, A>(in, weight, bias, pre_bias, pre_relu != 0, out, B, C, H, W, OH, OW, stride, s)
  if (kernel_size == 3) {
    if (activation == 0) DVMVS_DW(3, 0);
    if (activation == 1) DVMVS_DW(3, 1);
    DVMVS_DW(3, 2);
  }
  if (activation == 0) DVMVS_DW(5, 0);
  if (activation == 1) DVMVS_DW(5, 1);
  DVMVS_DW(5, 2);
#undef DVMVS_DW
}

extern "C" int dvmvs_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!src || !dst || src == dst) return DVMVS_EINVAL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || B > 65535) return DVMVS_EINVAL;
  if (C > 64 || C % 4 != 0) return DVMVS_EUNSUPPORTED;
  const int HW = H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((HW + 63) / 64, B), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst, C, HW);
  return launch_status();
}

extern "C" int dvmvs_copy_batch(const float* const* srcs, float* const* dsts, const long long* n_floats, int n, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!srcs || !dsts || !n_floats || n <= 0 || n > 8) return DVMVS_EINVAL;
  CopyBatchArgs a = {};
  unsigned int most = 0;
  for (int j = 0; j < n; ++j) {
    if (!srcs[j] || !dsts[j] || n_floats[j] <= 0 || n_floats[j] % 4 != 0 || n_floats[j] > (1LL << 33)) return DVMVS_EINVAL;
    if ((reinterpret_cast<uintptr_t>(srcs[j]) | reinterpret_cast<uintptr_t>(dsts[j])) & 15u) return DVMVS_EINVAL;      // 16-byte aligned
    a.src[j] = reinterpret_cast<const float4*>(srcs[j]);
    a.dst[j] = reinterpret_cast<float4*>(dsts[j]);
    a.quads[j] = static_cast<unsigned int>(n_floats[j] / 4);
    most = a.quads[j] > most ? a.quads[j] : most;
  }
  const unsigned int blocks = (most + 1023u) / 1024u < 256u ? ((most + 1023u) / 1024u ? (most + 1023u) / 1024u : 1u) : 256u;      // <= 4 quads per thread, <= one block per CU per copy
  hipLaunchKernelGGL(copy_batch_kernel, dim3(blocks, n), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return launch_status();
}
