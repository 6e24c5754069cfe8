// Backward kernels of the two frame-path ops that training (BASELINE.json configs[4]) otherwise leaves to slow generic paths, gfx950:
//   * x2 bilinear up-sampling (align_corners): ATen's forward takes 130 us per call on these maps (81 calls per training step =
//     10.5 ms, profiles/r03_train_timed_region.csv) and its backward scatters with atomics; dvmvs_upsample2x_fwd has been the inference
//     path since round 2 -- this file adds its adjoint as a GATHER (every input pixel sums the <= 5 x 5 output pixels whose taps
//     include it, weights recomputed with the forward's own fp32 expressions), so training can use the HIP forward;
//   * depthwise k x k convolution (MnasNet): MIOpen falls back to naive_conv_ab_nonpacked_{fwd,bwd,wrw}_nchw (24-47 us per call,
//     ~340 calls per step = 9.8 ms); the forward kernel exists (frame_ops.hip), here are the data gradient (gather over the k x k
//     outputs that read an input pixel) and the weight gradient (workgroups per channel x slice, fixed-order reductions).
// No atomics anywhere: gradients are bit-reproducible.  Reference for the ops themselves: /root/reference/dvmvs/fusionnet/model.py:59,114
// (F.interpolate(scale_factor=2, mode='bilinear', align_corners=True)) and torchvision's MnasNet depthwise layers.
#include "dvmvs_device.h"

namespace dvmvs {

// grad_in[pl, y, x] = sum over (oy, ox) of grad_out[pl, oy, ox] * weight of tap (y, x) in output pixel (oy, ox)
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ grad_out, float* __restrict__ grad_in, int planes, int H, int W) {
#pragma clang fp contract(off)
  const int OH = 2 * H, OW = 2 * W;
  const float sh = OH > 1 ? static_cast<float>(H - 1) / static_cast<float>(OH - 1) : 0.0f;
  const float sw = OW > 1 ? static_cast<float>(W - 1) / static_cast<float>(OW - 1) : 0.0f;
  const long long total = static_cast<long long>(planes) * H * W;
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < total; i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int x = static_cast<int>(i % W), y = static_cast<int>((i / W) % H);
    const long long pl = i / (static_cast<long long>(W) * H);
    const float* g = grad_out + pl * OH * OW;
    // output rows whose source position sh * oy lies in (y - 1, y + 1): oy in [2y - 2, 2y + 3] covers them for every H >= 1
    float wy[6], wx[6];
    int oy0 = 2 * y - 2, ox0 = 2 * x - 2;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int oy = oy0 + k, ox = ox0 + k;
      wy[k] = 0.0f;
      wx[k] = 0.0f;
      if (oy >= 0 && oy < OH) {
        const float fy = sh * static_cast<float>(oy);
        const int y0 = static_cast<int>(fy), y1 = y0 + (y0 < H - 1 ? 1 : 0);
        const float h1 = fy - static_cast<float>(y0), h0 = 1.0f - h1;
        wy[k] = (y0 == y ? h0 : 0.0f) + (y1 == y ? h1 : 0.0f);
      }
      if (ox >= 0 && ox < OW) {
        const float fx = sw * static_cast<float>(ox);
        const int x0 = static_cast<int>(fx), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float w1 = fx - static_cast<float>(x0), w0 = 1.0f - w1;
        wx[k] = (x0 == x ? w0 : 0.0f) + (x1 == x ? w1 : 0.0f);
      }
    }
    float acc = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) {
      if (wy[ky] == 0.0f) continue;
      float row = 0.0f;
#pragma unroll
      for (int kx = 0; kx < 6; ++kx)
        if (wx[kx] != 0.0f) row += wx[kx] * g[(oy0 + ky) * OW + (ox0 + kx)];
      acc += wy[ky] * row;
    }
    grad_in[i] = acc;
  }
}

// grad_in[b, c, y, x] = sum_{ky, kx} grad_out[b, c, oy, ox] * w[c, ky, kx]   with  oy * stride - K/2 + ky == y  (ditto x)
template <int K>
__global__ __launch_bounds__(256) void depthwise_bwd_data_kernel(const float* __restrict__ grad_out, const float* __restrict__ weight,
                                                                 float* __restrict__ grad_in, int C, int H, int W, int OH, int OW, int stride) {
  const int c = blockIdx.y, b = blockIdx.z;
  const float* wk = weight + static_cast<size_t>(c) * K * K;
  const float* g = grad_out + (static_cast<size_t>(b) * C + c) * OH * OW;
  float* dst = grad_in + (static_cast<size_t>(b) * C + c) * H * W;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
    const int y = i / W, x = i - y * W;
    float acc = 0.0f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int ty = y + K / 2 - ky;
      if (ty < 0 || ty % stride != 0) continue;
      const int oy = ty / stride;
      if (oy >= OH) continue;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int tx = x + K / 2 - kx;
        if (tx < 0 || tx % stride != 0) continue;
        const int ox = tx / stride;
        if (ox < OW) acc = fmaf(g[oy * OW + ox], wk[ky * K + kx], acc);
      }
    }
    dst[i] = acc;
  }
}

// grad_w[c, ky, kx] = sum_{b, oy, ox} grad_out[b, c, oy, ox] * in[b, c, oy * stride - K/2 + ky, ox * stride - K/2 + kx]
// A workgroup = one channel x one slice of the (b, pixel) range: every thread keeps K*K partial sums over its share, a fixed-order tree
// over LDS reduces the workgroup, and (when there is more than one slice) depthwise_bwd_weight_reduce_kernel adds the slices in order.
// (Round 4, first form: one workgroup per channel -- 48 ... 576 workgroups of 256 threads for 65 536 products each: 58 us per call at
// k = 3, no faster than MIOpen's naive kernel; sliced: the chip is full.)
template <int K>
__global__ __launch_bounds__(256) void depthwise_bwd_weight_kernel(const float* __restrict__ grad_out, const float* __restrict__ in,
                                                                   float* __restrict__ partial, int B, int C, int H, int W, int OH, int OW, int stride) {
  __shared__ float s_part[256];
  const int c = blockIdx.x, slice = blockIdx.y, slices = gridDim.y, tid = threadIdx.x;
  float acc[K * K];
#pragma unroll
  for (int k = 0; k < K * K; ++k) acc[k] = 0.0f;
  const int per_image = OH * OW, total = B * per_image;
  const int chunk = (total + slices - 1) / slices;
  const int begin = slice * chunk, end = min(total, begin + chunk);
  for (int i = begin + tid; i < end; i += 256) {
    const int b = i / per_image, p = i - b * per_image;
    const int oy = p / OW, ox = p - oy * OW;
    const float gv = grad_out[(static_cast<size_t>(b) * C + c) * per_image + p];
    const float* src = in + (static_cast<size_t>(b) * C + c) * H * W;
    const int y0 = oy * stride - K / 2, x0 = ox * stride - K / 2;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int y = y0 + ky;
      const bool yin = y >= 0 && y < H;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int x = x0 + kx;
        const float v = (yin && x >= 0 && x < W) ? src[y * W + x] : 0.0f;
        acc[ky * K + kx] = fmaf(gv, v, acc[ky * K + kx]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < K * K; ++k) {
    s_part[tid] = acc[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (tid < off) s_part[tid] += s_part[tid + off];
      __syncthreads();
    }
    if (tid == 0) partial[(static_cast<size_t>(slice) * C + c) * K * K + k] = s_part[0];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void depthwise_bwd_weight_reduce_kernel(const float* __restrict__ partial, float* __restrict__ grad_w, int n, int slices) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = partial[i];
  for (int s = 1; s < slices; ++s) v += partial[static_cast<size_t>(s) * n + i];
  grad_w[i] = v;
}

}  // namespace dvmvs

extern "C" int dvmvs_upsample2x_bwd(const float* grad_out, float* grad_in, int B, int C, int H, int W, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!grad_out || !grad_in || B <= 0 || C <= 0 || H <= 0 || W <= 0) return DVMVS_EINVAL;
  const long long total = static_cast<long long>(B) * C * H * W;
  long long blocks = (total + 255) / 256;
  if (blocks > 256LL * 16) blocks = 256LL * 16;
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), grad_out, grad_in,
                     B * C, H, W);
  return launch_status();
}

// slices of the (b, pixel) range per channel in the weight gradient: enough workgroups to fill the chip, at least 4096 products each
static int depthwise_weight_slices(int B, int C, int OH, int OW) {
  const long long total = static_cast<long long>(B) * OH * OW;
  long long slices = (1024 + C - 1) / C;
  if (slices > total / 4096) slices = total / 4096;
  return static_cast<int>(slices < 1 ? 1 : (slices > 64 ? 64 : slices));
}

extern "C" size_t dvmvs_depthwise_conv_bwd_workspace_bytes(int B, int C, int H, int W, int kernel_size, int stride) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (kernel_size != 3 && kernel_size != 5) || (stride != 1 && stride != 2)) return 0;
  const int pad = kernel_size / 2;
  const int OH = (H + 2 * pad - kernel_size) / stride + 1, OW = (W + 2 * pad - kernel_size) / stride + 1;
  const int slices = depthwise_weight_slices(B, C, OH, OW);
  return slices > 1 ? sizeof(float) * static_cast<size_t>(slices) * C * kernel_size * kernel_size : 0;
}

extern "C" int dvmvs_depthwise_conv_bwd(const float* grad_out, const float* in, const float* weight, float* grad_in, float* grad_weight,
                                        float* workspace, int B, int C, int H, int W, int kernel_size, int stride, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!grad_out || !in || !weight || B <= 0 || C <= 0 || H <= 0 || W <= 0) return DVMVS_EINVAL;
  if ((kernel_size != 3 && kernel_size != 5) || (stride != 1 && stride != 2)) return DVMVS_EUNSUPPORTED;
  if (C > 65535 || B > 65535) return DVMVS_EUNSUPPORTED;
  const int pad = kernel_size / 2;
  const int OH = (H + 2 * pad - kernel_size) / stride + 1, OW = (W + 2 * pad - kernel_size) / stride + 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (grad_in) {
    const dim3 grid(max(1, min((H * W + 255) / 256, 64)), C, B), block(256);
    if (kernel_size == 3) hipLaunchKernelGGL(depthwise_bwd_data_kernel<3>, grid, block, 0, s, grad_out, weight, grad_in, C, H, W, OH, OW, stride);
    else hipLaunchKernelGGL(depthwise_bwd_data_kernel<5>, grid, block, 0, s, grad_out, weight, grad_in, C, H, W, OH, OW, stride);
    const int rc = launch_status();
    if (rc != 0) return rc;
  }
  if (grad_weight) {
    const int slices = depthwise_weight_slices(B, C, OH, OW);
    if (slices > 1 && !workspace) return DVMVS_EINVAL;      // dvmvs_depthwise_conv_bwd_workspace_bytes() bytes of scratch
    float* partial = slices > 1 ? workspace : grad_weight;
    const dim3 grid(C, slices);
    if (kernel_size == 3) hipLaunchKernelGGL(depthwise_bwd_weight_kernel<3>, grid, dim3(256), 0, s, grad_out, in, partial, B, C, H, W, OH, OW, stride);
    else hipLaunchKernelGGL(depthwise_bwd_weight_kernel<5>, grid, dim3(256), 0, s, grad_out, in, partial, B, C, H, W, OH, OW, stride);
    int rc = launch_status();
    if (rc != 0 || slices == 1) return rc;
    const int n = C * kernel_size * kernel_size;
    hipLaunchKernelGGL(depthwise_bwd_weight_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, partial, grad_weight, n, slices);
    return launch_status();
  }
  return 0;
}
