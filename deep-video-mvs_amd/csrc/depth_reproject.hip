// Forward splat of the previous depth map into the current half-resolution view (z-buffer, farthest wins), gfx950.
//
// The reference sorts all points by relu(z) descending, projects, rounds, and keeps the first point per target
// pixel through a host round trip (argsort + gather + ij.cpu().numpy() -> np.unique -> .cuda() + index_put_):
// /root/reference/dvmvs/utils.py:110-154.  "First after a descending sort" is "maximum relu(z) among the points
// that land on the pixel", and relu(z) >= 0, so its IEEE bit pattern orders like an unsigned integer: one
// atomicMax per source pixel reproduces the result exactly, in any execution order, with no sort and no sync.
#include "dvmvs_device.h"

namespace dvmvs {

constexpr float kReprojEps = 1e-8f;

// `trans` = inverse(reference_pose) * measurement_pose (utils.py:121) is the caller's (ABI 3): the reference's own fp32 matrix by
// default, dvmvs_relative_pose's fp64 one in "exact" mode.  Uniform per batch item: the loads below are scalar (SGPR) loads.
__global__ __launch_bounds__(256) void depth_splat_kernel(const float* __restrict__ trans, const float* __restrict__ prev_depth,
                                                          const float* __restrict__ full_K, const float* __restrict__ half_K,
                                                          unsigned int* __restrict__ out_bits, int B, int Hf, int Wf) {
#pragma clang fp contract(off)
  const int b = blockIdx.y;
  const float* s_T = trans + b * 16;
  const int HWf = Hf * Wf;
  const int hw = Wf / 2, hh = Hf / 2;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= HWf) return;
  const int y = pix / Wf, x = pix - y * Wf;
  const float* Kf = full_K + b * 9;
  const float* Kh = half_K + b * 9;
  const float d = prev_depth[static_cast<size_t>(b) * HWf + pix];
  // kornia.depth_to_3d with the full-resolution intrinsics
  const float px = ((static_cast<float>(x) - Kf[2]) / Kf[0]) * d;
  const float py = ((static_cast<float>(y) - Kf[5]) / Kf[4]) * d;
  const float pz = d;
  float q[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) q[r] = ((s_T[r * 4 + 0] * px + s_T[r * 4 + 1] * py) + s_T[r * 4 + 2] * pz) + s_T[r * 4 + 3];
  const float sw = fabsf(q[3]) > kReprojEps ? 1.0f / q[3] : 1.0f;
  const float X = sw * q[0], Y = sw * q[1], Z = sw * q[2];
  const float z_store = fmaxf(Z, 0.0f);                       // value written: relu(z)        (utils.py:129)
  const float sz = fabsf(Z) > kReprojEps ? 1.0f / Z : 1.0f;   // projection uses the raw z    (utils.py:134-136)
  const float u = rintf((X * sz) * Kh[0] + Kh[2]);            // torch.round = half-to-even
  const float v = rintf((Y * sz) * Kh[4] + Kh[5]);
  if (!(u >= 0.0f && v >= 0.0f && u < static_cast<float>(hw) && v < static_cast<float>(hh))) return;  // also drops NaN
  if (!(z_store > 0.0f)) return;                               // 0 (or NaN) never beats the zero-initialised buffer
  atomicMax(out_bits + static_cast<size_t>(b) * hh * hw + static_cast<int>(v) * hw + static_cast<int>(u), __float_as_uint(z_store));
}

// Frame path (round 5): the ConvLSTM reads only rows / columns 0, f, 2f, ... of the half-resolution estimate
// (F.interpolate(scale_factor=1/f, mode="nearest"), fusionnet/run-testing.py:176-189), so only source points that land on those
// pixels matter: the splat goes straight into the [Ho, Wo] estimate (same atomicMax on the same relu(z) bits: the value that the
// half-resolution z-buffer would hold there), and the launch clears the OTHER estimate buffer for the frame after next -- frames
// alternate between two buffers, so there is neither a half-resolution z-buffer nor a decimate / clear launch.
__global__ __launch_bounds__(256) void depth_splat_estimate_kernel(const float* __restrict__ trans, const float* __restrict__ prev_depth,
                                                                   const float* __restrict__ full_K, const float* __restrict__ half_K,
                                                                   unsigned int* __restrict__ estimate_bits, float* __restrict__ clear,
                                                                   int B, int Hf, int Wf, int f) {
#pragma clang fp contract(off)
  const int b = blockIdx.y;
  const float* s_T = trans + b * 16;
  const int HWf = Hf * Wf;
  const int hw = Wf / 2, hh = Hf / 2;
  const int wo = hw / f, ho = hh / f;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (clear != nullptr && pix < ho * wo) clear[static_cast<size_t>(b) * ho * wo + pix] = 0.0f;
  if (pix >= HWf) return;
  const int y = pix / Wf, x = pix - y * Wf;
  const float* Kf = full_K + b * 9;
  const float* Kh = half_K + b * 9;
  const float d = prev_depth[static_cast<size_t>(b) * HWf + pix];
  // (the arithmetic of depth_splat_kernel, operation for operation)
  const float px = ((static_cast<float>(x) - Kf[2]) / Kf[0]) * d;
  const float py = ((static_cast<float>(y) - Kf[5]) / Kf[4]) * d;
  const float pz = d;
  float q[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) q[r] = ((s_T[r * 4 + 0] * px + s_T[r * 4 + 1] * py) + s_T[r * 4 + 2] * pz) + s_T[r * 4 + 3];
  const float sw = fabsf(q[3]) > kReprojEps ? 1.0f / q[3] : 1.0f;
  const float X = sw * q[0], Y = sw * q[1], Z = sw * q[2];
  const float z_store = fmaxf(Z, 0.0f);
  const float sz = fabsf(Z) > kReprojEps ? 1.0f / Z : 1.0f;
  const float u = rintf((X * sz) * Kh[0] + Kh[2]);
  const float v = rintf((Y * sz) * Kh[4] + Kh[5]);
  if (!(u >= 0.0f && v >= 0.0f && u < static_cast<float>(hw) && v < static_cast<float>(hh))) return;
  if (!(z_store > 0.0f)) return;
  const int ui = static_cast<int>(u), vi = static_cast<int>(v);
  if (ui % f != 0 || vi % f != 0 || ui / f >= wo || vi / f >= ho) return;   // not a pixel the nearest-neighbour decimation picks
  atomicMax(estimate_bits + (static_cast<size_t>(b) * ho + vi / f) * wo + ui / f, __float_as_uint(z_store));
}

// Zero-fill as an ordinary kernel node.  hipMemsetAsync is avoided on purpose: captured into a hipGraph it becomes a
// memset node, and on ROCm 7.x replays of such a graph were observed to run the splat kernel against a buffer that was
// cleared late (all-zero output whenever the device was idle at launch); kernel -> kernel edges do not have the problem.
__global__ void zero_fill_kernel(float* __restrict__ p, long long n) {
  for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * blockDim.x)
    p[i] = 0.0f;
}

// F.interpolate(scale_factor=1/f, mode="nearest") for an integer factor: picks rows / columns 0, f, 2f, ...
__global__ void nearest_decimate_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W, int f) {
  const int Ho = H / f, Wo = W / f;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Ho * Wo) return;
  const int xo = i % Wo, yo = (i / Wo) % Ho, b = i / (Wo * Ho);
  out[i] = in[(static_cast<size_t>(b) * H + yo * f) * W + xo * f];
}

// The frame path needs only the decimated estimate: one kernel picks rows / columns 0, f, 2f, ... of the z-buffer AND clears it,
// so that the buffer is all-zero again when the next frame splats into it -- no zero-fill launch per frame.
__global__ void decimate_clear_kernel(float* __restrict__ zbuffer, float* __restrict__ out, int B, int H, int W, int f) {
  const int Ho = H / f, Wo = W / f;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H * W) return;
  const int x = i % W, y = (i / W) % H, b = i / (W * H);
  const float v = zbuffer[i];
  if (y % f == 0 && x % f == 0 && y / f < Ho && x / f < Wo) out[(static_cast<size_t>(b) * Ho + y / f) * Wo + x / f] = v;
  zbuffer[i] = 0.0f;
}

}  // namespace dvmvs

extern "C" int dvmvs_depth_reproject_lowres_fwd(const float* transformation, const float* previous_depth, const float* full_K,
                                                const float* half_K, float* zbuffer, float* out_lowres, int lowres_factor,
                                                int B, int full_height, int full_width, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!transformation || !previous_depth || !full_K || !half_K || !zbuffer || !out_lowres) return DVMVS_EINVAL;
  if (B <= 0 || full_height < 2 || full_width < 2 || B > 65535 || lowres_factor <= 0) return DVMVS_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int hh = full_height / 2, hw = full_width / 2;
  if (hh / lowres_factor <= 0 || hw / lowres_factor <= 0) return DVMVS_EINVAL;
  const int HWf = full_height * full_width;
  hipLaunchKernelGGL(depth_splat_kernel, dim3((HWf + 255) / 256, B), dim3(256), 0, s, transformation, previous_depth, full_K, half_K,
                     reinterpret_cast<unsigned int*>(zbuffer), B, full_height, full_width);
  int rc = launch_status();
  if (rc != 0) return rc;
  const int n = B * hh * hw;
  hipLaunchKernelGGL(decimate_clear_kernel, dim3((n + 255) / 256), dim3(256), 0, s, zbuffer, out_lowres, B, hh, hw, lowres_factor);
  return launch_status();
}

extern "C" int dvmvs_depth_reproject_estimate_fwd(const float* transformation, const float* previous_depth, const float* full_K,
                                                  const float* half_K, float* estimate, float* estimate_to_clear, int lowres_factor,
                                                  int B, int full_height, int full_width, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!transformation || !previous_depth || !full_K || !half_K || !estimate || estimate == estimate_to_clear) return DVMVS_EINVAL;
  if (B <= 0 || full_height < 2 || full_width < 2 || B > 65535 || lowres_factor <= 0) return DVMVS_EINVAL;
  const int hh = full_height / 2, hw = full_width / 2;
  if (hh / lowres_factor <= 0 || hw / lowres_factor <= 0) return DVMVS_EINVAL;
  const int HWf = full_height * full_width;
  hipLaunchKernelGGL(depth_splat_estimate_kernel, dim3((HWf + 255) / 256, B), dim3(256), 0, static_cast<hipStream_t>(stream), transformation,
                     previous_depth, full_K, half_K, reinterpret_cast<unsigned int*>(estimate), estimate_to_clear, B, full_height, full_width,
                     lowres_factor);
  return launch_status();
}

extern "C" int dvmvs_depth_reproject_fwd(const float* transformation, const float* previous_depth, const float* full_K,
                                         const float* half_K, float* out, float* out_lowres, int lowres_factor,
                                         int B, int full_height, int full_width, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!transformation || !previous_depth || !full_K || !half_K || !out) return DVMVS_EINVAL;
  if (B <= 0 || full_height < 2 || full_width < 2 || B > 65535) return DVMVS_EINVAL;
  if (out_lowres && lowres_factor <= 0) return DVMVS_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int hh = full_height / 2, hw = full_width / 2;
  const long long n_out = static_cast<long long>(B) * hh * hw;
  hipLaunchKernelGGL(zero_fill_kernel, dim3(static_cast<unsigned>((n_out + 255) / 256 < 1024 ? (n_out + 255) / 256 : 1024)), dim3(256), 0, s, out, n_out);
  int rc = launch_status();
  if (rc != 0) return rc;
  const int HWf = full_height * full_width;
  hipLaunchKernelGGL(depth_splat_kernel, dim3((HWf + 255) / 256, B), dim3(256), 0, s, transformation,
                     previous_depth, full_K, half_K, reinterpret_cast<unsigned int*>(out), B, full_height, full_width);
  rc = launch_status();
  if (rc != 0) return rc;
  if (out_lowres) {
    const int n = B * (hh / lowres_factor) * (hw / lowres_factor);
    if (n > 0) {
      hipLaunchKernelGGL(nearest_decimate_kernel, dim3((n + 255) / 256), dim3(256), 0, s, out, out_lowres, B, hh, hw, lowres_factor);
      rc = launch_status();
    }
  }
  return rc;
}
