// ConvLSTM gate fusion (forward + backward) for gfx950: three sigmoids, two spatial LayerNorms, two CELUs and the
// cell update in one pass over the 4*hidden-channel convolution output, instead of ~12 elementwise launches.
// Semantics: /root/reference/dvmvs/convlstm.py:45-59 (LayerNorm over [H,W] per (batch, channel), biased variance,
// eps 1e-5, no affine; activation celu, alpha 1; split order i,f,o,g).
//
// A LayerNorm row is one (batch, channel) plane of H*W elements (80 at 320x256, 64 in training).  A row is owned
// by a group of LPR lanes of one wavefront (16 lanes for small planes, so a wave64 normalises four channels at
// once; a full wave for larger planes); every element stays in registers between the statistics pass and the
// normalisation, so each input is read once and each output written once.
#include "dvmvs_device.h"

namespace dvmvs {

constexpr float kLnEps = 1e-5f;

template <int LPR>
__device__ inline float group_sum(float v) {
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

__device__ inline float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ inline float celu1(float x) { return x > 0.0f ? x : expm1f(x); }
__device__ inline float celu1_grad(float x) { return x > 0.0f ? 1.0f : expf(x); }

// mean / inverse std over the row held as v[0..EPL) across the LPR lanes of the group
template <int LPR, int EPL>
__device__ inline void row_stats(const float (&v)[EPL], const bool (&ok)[EPL], float inv_n, float* mean, float* rstd) {
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < EPL; ++k) s += ok[k] ? v[k] : 0.0f;
  const float mu = group_sum<LPR>(s) * inv_n;
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const float d = ok[k] ? v[k] - mu : 0.0f;
    q += d * d;
  }
  const float var = group_sum<LPR>(q) * inv_n;
  *mean = mu;
  *rstd = 1.0f / sqrtf(var + kLnEps);
}

// Everything after the convolution output of a row is in registers: LayerNorm(g), cell update, LayerNorm(c'), output gate.
template <int LPR, int EPL>
__device__ inline void gates_tail(const float (&gi)[EPL], const float (&gf)[EPL], const float (&go)[EPL], const float (&gg)[EPL],
                                  const float (&cc_)[EPL], const bool (&ok)[EPL], float inv_n, int lane, size_t st_base,
                                  float* __restrict__ h_next, float* c_next) {
  float mu, rstd;
  row_stats<LPR, EPL>(gg, ok, inv_n, &mu, &rstd);
  float z[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const float g = celu1((gg[k] - mu) * rstd);
    z[k] = sigmoidf(gf[k]) * cc_[k] + sigmoidf(gi[k]) * g;
  }
  row_stats<LPR, EPL>(z, ok, inv_n, &mu, &rstd);
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    if (!ok[k]) continue;
    const int e = lane + k * LPR;
    const float cn = (z[k] - mu) * rstd;
    c_next[st_base + e] = cn;
    h_next[st_base + e] = sigmoidf(go[k]) * celu1(cn);
  }
}

template <int LPR, int EPL>
// c_next may alias c_cur (the frame engine updates its state buffers in place): a row is read completely into registers before
// anything of it is written, and rows are owned by disjoint lane groups -- hence no __restrict__ on the two state pointers.
// n_partials > 1: the convolution output arrives as that many partial sums (K-splits of dvmvs_bottleneck_conv_fwd, partial_stride
// floats apart); they are added here in ascending order -- a fixed order, so the gates stay bit-reproducible.
__global__ __launch_bounds__(256) void lstm_gates_fwd_kernel(const float* __restrict__ cc, const float* c_cur,
                                                             float* __restrict__ h_next, float* c_next,
                                                             int B, int hidden, int HW, int n_partials, long long partial_stride) {
  constexpr int kRowsPerBlock = 256 / LPR;
  const int row = blockIdx.x * kRowsPerBlock + threadIdx.x / LPR;  // (b, channel)
  const int lane = threadIdx.x % LPR;
  const bool row_ok = row < B * hidden;
  const int b = row_ok ? row / hidden : 0;
  const int ch = row_ok ? row - b * hidden : 0;
  const size_t cc_base = (static_cast<size_t>(b) * 4 * hidden + ch) * HW;
  const size_t gate_stride = static_cast<size_t>(hidden) * HW;
  const size_t st_base = (static_cast<size_t>(b) * hidden + ch) * HW;
  const float inv_n = 1.0f / static_cast<float>(HW);

  float gi[EPL], gf[EPL], go[EPL], gg[EPL], cc_[EPL];
  bool ok[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const int e = lane + k * LPR;
    ok[k] = row_ok && e < HW;
    const int es = ok[k] ? e : 0;
    gi[k] = cc[cc_base + es];
    gf[k] = cc[cc_base + gate_stride + es];
    go[k] = cc[cc_base + 2 * gate_stride + es];
    gg[k] = cc[cc_base + 3 * gate_stride + es];
    cc_[k] = c_cur[st_base + es];
  }
#pragma unroll 4   // (four partial sums' loads in flight; the additions stay in ascending order)
  for (int s = 1; s < n_partials; ++s) {
    const float* part = cc + static_cast<size_t>(s) * partial_stride;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int es = ok[k] ? lane + k * LPR : 0;
      gi[k] += part[cc_base + es];
      gf[k] += part[cc_base + gate_stride + es];
      go[k] += part[cc_base + 2 * gate_stride + es];
      gg[k] += part[cc_base + 3 * gate_stride + es];
    }
  }
  gates_tail<LPR, EPL>(gi, gf, go, gg, cc_, ok, inv_n, lane, st_base, h_next, c_next);
}

// The cell's convolution as K-split partial sums (dvmvs_bottleneck_conv_fwd), rows of 65 .. 128 elements: a workgroup of four waves owns
// one (batch, channel) row, wave w adds up gate w's partial sums in ascending order (16 splits x 4 gates x 80 elements = 20 KB per row:
// four waves per row keep 2048 waves' loads in flight instead of 512), the sums meet in LDS and wave 0 finishes the row with the code of
// lstm_gates_fwd_kernel<64, 2> -- same element-to-lane mapping, same statistics order: bit-identical to "reduce, then gates".
__global__ __launch_bounds__(256) void lstm_gates_partials_kernel(const float* __restrict__ cc, const float* c_cur, float* __restrict__ h_next,
                                                                  float* c_next, int B, int hidden, int HW, int n_partials, long long partial_stride) {
  __shared__ float s_gate[4][128];
  const int row = blockIdx.x;      // (b, channel)
  const int gate = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = row / hidden, ch = row - b * hidden;
  const size_t cc_base = (static_cast<size_t>(b) * 4 * hidden + ch) * HW + static_cast<size_t>(gate) * hidden * HW;
  const size_t st_base = (static_cast<size_t>(b) * hidden + ch) * HW;
  float v[2];
  bool ok[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    ok[k] = lane + k * 64 < HW;
    v[k] = cc[cc_base + (ok[k] ? lane + k * 64 : 0)];
  }
#pragma unroll 4
  for (int s = 1; s < n_partials; ++s) {
    const float* part = cc + static_cast<size_t>(s) * partial_stride;
#pragma unroll
    for (int k = 0; k < 2; ++k) v[k] += part[cc_base + (ok[k] ? lane + k * 64 : 0)];
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) s_gate[gate][lane + k * 64] = v[k];
  __syncthreads();
  if (gate != 0) return;
  float gi[2], gf[2], go[2], gg[2], cc_[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = lane + k * 64;
    gi[k] = s_gate[0][e]; gf[k] = s_gate[1][e]; go[k] = s_gate[2][e]; gg[k] = s_gate[3][e];
    cc_[k] = c_cur[st_base + (ok[k] ? e : 0)];
  }
  gates_tail<64, 2>(gi, gf, go, gg, cc_, ok, 1.0f / static_cast<float>(HW), lane, st_base, h_next, c_next);
}

// Backward: gates are recomputed from (cc, c_cur); LayerNorm backward per row
//   dx = rstd * (dy - mean(dy) - xhat * mean(dy * xhat)).
template <int LPR, int EPL>
__global__ __launch_bounds__(256) void lstm_gates_bwd_kernel(const float* __restrict__ grad_h, const float* __restrict__ grad_c,
                                                             const float* __restrict__ cc, const float* __restrict__ c_cur,
                                                             float* __restrict__ grad_cc, float* __restrict__ grad_c_cur,
                                                             int B, int hidden, int HW) {
  constexpr int kRowsPerBlock = 256 / LPR;
  const int row = blockIdx.x * kRowsPerBlock + threadIdx.x / LPR;
  const int lane = threadIdx.x % LPR;
  const bool row_ok = row < B * hidden;
  const int b = row_ok ? row / hidden : 0;
  const int ch = row_ok ? row - b * hidden : 0;
  const size_t cc_base = (static_cast<size_t>(b) * 4 * hidden + ch) * HW;
  const size_t gate_stride = static_cast<size_t>(hidden) * HW;
  const size_t st_base = (static_cast<size_t>(b) * hidden + ch) * HW;
  const float inv_n = 1.0f / static_cast<float>(HW);

  float si[EPL], sf[EPL], so[EPL], gg[EPL], cc_[EPL], dh[EPL], dc[EPL];
  bool ok[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const int e = lane + k * LPR;
    ok[k] = row_ok && e < HW;
    const int es = ok[k] ? e : 0;
    si[k] = sigmoidf(cc[cc_base + es]);
    sf[k] = sigmoidf(cc[cc_base + gate_stride + es]);
    so[k] = sigmoidf(cc[cc_base + 2 * gate_stride + es]);
    gg[k] = cc[cc_base + 3 * gate_stride + es];
    cc_[k] = c_cur[st_base + es];
    dh[k] = (grad_h && ok[k]) ? grad_h[st_base + es] : 0.0f;
    dc[k] = (grad_c && ok[k]) ? grad_c[st_base + es] : 0.0f;
  }
  float mu_g, rstd_g;
  row_stats<LPR, EPL>(gg, ok, inv_n, &mu_g, &rstd_g);
  float ghat[EPL], g[EPL], z[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    ghat[k] = (gg[k] - mu_g) * rstd_g;
    g[k] = celu1(ghat[k]);
    z[k] = sf[k] * cc_[k] + si[k] * g[k];
  }
  float mu_z, rstd_z;
  row_stats<LPR, EPL>(z, ok, inv_n, &mu_z, &rstd_z);

  // through h' = o * celu(chat) and c' = chat
  float dchat[EPL], chat[EPL];
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    chat[k] = (z[k] - mu_z) * rstd_z;
    dchat[k] = ok[k] ? dc[k] + dh[k] * so[k] * celu1_grad(chat[k]) : 0.0f;
    s1 += dchat[k];
    s2 += dchat[k] * (ok[k] ? chat[k] : 0.0f);
  }
  s1 = group_sum<LPR>(s1) * inv_n;
  s2 = group_sum<LPR>(s2) * inv_n;
  float dghat[EPL];
  float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const float dz = ok[k] ? rstd_z * (dchat[k] - s1 - chat[k] * s2) : 0.0f;
    const int e = lane + k * LPR;
    if (ok[k]) {
      grad_c_cur[st_base + e] = dz * sf[k];
      grad_cc[cc_base + e] = dz * g[k] * si[k] * (1.0f - si[k]);                                  // i
      grad_cc[cc_base + gate_stride + e] = dz * cc_[k] * sf[k] * (1.0f - sf[k]);                  // f
      grad_cc[cc_base + 2 * gate_stride + e] = dh[k] * celu1(chat[k]) * so[k] * (1.0f - so[k]);   // o
    }
    dghat[k] = ok[k] ? dz * si[k] * celu1_grad(ghat[k]) : 0.0f;
    t1 += dghat[k];
    t2 += dghat[k] * (ok[k] ? ghat[k] : 0.0f);
  }
  t1 = group_sum<LPR>(t1) * inv_n;
  t2 = group_sum<LPR>(t2) * inv_n;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    if (!ok[k]) continue;
    const int e = lane + k * LPR;
    grad_cc[cc_base + 3 * gate_stride + e] = rstd_g * (dghat[k] - t1 - ghat[k] * t2);             // g
  }
}

template <bool FWD, int LPR, int EPL, typename... Args>
inline int launch_gates(hipStream_t stream, int rows, Args... args) {
  constexpr int kRowsPerBlock = 256 / LPR;
  dim3 grid((rows + kRowsPerBlock - 1) / kRowsPerBlock), block(256);
  if constexpr (FWD) hipLaunchKernelGGL((lstm_gates_fwd_kernel<LPR, EPL>), grid, block, 0, stream, args...);
  else hipLaunchKernelGGL((lstm_gates_bwd_kernel<LPR, EPL>), grid, block, 0, stream, args...);
  return launch_status();
}

// Picks the row-group width and per-lane element count for a plane of HW elements.
template <bool FWD, typename... Args>
inline int dispatch_gates(hipStream_t stream, int rows, int HW, Args... args) {
  if (HW <= 16 * 4) return launch_gates<FWD, 16, 4>(stream, rows, args...);
  // 65 .. 128 elements (80 at 320x256): one wave per row -- four times the workgroups of the 16-lane form (the cell's 512 rows: 128
  // workgroups instead of 32), and the row statistics are summed in ONE order whether the convolution arrives whole or as partial sums
  if (HW <= 64 * 2) return launch_gates<FWD, 64, 2>(stream, rows, args...);
  if (HW <= 64 * 4) return launch_gates<FWD, 64, 4>(stream, rows, args...);
  if (HW <= 64 * 8) return launch_gates<FWD, 64, 8>(stream, rows, args...);
  if (HW <= 64 * 16) return launch_gates<FWD, 64, 16>(stream, rows, args...);
  return DVMVS_EUNSUPPORTED;
}

}  // namespace dvmvs

extern "C" int dvmvs_lstm_gates_fwd(const float* combined_conv, const float* c_cur, float* h_next, float* c_next,
                                    int B, int hidden, int H, int W, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!combined_conv || !c_cur || !h_next || !c_next) return DVMVS_EINVAL;
  if (B <= 0 || hidden <= 0 || H <= 0 || W <= 0) return DVMVS_EINVAL;
  return dispatch_gates<true>(static_cast<hipStream_t>(stream), B * hidden, H * W, combined_conv, c_cur, h_next, c_next, B,
                              hidden, H * W, 1, 0LL);
}

extern "C" int dvmvs_lstm_gates_partials_fwd(const float* conv_partials, int n_partials, const float* c_cur, float* h_next, float* c_next,
                                             int B, int hidden, int H, int W, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!conv_partials || !c_cur || !h_next || !c_next) return DVMVS_EINVAL;
  if (B <= 0 || hidden <= 0 || H <= 0 || W <= 0 || n_partials <= 0) return DVMVS_EINVAL;
  // (K-split partial sums multiply the bytes of a row by n_partials: with one WAVE per LayerNorm row -- dispatch_gates, 512 waves for the
  // cell's 512 channels -- instead of 16 lanes per row (32 workgroups: 30 us over 16 splits, round 4) the separate reduction launch is gone)
  if (n_partials > 1 && H * W > 64 && H * W <= 128) {
    hipLaunchKernelGGL(lstm_gates_partials_kernel, dim3(B * hidden), dim3(256), 0, static_cast<hipStream_t>(stream), conv_partials, c_cur, h_next, c_next,
                       B, hidden, H * W, n_partials, static_cast<long long>(B) * 4 * hidden * H * W);
    return launch_status();
  }
  return dispatch_gates<true>(static_cast<hipStream_t>(stream), B * hidden, H * W, conv_partials, c_cur, h_next, c_next, B,
                              hidden, H * W, n_partials, static_cast<long long>(B) * 4 * hidden * H * W);
}

extern "C" int dvmvs_lstm_gates_bwd(const float* grad_h, const float* grad_c, const float* combined_conv,
                                    const float* c_cur, float* grad_cc, float* grad_c_cur,
                                    int B, int hidden, int H, int W, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!combined_conv || !c_cur || !grad_cc || !grad_c_cur) return DVMVS_EINVAL;
  if (B <= 0 || hidden <= 0 || H <= 0 || W <= 0) return DVMVS_EINVAL;
  return dispatch_gates<false>(static_cast<hipStream_t>(stream), B * hidden, H * W, grad_h, grad_c, combined_conv, c_cur,
                               grad_cc, grad_c_cur, B, hidden, H * W);
}
