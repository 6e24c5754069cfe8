// 3x3 convolutions on the bottleneck maps of a frame (8x10 and 16x20 pixels at 320x256) as a weight-streaming fp32 MFMA GEMM, gfx950.
//
// Which layers: the ConvLSTM convolution (1024 -> 2048 channels on the 8x10 map, /root/reference/dvmvs/convlstm.py:43-44,
// fusionnet/model.py:308-337) and the 256 / 512-channel 3x3 layers of the encoder's last block and the decoder's first block
// (fusionnet/model.py:167-305) -- 80 or 320 output pixels against 2 304 ... 9 216-term reductions, i.e. GEMMs with a tiny M whose
// cost is streaming the weights (75.5 MB for the ConvLSTM: 9.4 us at 8 TB/s) and 3.0 GFLOP of exact-fp32 arithmetic (19 us at the
// 157 TF fp32 MFMA rate).  Why not MIOpen here (round 4: it was the path of every other convolution then): for exactly these problems MIOpen
// picks its `igemm_fwd_gtcx35_nhwc_..._gkgs` kernels -- K split over workgroups and accumulated with float ATOMICS -- so the result
// differs from run to run (measured: tools/conv_determinism_probe.py), which through the discrete depth estimate makes whole depth
// maps process-dependent; and they cost 50 us + two layout transposes for the ConvLSTM (rocBLAS on an im2col
// operand, split-K as a batched GEMM: 30-32 us, tools/lstm_conv_probe.py).
//
// Formulation.  out[n, p] = sum_{c, tap} W[n, c, tap] * x[c, p @ tap].  A wave owns 16 output channels (MFMA rows) x 80 pixels (five
// 16-column tiles of v_mfma_f32_16x16x4_f32, 20 accumulator registers) x one split of the input channels; a workgroup is four waves
// = 64 output channels that share the split's slice of x, staged once into LDS with its zero ring (padding = 1), so the B operand
// of every MFMA is one ds_read_b32 at (channel plane + pixel offset + tap offset) and no im2col operand ever exists.  The weights
// are re-packed once (they are constants at inference) into the A-operand order -- [16-channel tile][16 input channels][tap][lane]
// float4 -- so a wave streams them with fully coalesced 1 KB loads, 9 per 16 input channels, double-buffered against 180 MFMAs.
// Splits are summed by the consumer in a fixed order (dvmvs_lstm_gates_partials_fwd, dvmvs_partial_sums_bias_act_fwd): the result
// is bit-reproducible.  An fp32 MFMA is an fmaf chain over k (cdna guide, "exact f32"): no reduced precision anywhere.
#include <stdlib.h>

#include "dvmvs_device.h"

namespace dvmvs {

typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int kBcRows = 16;     // output channels per wave
constexpr int kBcPixels = 80;   // pixels per wave
constexpr int kBcPT = kBcPixels / 16;
constexpr int kBcWaves = 4;
constexpr int kBcTargetWaves = 2048;   // two per SIMD
#ifndef DVMVS_BC_STAGE
#define DVMVS_BC_STAGE 8
#endif
constexpr int kBcStage = DVMVS_BC_STAGE;   // elements of the staged slice a thread has in flight at a time

// ---- optional timeline instrumentation (tools/bottleneck_conv_trace.py; built only by `make trace`): per wave, on the 100 MHz wall clock:
// [0] start, [1] slice staged (after the barrier), [2] sum over groups of "waiting for the group's weights", [3] of "MFMAs of the group",
// [4] group loop left, [5] end (partials stored), [6] SIMD-unique id (CU, SIMD), [7] groups ----
#ifdef DVMVS_SWEEP_TRACE
constexpr int kBcTraceWords = 8, kBcTraceWaves = 8192;
__device__ unsigned long long g_bc_trace[kBcTraceWaves * kBcTraceWords];
#define BC_TRACE(...) __VA_ARGS__
#define BC_NOW() __builtin_amdgcn_s_memrealtime()
#else
#define BC_TRACE(...)
#endif

struct BottleneckConvArgs {
  const float* x;        // [B, C_in, H_in, W_in]
  const float* packed;   // [n_tiles][C_in / 16][9][64] float4
  float* partials;       // [splits][B][C_out][P]
  int B, C_in, C_out, n_tiles, cs, splits;
};

// split count: the smallest divisor of C_in / 16 that gives at least kBcTargetWaves waves (or all of them) AND whose slice of x
// (C_in / splits channels x `plane` padded pixels) fits the 64 KB of LDS a workgroup stages it in
constexpr size_t kBcLdsBytes = 64 * 1024;
__host__ __device__ inline int bottleneck_splits(int B, int C_out, int C_in, int P, int plane) {
  const int n_tiles = (C_out + kBcRows - 1) / kBcRows, groups = C_in / 16, pixel_groups = P / kBcPixels;
  const int per_split = n_tiles * pixel_groups * B;
  int target = kBcTargetWaves;
#if defined(DVMVS_SWEEP_TUNING) && !defined(__HIP_DEVICE_COMPILE__)      // tools-only build: DVMVS_BC_WAVES overrides the wave target
  if (const char* w = getenv("DVMVS_BC_WAVES")) target = atoi(w);
#endif
  const int wanted = (target + per_split - 1) / per_split;
  for (int d = 1; d <= groups; ++d)
    if (groups % d == 0 && d >= wanted && sizeof(float) * static_cast<size_t>(C_in / d) * plane <= kBcLdsBytes) return d;
  return groups;
}

// CH: groups of 16 input channels whose weights are requested together; the split's group count is a multiple of it
// One element of the 2x bilinear up-sampling (align_corners = True) of a [H / 2, W / 2] plane at (oy, ox) of the [H, W] map: dvmvs_upsample2x_fwd's
// expression (csrc/frame_ops.hip: ATen's op order, no FMA contraction), so a map up-sampled on the fly here holds the bits that kernel writes.
template <int H, int W>
__device__ inline void up2x_taps(int oy, int ox, int* o00, int* o01, int* o10, int* o11, float* h0, float* h1, float* w0, float* w1) {
#pragma clang fp contract(off)
  constexpr int HS = H / 2, WS = W / 2;
  const float sh = static_cast<float>(HS - 1) / static_cast<float>(H - 1), sw = static_cast<float>(WS - 1) / static_cast<float>(W - 1);
  const float fy = sh * static_cast<float>(oy), fx = sw * static_cast<float>(ox);
  const int y0 = static_cast<int>(fy), x0 = static_cast<int>(fx);
  const int y1 = y0 + (y0 < HS - 1 ? 1 : 0), x1 = x0 + (x0 < WS - 1 ? 1 : 0);
  *h1 = fy - static_cast<float>(y0);
  *h0 = 1.0f - *h1;
  *w1 = fx - static_cast<float>(x0);
  *w0 = 1.0f - *w1;
  *o00 = y0 * WS + x0; *o01 = y0 * WS + x1; *o10 = y1 * WS + x0; *o11 = y1 * WS + x1;
}

__device__ inline float up2x_blend(float v00, float v01, float v10, float v11, float h0, float h1, float w0, float w1) {
#pragma clang fp contract(off)
  return h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11);
}

// UP2X (round 6): ``x`` is the [B, C_in, H_IN / 2, W_IN / 2] map whose 2x bilinear up-sampling the layer convolves (the decoder's first
// up-convolution: the ConvLSTM state, 512 x 8 x 10): every staged element is interpolated from its four taps on the way into LDS instead of
// being read from a map a separate up-sampling launch wrote.
template <int H_IN, int W_IN, int STRIDE, int CH, bool WINDOW = false, bool UP2X = false>
__global__ __launch_bounds__(kBcWaves * 64, 2) void bottleneck_conv_kernel(BottleneckConvArgs a) {
  // WINDOW (round 6): a workgroup stages only the padded input rows its 80 output pixels read -- (rows - 1) * STRIDE + 3 of them -- instead of
  // the whole padded map: the stride-2 layer on the 32 x 40 map (9 of 34 rows: 24 KB of LDS per 16 channels instead of 91 KB) and the stride-1
  // layers of the 16 x 20 map (6 of 18 rows).  Same MFMAs on the same operands in the same order: bit-identical partial sums; the split count
  // of the 16 x 20 layers is still chosen with the whole map's LDS footprint (bottleneck_plane), so their sums are those of rounds 4-5.
  constexpr int PW = W_IN + 2;
  constexpr int W_OUT = W_IN / STRIDE, H_OUT = H_IN / STRIDE, P = H_OUT * W_OUT, PG = P / kBcPixels;
  constexpr int ROWS_OUT = kBcPixels / W_OUT;                                                  // output rows of a pixel group
  constexpr int STAGED_ROWS = WINDOW ? (ROWS_OUT - 1) * STRIDE + 3 : H_IN + 2;
  constexpr int PLANE = STAGED_ROWS * PW;
  static_assert(P % kBcPixels == 0, "the map must split into 80-pixel groups");
  static_assert(!WINDOW || (kBcPixels % W_OUT == 0 && STAGED_ROWS <= H_IN + 2), "a pixel group must be whole output rows");
  extern __shared__ __attribute__((aligned(16))) float xs[];   // [cs][PLANE]: the split's channels with their zero ring

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_tile = blockIdx.x * kBcWaves + wave;
  const int split = blockIdx.y;
  BC_TRACE(const unsigned long long tr_start = BC_NOW(); unsigned long long tr_staged = 0, tr_wait = 0, tr_mfma = 0, tr_loop = 0;)
  const int b = blockIdx.z / PG, pg = blockIdx.z - b * PG;
  const int c0 = split * a.cs;

  // ---- stage x[b, c0 : c0 + cs] into LDS: eight elements per thread in flight at a time, the zero ring through out-of-range
  // offsets of a raw buffer descriptor (a branch per element and one load at a time took a third of the kernel) ----
  constexpr int SRC_PLANE = UP2X ? (H_IN / 2) * (W_IN / 2) : H_IN * W_IN;
  gcfloat_p xg = as_global(a.x) + (static_cast<size_t>(b) * a.C_in + c0) * SRC_PLANE;
  const __amdgpu_buffer_rsrc_t x_resource =
      __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, static_cast<int>(sizeof(float) * static_cast<unsigned int>(a.cs) * SRC_PLANE), 0x00020000);
  const int staged = a.cs * PLANE;
  const int row0 = WINDOW ? pg * ROWS_OUT * STRIDE : 0;      // first staged row of the padded map
  if constexpr (UP2X) {
    for (int i = tid; i < staged; i += kBcWaves * 64) {
      const int c = i / PLANE, r = i - c * PLANE;
      const int yw = r / PW, xx = r - yw * PW;
      const int yy = yw + row0;
      float v = 0.0f;
      if (yy >= 1 && yy <= H_IN && xx >= 1 && xx <= W_IN) {
        int o00, o01, o10, o11;
        float h0, h1, w0, w1;
        up2x_taps<H_IN, W_IN>(yy - 1, xx - 1, &o00, &o01, &o10, &o11, &h0, &h1, &w0, &w1);
        gcfloat_p plane = xg + c * SRC_PLANE;
        v = up2x_blend(plane[o00], plane[o01], plane[o10], plane[o11], h0, h1, w0, w1);
      }
      xs[i] = v;
    }
  } else
  for (int i0 = tid; i0 < staged; i0 += kBcStage * kBcWaves * 64) {
    float v[kBcStage];
#pragma unroll
    for (int k = 0; k < kBcStage; ++k) {
      const int i = i0 + k * kBcWaves * 64;
      const int c = i / PLANE, r = i - c * PLANE;
      const int yw = r / PW, xx = r - yw * PW;
      const int yy = yw + row0;      // (row of the padded map)
      const bool in = i < staged && yy >= 1 && yy <= H_IN && xx >= 1 && xx <= W_IN;
      const unsigned int offset = in ? static_cast<unsigned int>(sizeof(float)) * static_cast<unsigned int>(c * (H_IN * W_IN) + (yy - 1) * W_IN + (xx - 1)) : 0x80000000u;
      v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_resource, static_cast<int>(offset), 0, 0));
    }
#pragma unroll
    for (int k = 0; k < kBcStage; ++k) {
      const int i = i0 + k * kBcWaves * 64;
      if (i < staged) xs[i] = v[k];
    }
  }
  __syncthreads();
  BC_TRACE(tr_staged = BC_NOW();)
  if (n_tile >= a.n_tiles) return;

  // this lane's B-operand position: input channel (lane >> 4) of a group of four, pixel (lane & 15) of each 16-pixel tile
  int pix[kBcPT];
#pragma unroll
  for (int pt = 0; pt < kBcPT; ++pt) {
    const int p = pg * kBcPixels + pt * 16 + (lane & 15);
    const int py = p / W_OUT, px = p - py * W_OUT;
    pix[pt] = (py * STRIDE - row0) * PW + px * STRIDE + (lane >> 4) * PLANE;
  }
  float4v acc[kBcPT];
#pragma unroll
  for (int pt = 0; pt < kBcPT; ++pt) acc[pt] = float4v{0.0f, 0.0f, 0.0f, 0.0f};

  const int groups = a.cs / 16;
  const float4v DVMVS_GLOBAL* wp = reinterpret_cast<const float4v DVMVS_GLOBAL*>(as_global(a.packed)) +
                                   (static_cast<size_t>(n_tile) * (a.C_in / 16) + c0 / 16) * (9 * 64) + lane;
  // Weights of CH groups of 16 input channels (9 float4 each) are requested in one straight-line burst and consumed group by group:
  // the wave waits for the first nine loads only (the compiler counts the newer ones: s_waitcnt vmcnt(9 * (CH - 1))) while the rest
  // arrive behind 180 MFMAs per group.  The product launches CH = 1 (see launch_bottleneck_conv for the measurement).
  if constexpr (CH == 0) {
    // Rolling double buffer (round 6): the nine requests of group g + 1 are issued BEFORE the 180 MFMAs of group g and waited for behind them --
    // two register sets that swap roles, the loop unrolled by two, the request behind the last group issued out of range (no traffic, no
    // condition around a request: the compiler's s_waitcnt placement then waits for exactly the older nine).  Round 4's bursts (CH = 1 / 2 / 4
    // groups requested together, consumed together) left a wave waiting for its weights at the head of every group with only the other wave of
    // its SIMD to cover it: 34.1 us for the ConvLSTM layer = 56 % of the MFMA rate.  Same MFMAs in the same order: bit-identical partial sums.
    const unsigned int packed_bytes = static_cast<unsigned int>(sizeof(float4v)) * static_cast<unsigned int>(a.n_tiles) * static_cast<unsigned int>(a.C_in / 16) * (9u * 64u);
    const __amdgpu_buffer_rsrc_t w_resource = __builtin_amdgcn_make_buffer_rsrc((void*)as_global(a.packed), 0, static_cast<int>(packed_bytes), 0x00020000);
    const unsigned int w_base = static_cast<unsigned int>(sizeof(float4v)) * ((static_cast<unsigned int>(n_tile) * static_cast<unsigned int>(a.C_in / 16) +
                                                                               static_cast<unsigned int>(c0 / 16)) * (9u * 64u) + static_cast<unsigned int>(lane));
    auto request_taps = [&](float4v (&w)[9], int g) {
      const unsigned int vo = g < groups ? w_base + static_cast<unsigned int>(g) * (9u * 64u * 16u) : 0x80000000u;
#pragma unroll
      for (int t = 0; t < 9; ++t)
        w[t] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(w_resource, static_cast<int>(vo + (g < groups ? static_cast<unsigned int>(t) * (64u * 16u) : 0u)), 0, 0));
    };
    auto consume = [&](const float4v (&w)[9], int g) {
      const float* xc = xs + g * 16 * PLANE;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int toff = (t / 3) * PW + (t % 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int pt = 0; pt < kBcPT; ++pt)
            acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][j], xc[j * 4 * PLANE + pix[pt] + toff], acc[pt], 0, 0, 0);
        }
      }
    };
    float4v wa[9], wb[9];
    request_taps(wa, 0);
    for (int g = 0; g < groups; g += 2) {
      request_taps(wb, g + 1);
      __builtin_amdgcn_sched_barrier(0);
      consume(wa, g);
      __builtin_amdgcn_sched_barrier(0);
      request_taps(wa, g + 2);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 1 < groups) consume(wb, g + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else
  for (int g0 = 0; g0 < groups; g0 += (CH > 0 ? CH : 1)) {
    float4v w[CH > 0 ? CH : 1][9];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) w[c][t] = wp[(g0 + c) * (9 * 64) + t * 64];
    __builtin_amdgcn_sched_barrier(0);   // the requests stay in front of the MFMAs (the scheduler otherwise sinks each next to its use)
    BC_TRACE(const unsigned long long tr_a = BC_NOW(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long tr_b = BC_NOW(); tr_wait += tr_b - tr_a;)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float* xc = xs + (g0 + c) * 16 * PLANE;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int toff = (t / 3) * PW + (t % 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int pt = 0; pt < kBcPT; ++pt)
            acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c][t][j], xc[j * 4 * PLANE + pix[pt] + toff], acc[pt], 0, 0, 0);
        }
      }
    }
    BC_TRACE(__builtin_amdgcn_sched_barrier(0); tr_mfma += BC_NOW() - tr_b;)
  }
  BC_TRACE(tr_loop = BC_NOW();)

  // D[row = (lane >> 4) * 4 + r][col = lane & 15] -> partials[split][b][n][p]
  gfloat_p out = as_global(a.partials) + ((static_cast<size_t>(split) * a.B + b) * a.C_out) * P;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n_tile * kBcRows + (lane >> 4) * 4 + r;
    if (n < a.C_out) {
#pragma unroll
      for (int pt = 0; pt < kBcPT; ++pt) out[static_cast<size_t>(n) * P + pg * kBcPixels + pt * 16 + (lane & 15)] = acc[pt][r];
    }
  }
  BC_TRACE({
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long tr_end = BC_NOW();
    const size_t wv = (static_cast<size_t>(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * kBcWaves + wave;
    if (lane == 0 && wv < kBcTraceWaves) {
      unsigned int hw_id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
      unsigned int xcc_id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
      unsigned long long* t = g_bc_trace + wv * kBcTraceWords;
      t[0] = tr_start; t[1] = tr_staged; t[2] = tr_wait; t[3] = tr_mfma; t[4] = tr_loop; t[5] = tr_end;
      t[6] = (static_cast<unsigned long long>(xcc_id & 0xf) << 32) | hw_id; t[7] = static_cast<unsigned long long>(groups);
    }
  })
}

// packed[((tile * G + g) * 9 + tap) * 64 + lane][j] = W[16 tile + (lane & 15)][16 g + 4 j + (lane >> 4)][tap]
__global__ __launch_bounds__(256) void bottleneck_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int C_out, int C_in, long long total) {
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= total) return;
  const int j = static_cast<int>(i & 3), lane = static_cast<int>((i >> 2) & 63);
  long long rest = i >> 8;
  const int tap = static_cast<int>(rest % 9);
  rest /= 9;
  const int G = C_in / 16;
  const int g = static_cast<int>(rest % G), tile = static_cast<int>(rest / G);
  const int n = tile * 16 + (lane & 15), c = g * 16 + j * 4 + (lane >> 4);
  packed[i] = n < C_out ? w[(static_cast<size_t>(n) * C_in + c) * 9 + tap] : 0.0f;
}

// dst[b, n, :] = act(sum_s partials[s, b, n, :] + bias[n]): the splits in ascending order, then the bias -- one fixed order.
template <int ACT>
__global__ __launch_bounds__(256) void partial_sums_bias_act_kernel(const float* __restrict__ partials, int n_partials, float* __restrict__ dst,
                                                                    long long dst_batch_stride, const float* __restrict__ bias, int B, int C, int HW) {
  const long long per_split = static_cast<long long>(B) * C * HW;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < per_split; i += static_cast<long long>(gridDim.x) * 256) {
    // eight splits' loads in flight at a time (a plain `v += partials[s]` loop waits for every load before it issues the next:
    // 11 us for 32 splits of a 512 x 80 map); the additions keep the ascending order
    float v = 0.0f;
    for (int s0 = 0; s0 < n_partials; s0 += 8) {
      float part[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) part[k] = s0 + k < n_partials ? partials[(s0 + k) * per_split + i] : 0.0f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (s0 + k < n_partials) v = (s0 + k == 0) ? part[k] : v + part[k];
    }
    const int plane = static_cast<int>(i / HW), b = plane / C, c = plane - b * C;
    v += bias ? bias[c] : 0.0f;
    if (ACT == 1) v = fmaxf(v, 0.0f);
    dst[static_cast<size_t>(b) * dst_batch_stride + static_cast<size_t>(c) * HW + (i - static_cast<long long>(plane) * HW)] = v;
  }
}

inline constexpr int bottleneck_staged_plane(int H_in, int W_in, int stride, bool window) {
  return (window ? (kBcPixels / (W_in / stride) - 1) * stride + 3 : H_in + 2) * (W_in + 2);
}

template <int H_IN, int W_IN, int STRIDE, bool WINDOW = false, bool UP2X = false>
int launch_bottleneck_conv(const BottleneckConvArgs& a, hipStream_t stream) {
  constexpr int P = (H_IN / STRIDE) * (W_IN / STRIDE), PLANE = bottleneck_staged_plane(H_IN, W_IN, STRIDE, WINDOW);
  size_t lds = sizeof(float) * static_cast<size_t>(a.cs) * PLANE;
  if (lds > kBcLdsBytes) return DVMVS_EUNSUPPORTED;
  const dim3 grid((a.n_tiles + kBcWaves - 1) / kBcWaves, a.splits, a.B * (P / kBcPixels)), block(kBcWaves * 64);
#ifdef DVMVS_SWEEP_TUNING
  if (const char* kb = getenv("DVMVS_BC_LDS_KB")) {      // tools-only build: a larger LDS request caps the workgroups per CU (160 KB / request)
    const size_t want = static_cast<size_t>(atoi(kb)) * 1024;
    if (want > lds) {
      lds = want;
      if (want > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bottleneck_conv_kernel<H_IN, W_IN, STRIDE, 1, WINDOW, UP2X>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(want));
      }
    }
  }
  const int groups = a.cs / 16;      // tools-only build: DVMVS_BC_CH=1|2|4 forces the request burst (tools/lstm_conv_probe.py)
  if (const char* ch = getenv("DVMVS_BC_CH")) {
    const int c = atoi(ch);
    if (c == 0 || c == 1 || groups % c == 0) {
      if (c == 0) hipLaunchKernelGGL((bottleneck_conv_kernel<H_IN, W_IN, STRIDE, 0, WINDOW, UP2X>), grid, block, lds, stream, a);
      else if (c == 1) hipLaunchKernelGGL((bottleneck_conv_kernel<H_IN, W_IN, STRIDE, 1, WINDOW, UP2X>), grid, block, lds, stream, a);
      else if (c == 2) hipLaunchKernelGGL((bottleneck_conv_kernel<H_IN, W_IN, STRIDE, 2, WINDOW, UP2X>), grid, block, lds, stream, a);
      else hipLaunchKernelGGL((bottleneck_conv_kernel<H_IN, W_IN, STRIDE, 4, WINDOW, UP2X>), grid, block, lds, stream, a);
      return launch_status();
    }
  }
#endif
  // CH = 1: one group's nine requests per burst.  Measured on the ConvLSTM layer (tools/bc_tuning_probe.sh, MI355X): CH 1 / 2 / 4 =
  // 39.8 / 40.8 / 43.7 us at 16 splits, 50-53 us at 8 splits, 43-44 us at 32 -- the kernel is not waiting for its weights (two waves
  // per SIMD cover each other's requests); it runs at ~48 % of the fp32 MFMA rate whatever the burst (PMC: MFMA pipe busy 50 % of the
  // kernel, LDS pipe 25 %, profiles/r04_bottleneck_conv_pmc.txt).
  hipLaunchKernelGGL((bottleneck_conv_kernel<H_IN, W_IN, STRIDE, 1, WINDOW, UP2X>), grid, block, lds, stream, a);
  return launch_status();
}

inline bool bottleneck_shape_ok(int C_out, int C_in, int H_in, int W_in, int stride) {
  if (C_out <= 0 || C_in <= 0 || C_in % 16 != 0) return false;
  // the 1/32 and 1/16 maps of a 320x256 frame (80-pixel groups), and (round 6) the stride-2 layer that takes the 1/8 map down to the 1/16 one
  // (encoder_block2's down-convolution: MIOpen ran it as im2col + GEMM + an epilogue launch); other sizes are not this kernel's
  return (H_in == 8 && W_in == 10 && stride == 1) || (H_in == 16 && W_in == 20 && (stride == 1 || stride == 2)) || (H_in == 32 && W_in == 40 && stride == 2);
}

// padded pixels per channel the split count of a shape is chosen with (what a workgroup stages, except for the 16 x 20 stride-1 layers: see WINDOW)
inline int bottleneck_plane(int H_in, int W_in, int stride) { return bottleneck_staged_plane(H_in, W_in, stride, H_in == 32); }

}  // namespace dvmvs

extern "C" size_t dvmvs_bottleneck_conv_packed_bytes(int C_out, int C_in) {
  if (C_out <= 0 || C_in <= 0 || C_in % 16 != 0) return 0;
  return sizeof(float) * static_cast<size_t>((C_out + 15) / 16) * 16 * C_in * 9;
}

extern "C" int dvmvs_bottleneck_conv_pack(const float* weight, float* packed, int C_out, int C_in, dvmvs_stream_t stream) {
  if (!weight || !packed) return DVMVS_EINVAL;
  const size_t bytes = dvmvs_bottleneck_conv_packed_bytes(C_out, C_in);
  if (bytes == 0) return DVMVS_EUNSUPPORTED;
  const long long total = static_cast<long long>(bytes / sizeof(float));
  hipLaunchKernelGGL(dvmvs::bottleneck_pack_kernel, dim3(static_cast<unsigned int>((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     weight, packed, C_out, C_in, total);
  return dvmvs::launch_status();
}

extern "C" int dvmvs_bottleneck_conv_splits(int B, int C_out, int C_in, int H_in, int W_in, int stride) {
  if (B <= 0 || !dvmvs::bottleneck_shape_ok(C_out, C_in, H_in, W_in, stride)) return DVMVS_EUNSUPPORTED;
  return dvmvs::bottleneck_splits(B, C_out, C_in, (H_in / stride) * (W_in / stride), dvmvs::bottleneck_plane(H_in, W_in, stride));
}

extern "C" int dvmvs_bottleneck_conv_fwd(const float* x, const float* packed, float* partials, int B, int C_in, int H_in, int W_in, int C_out,
                                         int stride, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!x || !packed || !partials || B <= 0) return DVMVS_EINVAL;
  if (!bottleneck_shape_ok(C_out, C_in, H_in, W_in, stride)) return DVMVS_EUNSUPPORTED;
  BottleneckConvArgs a;
  a.x = x; a.packed = packed; a.partials = partials;
  a.B = B; a.C_in = C_in; a.C_out = C_out;
  a.n_tiles = (C_out + kBcRows - 1) / kBcRows;
  a.splits = bottleneck_splits(B, C_out, C_in, (H_in / stride) * (W_in / stride), bottleneck_plane(H_in, W_in, stride));
  a.cs = C_in / a.splits;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (H_in == 32) return launch_bottleneck_conv<32, 40, 2, true>(a, s);
  if (H_in == 8 && W_in == 10) return launch_bottleneck_conv<8, 10, 1>(a, s);
  if (stride == 1) {
#ifdef DVMVS_SWEEP_TUNING
    if (getenv("DVMVS_BC_WHOLE_MAP")) return launch_bottleneck_conv<16, 20, 1>(a, s);      // tools-only build: rounds 4-5's whole-map staging
#endif
    return launch_bottleneck_conv<16, 20, 1, true>(a, s);
  }
  return launch_bottleneck_conv<16, 20, 2>(a, s);
}

extern "C" int dvmvs_bottleneck_conv_up2x_fwd(const float* x, const float* packed, float* partials, int B, int C_in, int H_in, int W_in, int C_out,
                                              dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!x || !packed || !partials || B <= 0) return DVMVS_EINVAL;
  if (H_in != 16 || W_in != 20 || !bottleneck_shape_ok(C_out, C_in, H_in, W_in, 1)) return DVMVS_EUNSUPPORTED;
  BottleneckConvArgs a;
  a.x = x; a.packed = packed; a.partials = partials;
  a.B = B; a.C_in = C_in; a.C_out = C_out;
  a.n_tiles = (C_out + kBcRows - 1) / kBcRows;
  a.splits = bottleneck_splits(B, C_out, C_in, H_in * W_in, bottleneck_plane(H_in, W_in, 1));      // (the split count of dvmvs_bottleneck_conv_fwd: the same sums)
  a.cs = C_in / a.splits;
  return launch_bottleneck_conv<16, 20, 1, true, true>(a, static_cast<hipStream_t>(stream));
}

#ifdef DVMVS_SWEEP_TRACE
extern "C" int dvmvs_debug_bottleneck_conv_trace(unsigned long long* host, int waves) {
  if (waves > dvmvs::kBcTraceWaves) waves = dvmvs::kBcTraceWaves;
  return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(dvmvs::g_bc_trace), sizeof(unsigned long long) * dvmvs::kBcTraceWords * waves));
}
#endif

extern "C" int dvmvs_partial_sums_bias_act_fwd(const float* partials, int n_partials, float* dst, long long dst_batch_stride, const float* bias,
                                               int B, int C, int HW, int activation, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!partials || !dst || n_partials <= 0 || B <= 0 || C <= 0 || HW <= 0) return DVMVS_EINVAL;
  if (activation != 0 && activation != 1) return DVMVS_EUNSUPPORTED;
  const long long total = static_cast<long long>(B) * C * HW;
  const unsigned int grid = static_cast<unsigned int>((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (activation == 1) hipLaunchKernelGGL((partial_sums_bias_act_kernel<1>), dim3(grid), dim3(256), 0, s, partials, n_partials, dst, dst_batch_stride, bias, B, C, HW);
  else hipLaunchKernelGGL((partial_sums_bias_act_kernel<0>), dim3(grid), dim3(256), 0, s, partials, n_partials, dst, dst_batch_stride, bias, B, C, HW);
  return launch_status();
}
