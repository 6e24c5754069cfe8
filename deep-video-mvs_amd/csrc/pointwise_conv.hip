// 1x1 convolutions of a frame at small batch: fp32-MFMA GEMM straight from the NCHW maps with bias, ReLU and the residual add in its
// store path, gfx950.
//
// Which layers (fusionnet/model.py:20-77: the MnasNet feature extractor's expansion / projection layers and the feature-pyramid's lateral
// layers; pairnet alike): 36 launches of a frame.  They were MIOpen's 1x1 path -- a rocBLAS GEMM of 4.7 - 10.8 us per layer -- followed,
// for the 22 of them that are not directly consumed by a depthwise layer, by one dvmvs_bias_act_fwd launch (bias, activation, residual:
// 3.9 - 4.9 us each): 234 + 97 us of kernel time of a 1 207 us frame (profiles/r06_first_half_bench_timed_region_lookahead1.csv).
//
// The problems are tiny -- out[co, p] = sum_ci W[co, ci] x[ci, p] with 16 ... 1 152 channels on either side and 80 ... 20 480 pixels: 31 MFLOP at
// most, a fraction of a microsecond of the matrix cores -- so a launch costs what its longest dependent chain of memory round trips costs.
// Formulation: one wave owns ONE 16-pixel x 16-channel output tile and a slice of the input channels; the waves of a workgroup (1 ... 16
// input-channel splits of the same tile) add their partial tiles through LDS in a fixed order (deterministic, no atomics), and the first
// wave applies the epilogue and stores.  Per 16 input channels (a "quad": four v_mfma_f32_16x16x4_f32) a wave needs four dwords of the map
// -- A operand: lane l holds x[ci = 4 g + l / 16][p0 + l % 16], rows of 64 contiguous bytes straight from the NCHW map, no im2col, no LDS --
// and ONE float4 of the weights, which are constants at inference and packed once into B-operand order ([16-channel output tile][quad][lane]
// float4: W[co = 16 t + l % 16][ci = 16 q + 4 j + l / 16], j = 0 ... 3; zero beyond C_out / C_in).  All requests of a round of four quads
// are issued before the first MFMA and the next round's behind them, so a wave's chain is one memory round trip per eight quads; the split
// count is chosen so that most layers need a single round (launch -> one round trip -> 4 ... 16 MFMAs -> LDS -> store).
// Out-of-range rows cost no branch: the map is read through a raw buffer descriptor of exactly C_in * H * W floats (a quad beyond C_in reads
// zeros), and a pixel tile that hangs over the end of a plane reads the next plane's first pixels into rows that are never stored (a row
// of D depends on the same row of A only).
#include "dvmvs_device.h"

namespace dvmvs {

typedef float float4v __attribute__((ext_vector_type(4)));

struct PointwiseConvArgs {
  const float* x;          // [B, C_in, H*W]; batch item b at x + b * x_batch_stride, its planes dense
  const float* packed;     // pointwise_conv_pack_kernel's layout
  const float* bias;       // [C_out] or null
  const float* residual;   // null, [B, C_out, H, W] (mode 1) or [B, C_out, H/2, W/2] (mode 2); batch item b at residual + b * residual_batch_stride
  float* dst;              // [B, C_out, H*W]; batch item b at dst + b * dst_batch_stride, its planes dense
  long long x_batch_stride, dst_batch_stride, residual_batch_stride;
  int C_in, C_out, HW, W;
  int n_quads;             // ceil(C_in / 16)
  int quads_per_split;     // input-channel quads per wave of a workgroup
};

constexpr int kPwRound = 4;                      // quads whose requests are issued together
constexpr unsigned int kPwOutOfRange = 0x7fffffffu;

__global__ __launch_bounds__(256) void pointwise_conv_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int C_out, int C_in, int n_quads,
                                                                  long long total) {
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= total) return;
  const int j = static_cast<int>(i & 3), lane = static_cast<int>((i >> 2) & 63);
  const long long tq = i >> 8;
  const int q = static_cast<int>(tq % n_quads), t = static_cast<int>(tq / n_quads);
  const int co = 16 * t + (lane & 15), ci = 16 * q + 4 * j + (lane >> 4);
  packed[i] = (co < C_out && ci < C_in) ? w[static_cast<size_t>(co) * C_in + ci] : 0.0f;
}

struct PwOperands {
  float4v w;
  float x[4];
};

// ACT: 0 none, 1 ReLU.  RES: 0 none; 1 a residual of the output's shape is added after the activation; 2 the residual has half the
// resolution and is nearest-up-sampled on the fly (dvmvs_bias_act_fwd's modes: the epilogue this kernel's store path replaces).
template <int ACT, int RES>
__global__ __launch_bounds__(1024) void pointwise_conv_kernel(PointwiseConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float4v s_partial[];      // [split][lane]
  const int lane = threadIdx.x & 63, split = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), splits = blockDim.x >> 6;
  const int p0 = blockIdx.x * 16, tile = blockIdx.y, b = blockIdx.z;
  gcfloat_p xg = as_global(a.x) + static_cast<size_t>(b) * a.x_batch_stride;
  const __amdgpu_buffer_rsrc_t x_resource =
      __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, static_cast<int>(sizeof(float) * static_cast<unsigned int>(a.C_in) * a.HW), 0x00020000);
  const unsigned int tile_bytes = static_cast<unsigned int>(a.n_quads) * 1024u;
  const __amdgpu_buffer_rsrc_t w_resource = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(as_global(a.packed) + static_cast<size_t>(tile) * (tile_bytes / 4)), 0, static_cast<int>(tile_bytes), 0x00020000);

  const int q0 = split * a.quads_per_split, q1 = min(q0 + a.quads_per_split, a.n_quads);
  const unsigned int group_bytes = 16u * static_cast<unsigned int>(a.HW);      // four planes
  const unsigned int x_lane = 4u * (static_cast<unsigned int>(lane >> 4) * a.HW + p0 + (lane & 15));
  const unsigned int w_lane = 16u * lane;

  // the requests of one round: quad q + i for i < kPwRound (beyond q1: out of range, zeros, no memory access)
  auto request = [&](PwOperands (&r)[kPwRound], int q) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < kPwRound; ++i) {
      const bool live = q + i < q1;
      r[i].w = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(w_resource, static_cast<int>(live ? w_lane + 1024u * (q + i) : kPwOutOfRange), 0, 0));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        r[i].x[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                  x_resource, static_cast<int>(live ? x_lane + group_bytes * (4 * (q + i) + j) : kPwOutOfRange), 0, 0));
    }
  };

  float4v acc[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};      // two chains: a dependent MFMA waits ~10 cycles less than its 8 passes
  PwOperands now[kPwRound], next[kPwRound];
  request(now, q0);
  for (int q = q0; q < q1; q += kPwRound) {
    const bool more = q + kPwRound < q1;
    if (more) request(next, q + kPwRound);
#pragma unroll
    for (int i = 0; i < kPwRound; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(now[i].x[j], now[i].w[j], acc[j & 1], 0, 0, 0);
    if (more) {
#pragma unroll
      for (int i = 0; i < kPwRound; ++i) now[i] = next[i];
    }
  }
  float4v v = acc[0] + acc[1];

  // ---- the splits, added in a fixed order by the first wave ----
  if (splits > 1) {
    s_partial[split * 64 + lane] = v;
    __syncthreads();
    if (split != 0) return;
    v = s_partial[lane];
#pragma unroll 4
    for (int s = 1; s < splits; ++s) v += s_partial[s * 64 + lane];
  }

  // ---- epilogue: lane l holds pixels p0 + 4 (l / 16) + (0 ... 3) of output channel 16 tile + l % 16 ----
  const int co = 16 * tile + (lane & 15), px = p0 + 4 * (lane >> 4);
  if (co >= a.C_out || px >= a.HW) return;
  const float bv = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    v[r] += bv;
    if (ACT == 1) v[r] = fmaxf(v[r], 0.0f);
  }
  if (RES == 1) {
    const float4v r4 = *reinterpret_cast<const float4v*>(a.residual + static_cast<size_t>(b) * a.residual_batch_stride + static_cast<size_t>(co) * a.HW + px);
    v += r4;
  }
  if (RES == 2) {
    const int y = px / a.W, xx = px - y * a.W;      // (W % 4 == 0: the four pixels lie in one row)
    const float* r = a.residual + static_cast<size_t>(b) * a.residual_batch_stride + static_cast<size_t>(co) * (a.HW >> 2) + (y >> 1) * (a.W >> 1) + (xx >> 1);
    const float r0 = r[0], r1 = r[1];
    v[0] += r0; v[1] += r0; v[2] += r1; v[3] += r1;
  }
  *reinterpret_cast<float4v*>(a.dst + static_cast<size_t>(b) * a.dst_batch_stride + static_cast<size_t>(co) * a.HW + px) = v;
}

// Input-channel splits of a problem.  Every launch lies within ~2 us of an empty kernel's 3.4 us (tools/pointwise_probe.py), and what it adds is its
// chain of memory round trips: one per round of four quads, the second of a wave hidden behind the first.  So: four quads per wave (one round)
// where the channels allow, at most eight splits -- sixteen make the workgroup's start and its fixed-order sum through LDS cost more than the
// third round they save (1 152 -> 192 channels on the 8 x 10 map: 15.3 / 9.0 / 6.3 / 6.1 / 7.2 us with 1 / 2 / 4 / 8 / 16 splits).
inline int pointwise_conv_splits(int B, int C_in, int C_out, int HW) {
  const int n_quads = (C_in + 15) / 16;
  int splits = (n_quads + kPwRound - 1) / kPwRound;
  if (splits > 8) splits = 8;
  const int per = (n_quads + splits - 1) / splits;
  return (n_quads + per - 1) / per;
}

inline bool pointwise_conv_supports(int B, int C_in, int H, int W, int C_out, int activation, int residual_mode) {
  if (B <= 0 || B > 65535 || C_in <= 0 || C_out <= 0 || H <= 0 || W <= 0) return false;
  const long long HW = static_cast<long long>(H) * W;
  if (C_in % 4 != 0 || HW % 4 != 0) return false;                                  // whole MFMA groups; float4 stores
  if (HW * C_in >= (1LL << 29) || HW * C_out >= (1LL << 29)) return false;          // 32-bit byte offsets
  if ((HW + 15) / 16 > 0x7fffffffLL || (C_out + 15) / 16 > 65535) return false;
  if (activation != 0 && activation != 1) return false;
  if (residual_mode < 0 || residual_mode > 2) return false;
  if (residual_mode == 2 && ((H & 1) || (W & 3))) return false;
  return true;
}

}  // namespace dvmvs

extern "C" int dvmvs_pointwise_conv_supported(int B, int C_in, int H, int W, int C_out, int activation, int residual_mode) {
  return dvmvs::pointwise_conv_supports(B, C_in, H, W, C_out, activation, residual_mode) ? 1 : 0;
}

extern "C" size_t dvmvs_pointwise_conv_packed_bytes(int C_out, int C_in) {
  if (C_out <= 0 || C_in <= 0) return 0;
  return sizeof(float) * 256 * static_cast<size_t>((C_out + 15) / 16) * static_cast<size_t>((C_in + 15) / 16);
}

extern "C" int dvmvs_pointwise_conv_pack(const float* weight, float* packed, int C_out, int C_in, dvmvs_stream_t stream) {
  if (!weight || !packed) return DVMVS_EINVAL;
  const size_t bytes = dvmvs_pointwise_conv_packed_bytes(C_out, C_in);
  if (bytes == 0) return DVMVS_EINVAL;
  const long long total = static_cast<long long>(bytes / sizeof(float));
  hipLaunchKernelGGL(dvmvs::pointwise_conv_pack_kernel, dim3(static_cast<unsigned int>((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     weight, packed, C_out, C_in, (C_in + 15) / 16, total);
  return dvmvs::launch_status();
}

extern "C" int dvmvs_pointwise_conv_fwd(const float* x, long long x_batch_stride, const float* packed, const float* bias, const float* residual,
                                        long long residual_batch_stride, int residual_mode, float* dst, long long dst_batch_stride, int B, int C_in,
                                        int H, int W, int C_out, int activation, int splits, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!x || !packed || !dst || B <= 0 || C_in <= 0 || C_out <= 0 || H <= 0 || W <= 0) return DVMVS_EINVAL;
  if (residual_mode < 0 || residual_mode > 2 || (residual_mode != 0 && !residual) || splits < 0 || splits > 16) return DVMVS_EINVAL;
  if (!pointwise_conv_supports(B, C_in, H, W, C_out, activation, residual_mode)) return DVMVS_EUNSUPPORTED;
  const int HW = H * W;
  if (x_batch_stride == 0) x_batch_stride = static_cast<long long>(C_in) * HW;
  if (dst_batch_stride == 0) dst_batch_stride = static_cast<long long>(C_out) * HW;
  if (residual_batch_stride == 0) residual_batch_stride = static_cast<long long>(C_out) * (residual_mode == 2 ? HW / 4 : HW);
  if (x_batch_stride < static_cast<long long>(C_in) * HW || dst_batch_stride < static_cast<long long>(C_out) * HW) return DVMVS_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) & 3) || (reinterpret_cast<uintptr_t>(dst) & 15) || (dst_batch_stride & 3) ||
      (residual_mode == 1 && ((reinterpret_cast<uintptr_t>(residual) & 15) || (residual_batch_stride & 3))))
    return DVMVS_EUNSUPPORTED;
  PointwiseConvArgs a;
  a.x = x; a.packed = packed; a.bias = bias; a.residual = residual; a.dst = dst;
  a.x_batch_stride = x_batch_stride; a.dst_batch_stride = dst_batch_stride; a.residual_batch_stride = residual_batch_stride;
  a.C_in = C_in; a.C_out = C_out; a.HW = HW; a.W = W;
  a.n_quads = (C_in + 15) / 16;
  if (splits == 0) splits = pointwise_conv_splits(B, C_in, C_out, HW);
  if (splits > a.n_quads) splits = a.n_quads;
  a.quads_per_split = (a.n_quads + splits - 1) / splits;
  splits = (a.n_quads + a.quads_per_split - 1) / a.quads_per_split;
  const dim3 grid(static_cast<unsigned int>((HW + 15) / 16), static_cast<unsigned int>((C_out + 15) / 16), static_cast<unsigned int>(B)), block(64 * splits);
  const size_t lds = splits > 1 ? sizeof(float) * 256 * splits : 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
#define DVMVS_PW(A, R) hipLaunchKernelGGL((pointwise_conv_kernel<A, R>), grid, block, lds, s, a)
  if (activation == 0) { if (residual_mode == 0) DVMVS_PW(0, 0); else if (residual_mode == 1) DVMVS_PW(0, 1); else DVMVS_PW(0, 2); }
  else { if (residual_mode == 0) DVMVS_PW(1, 0); else if (residual_mode == 1) DVMVS_PW(1, 1); else DVMVS_PW(1, 2); }
#undef DVMVS_PW
  return launch_status();
}
