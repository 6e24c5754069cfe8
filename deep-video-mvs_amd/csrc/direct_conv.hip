// Dense 3x3 / 5x5 convolutions of a frame's 1/8 ... full-resolution maps at small batch: direct fp32-MFMA convolution with the bias
// add and ReLU in its store path, gfx950.  Plus the one-output-channel depth heads.
//
// Which layers (fusionnet/model.py:167-305, pairnet alike): the cost-volume encoder's aggregators and blocks, the decoder's
// up-convolutions, blocks and refinement, the FPN's 3x3 smoothing layers and the stem -- 30 layers, 27.5 GFLOP of a 31.5 GFLOP frame.
// Why not MIOpen for them (it was the path of rounds 1-3): at batch 1 its immediate-mode choice is the fp32 Winograd F(2,3) kernel
// or an im2col GEMM with a floor of 14-20 us per layer whatever the size (0.4 GFLOP layers: 19-20 us = 19 TFLOP/s; the
// one-output-channel depth heads: 13-27 us for 1-47 MFLOP) and 42-54 TFLOP/s on the 5x5 layers (tools/conv_layer_probe.py:
// 871 us of a frame).  The layers are small GEMMs -- M = 1 280 ... 81 920 pixels, N = 32 ... 128, K = 288 ... 2 400 -- whose problem
// at batch 1 is filling 1 024 SIMDs, not arithmetic: here every launch is cut into >= 2 048 waves by splitting the INPUT CHANNELS
// over the waves of a workgroup where the pixels alone do not give that many (deterministic: the splits are added in a fixed order
// through LDS, no atomics).
//
// Formulation.  out[co, p] = sum_{ci, ky, kx} x[ci, p*S + (ky, kx) - K/2] * W[co, ci, ky, kx].  One v_mfma_f32_16x16x4_f32 takes
// A = 16 pixels x 4 input channels and B = 4 input channels x 16 output channels for one tap; an fp32 MFMA is an fmaf chain over k
// (exact fp32, cdna guide).  A wave owns MT = 5 16-pixel tiles (MW x MH pixels each: 16 x 1 or 8 x 2) = 80 pixels x NT 16-channel
// tiles of the output; a workgroup is PW pixel-waves (stacked in y) x KS channel-split waves = 8 waves that share one input patch,
// staged per chunk of 4*KS*G input channels into LDS with its zero padding, so the A operand of every MFMA is ONE ds_read_b32 at
// (channel plane + row + tap offset): no im2col operand exists.  Plane and row strides of the patch are chosen so that the 64 lanes
// of that read fall on 64 different banks (channel stride = 16 mod 64, row stride = 8 mod 16 for the two-row tiles).
// Balance: all workgroups of a launch are resident at once and MFMA-bound, so a launch lasts as long as its busiest CU -- 320
// workgroups on 256 CUs cost as much as 512.  The four shapes below cut the maps of a 320x256 frame into EXACTLY 256 workgroups
// (full resolution: 4 rows x 80 columns x 32 channels, 2 splits; 1/2: 2 x 40 x 32, 8 splits; 1/4: 1 x 80 x 16, 8 splits; the 1/8
// maps give 128), i.e. two waves per SIMD everywhere; the shape is chosen per problem by the fraction of CU-rounds it fills.
// The weights are constants at inference: packed once into B-operand order -- [16*NT-channel tile][4-channel group][ky][quad][lane]
// float4 -- and streamed from L2 with coalesced 1 KB loads, one (group, ky) step ahead of the MFMAs that use them; the next chunk's
// patch is requested after the last weight request of the current chunk, so that the MFMAs of the last two steps wait for weights
// only (s_waitcnt vmcnt(n) retires in order).
#include <stdlib.h>

#include "dvmvs_device.h"

namespace dvmvs {

typedef float float4v __attribute__((ext_vector_type(4)));

struct DirectConvArgs {
  const float* x;        // [B, C_in, H, W]; batch item b at x + b * x_batch_stride, its planes dense
  const float* packed;   // direct_conv_pack_kernel's layout
  const float* bias;     // [C_out] or null
  float* dst;            // [B, C_out, OH, OW]; batch item b at dst + b * dst_batch_stride, its planes dense
  float* dst_nhwc;       // null, or a second, channels-last copy of the output [B, OH, OW, C_out] (dense), written in the same epilogue
  long long x_batch_stride, dst_batch_stride;
  int B, C_in, C_out, H, W, OH, OW;
  int act;               // 0 none, 1 ReLU
  int n_groups;          // ceil(C_in / 4)
  int packed_groups;     // groups per output-channel tile in `packed` (n_groups rounded up to kDcGroupPad)
  int n_chunks;          // ceil(n_groups / (KS * G))
  int tiles_x;           // workgroup tiles per output row
  int vec4;              // the destination allows 16-byte stores
};

// ---- optional timeline instrumentation (tools/direct_conv_trace.py; built only by `make trace`): per wave, on the 100 MHz wall clock:
// [0] start, [1] first patch staged, [2] sum over chunks of "requests issued", [3] of "MFMAs", [4] of "patch stored + barrier",
// [5] chunk loop left, [6] channel splits added, [7] end ----
#ifdef DVMVS_SWEEP_TRACE
constexpr int kDcTraceWords = 8, kDcTraceWaves = 16384;
__device__ unsigned long long g_dc_trace[kDcTraceWaves * kDcTraceWords];
#define DC_TRACE(...) __VA_ARGS__
#define DC_NOW() __builtin_amdgcn_s_memrealtime()
#else
#define DC_TRACE(...)
#endif

constexpr int kDcGroupPad = 16;   // packed input-channel groups are padded (with zeros) to a multiple of every KS * G in use

__host__ __device__ constexpr int dc_round_to_residue(int v, int mod, int res) { return v + ((res - v) % mod + mod) % mod; }

// K: filter size (3, 5; padding K / 2); S: stride (1, 2); MW: pixels of a 16-pixel MFMA tile along x (16: one row; 8: two rows);
// MT: tiles per wave along x; NT: 16-channel output tiles per wave; PW x KS: pixel-waves x channel-split waves of the 8-wave
// workgroup; G: groups of 4 input channels a wave takes per chunk.
template <int K_, int S_, int MW_, int MT_, int NT_, int PW_, int KS_, int G_>
struct DirectConvConfig {
  static constexpr int K = K_, S = S_, MW = MW_, MT = MT_, NTILE = NT_, PW = PW_, KS = KS_, G = G_;
  static constexpr int MH = 16 / MW;
  static constexpr int NWAVES = PW * KS, NT = 64 * NWAVES;
  static constexpr int CIC = 4 * KS * G;                        // input channels per staged chunk
  static constexpr int TILE_H = PW * MH, TILE_W = MW * MT;      // output pixels of a workgroup
  // its input patch: PH rows; columns from the 16-byte boundary 4 floats left of the first output pixel's input column to the one
  // behind the last tap -- every row of the patch is then a run of ALIGNED float4 in the image (widths are multiples of 4), each
  // either inside the image or entirely outside it: one buffer_load_dwordx4 + one ds_write_b128 per four elements
  static constexpr int PH = (TILE_H - 1) * S + K, PWD = TILE_W * S + 8, PWD4 = PWD / 4;
  static constexpr int SHIFT = 4 - K / 2;                                       // patch column of (first output pixel, tap 0)
  static constexpr int RS = (MH == 1) ? PWD : dc_round_to_residue(PWD, 16, 8);  // patch row stride in LDS (floats)
  static constexpr int CS = dc_round_to_residue(PH * RS, 64, 16);               // channel stride
  static constexpr int Q = (NTILE * K + 3) / 4;                 // float4 per lane and (group, ky) step: K taps x NT output-channel tiles
  static constexpr int NS = G * K;                              // steps per chunk and wave
  // PRIVATE (one pixel-wave per workgroup): every wave reads only its own 4 G channels of a chunk, so it stages exactly those, into
  // its own part of the buffer, and the chunk loop needs no workgroup barrier at all -- the waves drift apart, and one wave's
  // requests / LDS stores overlap the other wave's MFMAs on the same SIMD instead of all eight idling the MFMA pipe together
  static constexpr bool PRIVATE = PW == 1;
  static constexpr int STAGE_CH = PRIVATE ? 4 * G : CIC;        // channels one staging unit (wave / workgroup) stages per chunk
  static constexpr int STAGE_NT = PRIVATE ? 64 : NT;            // its threads
  static constexpr int PATCH = STAGE_CH * PH * PWD4;            // float4 elements a staging unit stages per chunk
  static constexpr int XREGS = (PATCH + STAGE_NT - 1) / STAGE_NT;      // ... per thread
  static constexpr int RED_FLOATS = (KS > 1) ? (KS / 2) * PW * MT * NTILE * 256 : 0;    // one round of the split tree
  static constexpr int BUF = CIC * CS;                          // floats of one staged chunk
  static constexpr bool PREFETCH = XREGS <= 8;                  // the next chunk's patch waits in registers while this one is multiplied
  static constexpr bool DOUBLE = PREFETCH && sizeof(float) * 2 * BUF <= 150 * 1024;    // two patch buffers: the next chunk is written while this one is read
  static constexpr bool WHOLE = DOUBLE && K == 3 && NS * Q <= 6;           // a whole chunk's weights are requested one chunk ahead
  static constexpr int LDS_FLOATS = ((DOUBLE ? 2 : 1) * BUF > RED_FLOATS) ? (DOUBLE ? 2 : 1) * BUF : RED_FLOATS;
  static_assert((MW == 16 || MW == 8) && (TILE_W * S) % 4 == 0 && RS % 4 == 0 && CS % 4 == 0, "tile shape");
  static_assert(NWAVES == 8 && NS >= 2 && (kDcGroupPad % (KS * G)) == 0 && (KS & (KS - 1)) == 0, "workgroup shape");
  static_assert(sizeof(float) * LDS_FLOATS <= 160 * 1024, "LDS");
};

template <class Cfg>
__global__ __launch_bounds__(Cfg::NT) void direct_conv_kernel(DirectConvArgs a) {
  constexpr int K = Cfg::K, S = Cfg::S, MW = Cfg::MW, MT = Cfg::MT, NTILE = Cfg::NTILE, PW = Cfg::PW, KS = Cfg::KS, G = Cfg::G;
  constexpr int MH = Cfg::MH, CIC = Cfg::CIC, PH = Cfg::PH, PWD4 = Cfg::PWD4, RS = Cfg::RS, CS = Cfg::CS, Q = Cfg::Q, NS = Cfg::NS;
  constexpr int XREGS = Cfg::XREGS, PATCH = Cfg::PATCH, STAGE_NT = Cfg::STAGE_NT;
  constexpr bool PRIVATE = Cfg::PRIVATE;
  extern __shared__ __attribute__((aligned(16))) float s_x[];   // [CIC][CS]: the chunk's patch; afterwards the channel-split partial sums

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the per-wave branches are uniform)
  const int pw = wave % PW, ks = wave / PW;
  const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
  const int cot = blockIdx.y, b = blockIdx.z;
  const int oy0 = ty * Cfg::TILE_H, ox0 = tx * Cfg::TILE_W;
  const int iy0 = oy0 * S - K / 2, ix0 = ox0 * S - 4;      // (the patch starts on a 16-byte boundary)
  const int HW = a.H * a.W;
  DC_TRACE(const unsigned long long tr_start = DC_NOW(); unsigned long long tr_first = 0, tr_req = 0, tr_mfma = 0, tr_store = 0, tr_loop = 0, tr_red = 0, tr_a = 0, tr_b = 0, tr_c = 0;
           auto dump_trace = [&]() __attribute__((always_inline)) {
             const unsigned long long tr_end = DC_NOW();
             const size_t wv = (static_cast<size_t>(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * Cfg::NWAVES + (threadIdx.x >> 6);
             if ((threadIdx.x & 63) == 0 && wv < kDcTraceWaves) {
               unsigned long long* t = g_dc_trace + wv * kDcTraceWords;
               t[0] = tr_start; t[1] = tr_first; t[2] = tr_req; t[3] = tr_mfma; t[4] = tr_store; t[5] = tr_loop; t[6] = tr_red; t[7] = tr_end;
             }
           };)
  gcfloat_p xg = as_global(a.x) + static_cast<size_t>(b) * a.x_batch_stride;

  // ---- patch staging: float4 element e = tid + i * NT of the chunk's [CIC][PH][PWD4] patch, zero outside the image / beyond C_in.
  // Raw buffer descriptor over this batch item's input: a load whose byte offset is >= num_records returns 0 without touching memory
  // -- the zero padding ring costs no branch --, an element's offset inside a chunk never changes (computed once, up front) and the
  // chunk's channel offset rides in the scalar offset operand: a staged element is one buffer_load_dwordx4, no address arithmetic
  // (per-element bounds branches + 64-bit address multiplies were ~600 instructions per chunk in front of 45 MFMAs; dword elements
  // kept the eight waves of a workgroup 0.6 us per chunk in the memory pipeline's issue queue, tools/direct_conv_trace.py).
  // PREFETCH: the next chunk's elements wait in registers while this chunk is multiplied (stride-2 patches are too large for that:
  // they are staged in pieces of four elements per thread between the chunk's barriers) ----
  constexpr bool PREFETCH = Cfg::PREFETCH, DOUBLE = Cfg::DOUBLE, WHOLE = Cfg::WHOLE;
  constexpr int BUF = Cfg::BUF;
  const __amdgpu_buffer_rsrc_t x_resource =
      __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, static_cast<int>(sizeof(float) * static_cast<unsigned int>(a.C_in) * HW), 0x00020000);
  constexpr unsigned int kOutOfRange = 0x80000000u;      // > any offset inside an input (inputs are < 2 GiB, checked on the host)
  const int stage_tid = PRIVATE ? lane : tid;                   // this thread's place in its staging unit (its wave / the workgroup) ...
  const int stage_ch0 = PRIVATE ? ks * G * 4 : 0;               // ... and the unit's first channel within a chunk
  auto element_offset = [&](int i) __attribute__((always_inline)) -> unsigned int {
    const int e = stage_tid + i * STAGE_NT;
    const int ch = e / (PH * PWD4), rem = e - ch * (PH * PWD4);
    const int row = rem / PWD4, col4 = rem - row * PWD4;
    const int iy = iy0 + row, ix = ix0 + 4 * col4;
    const bool in = e < PATCH && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;      // (ix and W are multiples of 4: all four columns or none)
    return in ? static_cast<unsigned int>(sizeof(float)) * static_cast<unsigned int>((stage_ch0 + ch) * HW + iy * a.W + ix) : kOutOfRange;
  };
  unsigned int x_offset[PREFETCH ? XREGS : 1];
  if (PREFETCH) {
#pragma unroll
    for (int i = 0; i < XREGS; ++i) x_offset[i] = element_offset(i);
  }
  auto load_element = [&](int chunk, int i) __attribute__((always_inline)) -> float4v {
    const unsigned int offset = PREFETCH ? x_offset[i] : element_offset(i);
    const bool live = chunk * CIC + stage_ch0 + (stage_tid + i * STAGE_NT) / (PH * PWD4) < a.C_in;      // (a ragged last chunk, and the chunk behind the last)
    return __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(x_resource, static_cast<int>(live ? offset : kOutOfRange),
                                                                             static_cast<int>(sizeof(float) * static_cast<unsigned int>(chunk * CIC) * HW), 0));
  };
  auto store_element = [&](float* buffer, int i, float4v v) __attribute__((always_inline)) {
    const int e = stage_tid + i * STAGE_NT;
    const int ch = e / (PH * PWD4), rem = e - ch * (PH * PWD4);
    const int row = rem / PWD4, col4 = rem - row * PWD4;
    if (e < PATCH) *reinterpret_cast<float4v*>(buffer + (stage_ch0 + ch) * CS + row * RS + 4 * col4) = v;
  };
  float4v xr[PREFETCH ? XREGS : 1];
  auto load_patch = [&](int chunk) __attribute__((always_inline)) {
    if (PREFETCH) {
#pragma unroll
      for (int i = 0; i < XREGS; ++i) xr[i] = load_element(chunk, i);
    }
  };
  auto store_patch = [&](int chunk, float* buffer) __attribute__((always_inline)) {
    if (PREFETCH) {
#pragma unroll
      for (int i = 0; i < XREGS; ++i) store_element(buffer, i, xr[i]);
    } else {
#pragma unroll 1
      for (int i0 = 0; i0 < XREGS; i0 += 4) {
        float4v piece[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) piece[i] = load_element(chunk, i0 + i);      // (elements beyond the patch load nothing)
#pragma unroll
        for (int i = 0; i < 4; ++i) store_element(buffer, i0 + i, piece[i]);
      }
    }
  };

  // between a chunk's last LDS read and the next patch's first use: a workgroup barrier, or -- PRIVATE -- nothing but program order (a
  // wave's LDS instructions execute in order, and nobody else touches its channels)
  auto chunk_barrier = [&]() __attribute__((always_inline)) {
    if (PRIVATE) __builtin_amdgcn_wave_barrier();
    else __syncthreads();
  };

  // ---- weights: step (group cg, ky) = Q float4 per lane ----
  const float4v DVMVS_GLOBAL* wq = reinterpret_cast<const float4v DVMVS_GLOBAL*>(as_global(a.packed)) +
                                   static_cast<size_t>(cot) * a.packed_groups * (K * Q * 64) + lane;
  auto load_weights = [&](float4v* w, int cg, int ky) __attribute__((always_inline)) {
    cg = min(cg, a.packed_groups - 1);      // (requests for the chunk behind the last one are issued unconditionally, see below)
#pragma unroll
    for (int q = 0; q < Q; ++q) w[q] = wq[(static_cast<size_t>(cg) * K + ky) * (Q * 64) + q * 64];
  };

  // this lane's A-operand position: input channel (lane >> 4) of a group of four, pixel (lane & 15) of a tile
  const int m = lane & 15;
  const int a_base = ((lane >> 4) + ks * G * 4) * CS + ((pw * MH + m / MW) * S) * RS + (m % MW) * S + Cfg::SHIFT;

  float4v acc[MT][NTILE];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTILE; ++nt) acc[mt][nt] = float4v{0.0f, 0.0f, 0.0f, 0.0f};

  // one (group gg, row ky) step of a chunk: K taps x MT pixel tiles x NTILE channel tiles, A from the patch in `buffer`, B = w
  auto multiply_step = [&](const float* buffer, int gg, int ky, const float4v* w) __attribute__((always_inline)) {
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float av = buffer[a_base + gg * 4 * CS + ky * RS + kx + mt * MW * S];
#pragma unroll
        for (int nt = 0; nt < NTILE; ++nt) {
          const int j = kx * NTILE + nt;
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w[j / 4][j % 4], acc[mt][nt], 0, 0, 0);
        }
      }
    }
  };

  // Requests for the NEXT chunk (its patch, its weights) are issued unconditionally, also behind the last chunk, where the patch
  // requests are all out of range and the weights are not used: with requests and their waits under conditions the compiler's
  // s_waitcnt placement waits for the new requests in front of this chunk's MFMAs.
  if (WHOLE) {
    // 3x3 layers: all NS x Q quads of a chunk's weights and the chunk's patch are requested one whole chunk ahead, nothing is
    // requested in between (s_waitcnt vmcnt retires in order: any younger request would drag the wait for the patch forward), the
    // patch goes into the other LDS buffer when this chunk's MFMAs are done: one barrier per chunk.  Two weight register sets swap
    // roles from chunk to chunk (the loop is unrolled by two): copying "ahead" into "current" makes the compiler wait for the
    // requests right where they are issued.
    float4v w_even[NS][Q], w_odd[NS][Q];
    auto whole_chunk = [&](int chunk, float4v(*w_use)[Q], float4v(*w_load)[Q]) __attribute__((always_inline)) {
      const int cg0 = (chunk * KS + ks) * G;
      const float* buffer = s_x + (chunk & 1) * BUF;
      DC_TRACE(tr_a = DC_NOW(); __builtin_amdgcn_sched_barrier(0);)
#pragma unroll
      for (int s = 0; s < NS; ++s) load_weights(w_load[s], ((chunk + 1) * KS + ks) * G + s / K, s % K);
      load_patch(chunk + 1);
      __builtin_amdgcn_sched_barrier(0);   // requests stay in front of the chunk's MFMAs (the scheduler otherwise sinks them to their uses)
      DC_TRACE(tr_b = DC_NOW(); __builtin_amdgcn_sched_barrier(0);)
#pragma unroll
      for (int s = 0; s < NS; ++s)
        if (cg0 + s / K < a.n_groups) multiply_step(buffer, s / K, s % K, w_use[s]);      // wave-uniform: groups beyond C_in are zero padding
      DC_TRACE(__builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[MT - 1][NTILE - 1])); tr_c = DC_NOW(); __builtin_amdgcn_sched_barrier(0);)
      // the padding elements of the quads stay "in use" until here: otherwise their registers are handed out while the quad's load is
      // still in flight, and that write-after-write hazard costs an s_waitcnt vmcnt(0) in front of the MFMAs
#pragma unroll
      for (int s = 0; s < NS; ++s) asm volatile("" ::"v"(w_load[s][Q - 1]), "v"(w_use[s][Q - 1]));
      store_patch(chunk + 1, s_x + ((chunk + 1) & 1) * BUF);
      chunk_barrier();
      DC_TRACE(tr_req += tr_b - tr_a; tr_mfma += tr_c - tr_b; tr_store += DC_NOW() - tr_c;)
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) load_weights(w_even[s], ks * G + s / K, s % K);
    load_patch(0);
    store_patch(0, s_x);
    chunk_barrier();
    DC_TRACE(tr_first = DC_NOW();)
    for (int chunk = 0; chunk < a.n_chunks; chunk += 2) {
      whole_chunk(chunk, w_even, w_odd);
      if (chunk + 1 < a.n_chunks) whole_chunk(chunk + 1, w_odd, w_even);
    }
  } else {
    // 5x5 layers: weights one (group, ky) step ahead of the MFMAs that use them; the next chunk's patch is requested after the last
    // weight request of this chunk, so that the MFMAs of the last two steps wait for weights only
    float4v wb[2][Q], w_next[Q];
    load_patch(0);
    load_weights(w_next, ks * G, 0);
    store_patch(0, s_x);
    chunk_barrier();
    DC_TRACE(tr_first = DC_NOW();)
    for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
      const bool more = chunk + 1 < a.n_chunks;
      const int cg0 = (chunk * KS + ks) * G;
      const float* buffer = s_x + (DOUBLE ? (chunk & 1) * BUF : 0);
#pragma unroll
      for (int q = 0; q < Q; ++q) wb[0][q] = w_next[q];
      DC_TRACE(tr_a = tr_b = DC_NOW(); __builtin_amdgcn_sched_barrier(0);)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) load_weights(wb[(s + 1) & 1], cg0 + (s + 1) / K, (s + 1) % K);
        if (s == NS - 2 && (more || DOUBLE)) load_patch(chunk + 1);                  // (younger than every weight request of this chunk)
        if (s == NS - 1 && (more || DOUBLE)) load_weights(w_next, ((chunk + 1) * KS + ks) * G, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (cg0 + s / K < a.n_groups) multiply_step(buffer, s / K, s % K, wb[s & 1]);
        asm volatile("" ::"v"(wb[s & 1][Q - 1]));      // (as above)
      }
      DC_TRACE(__builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[MT - 1][NTILE - 1])); tr_c = DC_NOW(); __builtin_amdgcn_sched_barrier(0);)
      if (DOUBLE) {
        store_patch(chunk + 1, s_x + ((chunk + 1) & 1) * BUF);
        chunk_barrier();
        DC_TRACE(tr_mfma += tr_c - tr_b; tr_store += DC_NOW() - tr_c;)
      } else {
        chunk_barrier();       // every wave has read its part of this chunk's patch
        if (more) {
          store_patch(chunk + 1, s_x);
          chunk_barrier();
        }
      }
    }
  }

  DC_TRACE(tr_loop = DC_NOW();)
  if (PRIVATE && KS > 1) __syncthreads();      // (the split tree reuses the patch buffers: every wave is done with its own)
  // ---- channel splits: a fixed binary tree through LDS -- (0 + 4) + (2 + 6) + ((1 + 5) + (3 + 7)) for eight --, each round the
  // upper half of the remaining waves hands its sums to the lower half ----
  if (KS > 1) {
    float4v* red = reinterpret_cast<float4v*>(s_x);
#pragma unroll
    for (int half = KS / 2; half >= 1; half >>= 1) {
      if (ks >= half && ks < 2 * half) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTILE; ++nt) red[((((ks - half) * PW + pw) * MT + mt) * NTILE + nt) * 64 + lane] = acc[mt][nt];
      }
      __syncthreads();
      if (ks < half) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTILE; ++nt) acc[mt][nt] += red[(((ks * PW + pw) * MT + mt) * NTILE + nt) * 64 + lane];
      }
      if (half > 1) __syncthreads();
    }
    if (ks > 0) {
      DC_TRACE(tr_red = DC_NOW(); dump_trace();)
      return;
    }
  }

  // ---- epilogue: D[row = (lane >> 4) * 4 + r][col = lane & 15] = pixel (lane >> 4) * 4 + r of the tile, output channel lane & 15 ----
  const int m0 = (lane >> 4) * 4;
  const int oy = oy0 + pw * MH + m0 / MW;
  DC_TRACE(tr_red = DC_NOW();)
  if (oy >= a.OH) {
    DC_TRACE(dump_trace();)
    return;
  }
  gfloat_p dst = as_global(a.dst) + static_cast<size_t>(b) * a.dst_batch_stride;
#pragma unroll
  for (int nt = 0; nt < NTILE; ++nt) {
    const int co = cot * (16 * NTILE) + nt * 16 + (lane & 15);
    if (co >= a.C_out) continue;
    const float bv = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int ox = ox0 + mt * MW + m0 % MW;
      float4v v = acc[mt][nt] + bv;
      if (a.act == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
      }
      gfloat_p d = dst + (static_cast<size_t>(co) * a.OH + oy) * a.OW + ox;
      if (a.vec4 && ox + 3 < a.OW) {
        *reinterpret_cast<float4v DVMVS_GLOBAL*>(d) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ox + r < a.OW) d[r] = v[r];
      }
      if (a.dst_nhwc) {      // the same values once more, channels-last: lanes 0..15 of a row hold 16 consecutive channels of one pixel (64-byte runs)
        gfloat_p dn = as_global(a.dst_nhwc) + ((static_cast<size_t>(b) * a.OH + oy) * a.OW + ox) * a.C_out + co;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ox + r < a.OW) dn[static_cast<size_t>(r) * a.C_out] = v[r];
      }
    }
  }
  DC_TRACE(dump_trace();)
}

// packed[((((cot * GP + cg) * K + ky) * Q + q) * 64 + lane) * 4 + e] = W[16 NT cot + 16 nt + (lane & 15)][4 cg + (lane >> 4)][ky][kx],
// (kx, nt) = ((4 q + e) / NT, (4 q + e) % NT), Q = ceil(NT K / 4); zero beyond C_in, C_out and the K taps
__global__ __launch_bounds__(256) void direct_conv_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int C_out, int C_in, int K,
                                                               int n_tile, int packed_groups, long long total) {
  const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= total) return;
  const int Q = (n_tile * K + 3) / 4;
  const int e = static_cast<int>(i & 3), lane = static_cast<int>((i >> 2) & 63);
  long long rest = i >> 8;
  const int q = static_cast<int>(rest % Q);
  rest /= Q;
  const int ky = static_cast<int>(rest % K);
  rest /= K;
  const int cg = static_cast<int>(rest % packed_groups), cot = static_cast<int>(rest / packed_groups);
  const int j = 4 * q + e, kx = j / n_tile, nt = j % n_tile;
  const int co = cot * (16 * n_tile) + nt * 16 + (lane & 15), ci = cg * 4 + (lane >> 4);
  packed[i] = (kx < K && co < C_out && ci < C_in) ? w[((static_cast<size_t>(co) * C_in + ci) * K + ky) * K + kx] : 0.0f;
}

// ---- one-output-channel 3x3 heads (the decoder's depth layers, fusionnet/model.py:196-232) -------------------------------------------
// A workgroup = PX consecutive pixels x 256 / PX slices of the input channels (pixel = fastest thread index: the nine taps of a channel
// are row-contiguous, cached loads); the slices are added through LDS in ascending order.  ACT as dvmvs_bias_act_fwd (0 none, 1 ReLU,
// 2 sigmoid, 3 sigmoid then the depth mapping 1 / (p0 * s + p1)); RAW: no bias, no activation (the consumer applies them).
__device__ inline float head_activation(float v, int act, float p0, float p1) {
#pragma clang fp contract(off)
  if (act == 1) return fmaxf(v, 0.0f);
  if (act == 2 || act == 3) {
    const float s = 1.0f / (1.0f + expf(-v));
    return act == 2 ? s : 1.0f / (p0 * s + p1);
  }
  return v;
}

template <int PX>
__global__ __launch_bounds__(256) void conv_head_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ dst, long long x_batch_stride, long long dst_batch_stride, int C, int H,
                                                        int W, int act, float p0, float p1) {
  constexpr int SL = 256 / PX;
  __shared__ float s_part[SL][PX];
  const int px = threadIdx.x % PX, slice = threadIdx.x / PX;
  const int HW = H * W;
  const int p = blockIdx.x * PX + px, b = blockIdx.y;
  const int y = p / W, xx = p - y * W;
  const int per = (C + SL - 1) / SL;
  const int c_lo = slice * per, c_hi = min(C, c_lo + per);
  float sum = 0.0f;
  if (p < HW) {
    int off[9];
    bool in[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + t / 3 - 1, xs = xx + t % 3 - 1;
      in[t] = yy >= 0 && yy < H && xs >= 0 && xs < W;
      off[t] = in[t] ? yy * W + xs : 0;
    }
    const float* xb = x + static_cast<size_t>(b) * x_batch_stride;
#pragma unroll 2
    for (int c = c_lo; c < c_hi; ++c) {
      const float* plane = xb + static_cast<size_t>(c) * HW;
      float v[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) v[t] = in[t] ? plane[off[t]] : 0.0f;
#pragma unroll
      for (int t = 0; t < 9; ++t) sum = fmaf(v[t], w[c * 9 + t], sum);
    }
  }
  s_part[slice][px] = sum;
  __syncthreads();
  if (slice == 0 && p < HW) {
    float total = s_part[0][px];
#pragma unroll
    for (int s = 1; s < SL; ++s) total += s_part[s][px];
    if (bias) total += bias[0];
    dst[static_cast<size_t>(b) * dst_batch_stride + p] = head_activation(total, act, p0, p1);
  }
}

// ---- configuration choice -----------------------------------------------------------------------------------------------------------------
// id: 1 = 4 rows x 80 columns x 32 channels, 2 splits; 2 = 2 x 40 x 32, 8 splits; 3 = 1 x 80 x 16, 8 splits; 4 = 2 x 40 x 16, 8 splits.
struct DirectConvShape {
  int MW, MT, NT, PW, KS;
};
constexpr DirectConvShape kDcShapes[4] = {{16, 5, 2, 4, 2}, {8, 5, 2, 1, 8}, {16, 5, 1, 1, 8}, {8, 5, 1, 1, 8}};
constexpr int kDcCUs = 256;

inline int direct_conv_choose(int B, int C_in, int C_out, int H, int W, int K, int S) {
  if (B <= 0 || C_in <= 0 || C_out <= 0 || C_out % 16 != 0 || (K != 3 && K != 5) || (S != 1 && S != 2) || H <= 0 || W <= 0) return 0;
  if (S == 2 && (H % 2 != 0 || W % 2 != 0)) return 0;
  const int OH = H / S, OW = W / S;       // "same" padding K / 2: (H + 2 (K / 2) - K) / S + 1
#if defined(DVMVS_SWEEP_TUNING)          // tools-only build: DVMVS_DC_SHAPE=1..4 forces the workgroup shape (tools/direct_conv_probe.py)
  const char* f = getenv("DVMVS_DC_SHAPE");
  const int forced = f ? atoi(f) : 0;
#else
  const int forced = 0;
#endif
  int best = 0;
  double best_fill = 0.0;
  for (int id = 1; id <= 4; ++id) {
    const DirectConvShape& s = kDcShapes[id - 1];
    const int tile_h = s.PW * (16 / s.MW), tile_w = s.MW * s.MT;
    if (OW % tile_w != 0 || C_out % (16 * s.NT) != 0) continue;
    if (forced == id) return id;
    if (forced) continue;
    const long long groups = static_cast<long long>(OW / tile_w) * ((OH + tile_h - 1) / tile_h) * (C_out / (16 * s.NT)) * B;
    const double fill = static_cast<double>(groups) / static_cast<double>(kDcCUs * ((groups + kDcCUs - 1) / kDcCUs));   // of the CU-rounds
    if (fill > best_fill + 1e-9) {      // ties: the earlier shape (larger tiles: less patch overlap, more weight reuse)
      best_fill = fill;
      best = id;
    }
  }
  return best;
}

template <class Cfg>
int launch_direct_conv(DirectConvArgs a, hipStream_t stream) {
  static bool configured[64] = {};
  constexpr size_t lds = sizeof(float) * Cfg::LDS_FLOATS;
  if (lds > 64 * 1024) {
    int device = 0;
    DVMVS_RETURN_IF_HIP(hipGetDevice(&device));
    const bool tracked = device >= 0 && device < 64;
    if (!tracked || !configured[device]) {   // per-device function attribute; idempotent, racing threads write the same value
      DVMVS_RETURN_IF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(direct_conv_kernel<Cfg>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(lds)));
      if (tracked) configured[device] = true;
    }
  }
  a.n_chunks = (a.n_groups + Cfg::KS * Cfg::G - 1) / (Cfg::KS * Cfg::G);
  a.tiles_x = a.OW / Cfg::TILE_W;
  const int tiles_y = (a.OH + Cfg::TILE_H - 1) / Cfg::TILE_H;
  const dim3 grid(a.tiles_x * tiles_y, a.C_out / (16 * Cfg::NTILE), a.B), block(Cfg::NT);
  hipLaunchKernelGGL((direct_conv_kernel<Cfg>), grid, block, lds, stream, a);
  return launch_status();
}

template <int K, int S>
int launch_direct_conv_shape(int id, const DirectConvArgs& a, hipStream_t stream) {
  switch (id) {
    case 1: return launch_direct_conv<DirectConvConfig<K, S, 16, 5, 2, 4, 2, 2>>(a, stream);
    case 2: return launch_direct_conv<DirectConvConfig<K, S, 8, 5, 2, 1, 8, 1>>(a, stream);
    case 3: return launch_direct_conv<DirectConvConfig<K, S, 16, 5, 1, 1, 8, 1>>(a, stream);
    case 4: return launch_direct_conv<DirectConvConfig<K, S, 8, 5, 1, 1, 8, 1>>(a, stream);
  }
  return DVMVS_EUNSUPPORTED;
}

inline int dc_packed_groups(int C_in) { return ((C_in + 3) / 4 + kDcGroupPad - 1) / kDcGroupPad * kDcGroupPad; }

}  // namespace dvmvs

// 0: not taken by the direct kernel (the caller keeps MIOpen); else the number of 16-channel output tiles per wave (1 or 2) that
// the weights have to be packed for
extern "C" int dvmvs_direct_conv_tile(int B, int C_in, int H, int W, int C_out, int kernel_size, int stride) {
  const int id = dvmvs::direct_conv_choose(B, C_in, C_out, H, W, kernel_size, stride);
  return id == 0 ? 0 : dvmvs::kDcShapes[id - 1].NT;
}

extern "C" size_t dvmvs_direct_conv_packed_bytes(int C_out, int C_in, int kernel_size, int n_tile) {
  if (C_out <= 0 || C_in <= 0 || (n_tile != 1 && n_tile != 2) || C_out % (16 * n_tile) != 0 || (kernel_size != 3 && kernel_size != 5)) return 0;
  const size_t Q = (n_tile * kernel_size + 3) / 4;
  return sizeof(float) * (C_out / (16 * n_tile)) * dvmvs::dc_packed_groups(C_in) * kernel_size * Q * 256;
}

extern "C" int dvmvs_direct_conv_pack(const float* weight, float* packed, int C_out, int C_in, int kernel_size, int n_tile, dvmvs_stream_t stream) {
  if (!weight || !packed) return DVMVS_EINVAL;
  const size_t bytes = dvmvs_direct_conv_packed_bytes(C_out, C_in, kernel_size, n_tile);
  if (bytes == 0) return DVMVS_EUNSUPPORTED;
  const long long total = static_cast<long long>(bytes / sizeof(float));
  hipLaunchKernelGGL(dvmvs::direct_conv_pack_kernel, dim3(static_cast<unsigned int>((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     weight, packed, C_out, C_in, kernel_size, n_tile, dvmvs::dc_packed_groups(C_in), total);
  return dvmvs::launch_status();
}

extern "C" int dvmvs_direct_conv_fwd(const float* x, long long x_batch_stride, const float* packed, int n_tile, const float* bias, float* dst,
                                     long long dst_batch_stride, int B, int C_in, int H, int W, int C_out, int kernel_size, int stride, int activation,
                                     dvmvs_stream_t stream) {
  return dvmvs_direct_conv_dual_fwd(x, x_batch_stride, packed, n_tile, bias, dst, dst_batch_stride, nullptr, B, C_in, H, W, C_out, kernel_size, stride, activation,
                                    stream);
}

extern "C" int dvmvs_direct_conv_dual_fwd(const float* x, long long x_batch_stride, const float* packed, int n_tile, const float* bias, float* dst,
                                          long long dst_batch_stride, float* dst_nhwc, int B, int C_in, int H, int W, int C_out, int kernel_size, int stride,
                                          int activation, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!x || !packed || !dst || B <= 0) return DVMVS_EINVAL;
  if (activation != 0 && activation != 1) return DVMVS_EUNSUPPORTED;
  const int id = direct_conv_choose(B, C_in, C_out, H, W, kernel_size, stride);
  if (id == 0) return DVMVS_EUNSUPPORTED;
  if (kDcShapes[id - 1].NT != n_tile) return DVMVS_EINVAL;      // the weights were packed for another problem (dvmvs_direct_conv_tile)
  DirectConvArgs a;
  a.x = x; a.packed = packed; a.bias = bias; a.dst = dst; a.dst_nhwc = dst_nhwc;
  a.B = B; a.C_in = C_in; a.C_out = C_out; a.H = H; a.W = W; a.OH = H / stride; a.OW = W / stride;
  a.x_batch_stride = x_batch_stride ? x_batch_stride : static_cast<long long>(C_in) * H * W;
  a.dst_batch_stride = dst_batch_stride ? dst_batch_stride : static_cast<long long>(C_out) * a.OH * a.OW;
  if (a.x_batch_stride < static_cast<long long>(C_in) * H * W || a.dst_batch_stride < static_cast<long long>(C_out) * a.OH * a.OW) return DVMVS_EINVAL;
  if (W % 4 != 0 || a.x_batch_stride % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0) return DVMVS_EUNSUPPORTED;      // aligned float4 rows
  if (static_cast<long long>(C_in + 64) * H * W * 4 >= (1LL << 31)) return DVMVS_EUNSUPPORTED;      // 32-bit byte offsets into one batch item (incl. a padded chunk)
  a.act = activation;
  a.n_groups = (C_in + 3) / 4;
  a.packed_groups = dc_packed_groups(C_in);
  a.vec4 = (a.OW % 4 == 0 && (a.OH * a.OW) % 4 == 0 && a.dst_batch_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) ? 1 : 0;
  a.n_chunks = 0; a.tiles_x = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (kernel_size == 3) return stride == 1 ? launch_direct_conv_shape<3, 1>(id, a, s) : launch_direct_conv_shape<3, 2>(id, a, s);
  return stride == 1 ? launch_direct_conv_shape<5, 1>(id, a, s) : launch_direct_conv_shape<5, 2>(id, a, s);
}

extern "C" int dvmvs_conv_head_fwd(const float* x, long long x_batch_stride, const float* weight, const float* bias, float* dst,
                                   long long dst_batch_stride, int B, int C_in, int H, int W, int activation, float p0, float p1, dvmvs_stream_t stream) {
  using namespace dvmvs;
  if (!x || !weight || !dst || B <= 0 || C_in <= 0 || H <= 0 || W <= 0) return DVMVS_EINVAL;
  if (activation < 0 || activation > 3) return DVMVS_EINVAL;
  const long long HW = static_cast<long long>(H) * W;
  if (x_batch_stride == 0) x_batch_stride = HW * C_in;
  if (dst_batch_stride == 0) dst_batch_stride = HW;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // small maps: 16 pixels x 16 channel slices per workgroup (enough workgroups, short channel loops); large ones: 64 x 4
  if (HW >= 16384) {
    hipLaunchKernelGGL((conv_head_kernel<64>), dim3(static_cast<unsigned int>((HW + 63) / 64), B), dim3(256), 0, s, x, weight, bias, dst, x_batch_stride,
                       dst_batch_stride, C_in, H, W, activation, p0, p1);
  } else {
    hipLaunchKernelGGL((conv_head_kernel<16>), dim3(static_cast<unsigned int>((HW + 15) / 16), B), dim3(256), 0, s, x, weight, bias, dst, x_batch_stride,
                       dst_batch_stride, C_in, H, W, activation, p0, p1);
  }
  return launch_status();
}

#ifdef DVMVS_SWEEP_TRACE
extern "C" int dvmvs_debug_direct_conv_trace(unsigned long long* host, int waves) {
  if (waves > dvmvs::kDcTraceWaves) waves = dvmvs::kDcTraceWaves;
  return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(dvmvs::g_dc_trace), sizeof(unsigned long long) * dvmvs::kDcTraceWords * waves));
}
#endif
