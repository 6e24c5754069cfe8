// Correlate-then-interpolate plane sweep (dot-product mode, up to 32 channels) for gfx950: the fused warp + correlation kernel of the hot
// path on the fp32 matrix cores.  Semantics: /root/reference/dvmvs/utils.py:45-107; CPU restatement: oracle/dvmvs_oracle.py.
//
// cost(p, d) = 1/(C M) sum_m sum_taps w_tap(p, d, m) <f1(p), f2_m(q_tap)>: the bilinear weights are scalars, so the 32-channel dot
// products <f1(p), f2(q)> can be taken per measurement CELL q instead of per tap, and a cell is shared by the planes of a pixel
// (consecutive planes move ~1 px along the epipolar line) and by neighbouring pixels.  The LDS-tiled sweep (sweep_tiled.hip) reads
// 4 taps x 32 channels from LDS for every (pixel, plane, frame): 1.34 GB of ds_read_b128 per op, the pipe that bounds it.  Here
//   * a wave owns a GROUP of 16 reference pixels (GW x GH) and PW = 16 consecutive planes; lane = (pixel p = lane & 15, plane phase
//     q = lane >> 4), its planes are 4 j + q;
//   * per measurement frame every lane evaluates its four sample positions (the reference's fp32 arithmetic, sweep_sample.h), the wave
//     reduces the bounding box of the in-image taps (DPP row reductions, no LDS), and the box's cells, 16 at a time, become the A
//     operand of v_mfma_f32_16x16x4_f32 STRAIGHT FROM THE MEASUREMENT MAP (one 128-byte line per cell when the map is channels-last,
//     eight dword loads otherwise): D[cell][pixel] = sum_c f2[c][cell] f1[c][pixel], eight MFMAs per 16 cells, B = the group's
//     reference features held in 8 registers for the wave's lifetime.  An fp32 MFMA is an fmaf chain over k, so a dot product has one
//     fixed summation order (channels jj, 8 + jj, 16 + jj, 24 + jj for jj = 0..7) whatever the geometry;
//   * the 16 x cells dot table goes to a wave-private LDS slice (one ds_write_b128 per lane and tile: the MFMA leaves a lane with its
//     OWN pixel's dots), and every lane interpolates its samples from it: four ds_read_b32 + four FMAs per (pixel, plane, frame)
//     instead of 32 ds_read_b128 + 128 packed FMAs;
//   * a box of more than 2 CAP cells is redone per 4 planes; any box is processed in strips of CAP cells of its row-major index space
//     (one strip almost always), so no footprint is too large and there is no gather path.  Waves never synchronise with each other:
//     no barrier, no second pass, no work list, no host plan.
// Zeros padding: the box is clipped to the image, taps outside it get weight 0.  Bit-reproducible (no atomics, fixed orders).
#include "sweep_sample.h"

namespace dvmvs {

typedef short short2v __attribute__((ext_vector_type(2)));

template <int CTRL, int ROW_MASK = 0xf>
__device__ inline int dpp_exchange(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false); }

__device__ inline int pk_min_i16(int a, int b) {
  return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}
__device__ inline int pk_max_i16(int a, int b) {
  return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}

// Minimum / maximum of a packed (x | y << 16) pair of int16 over the wave, returned wave-uniform: butterflies inside each row of 16 lanes
// and the two row broadcasts of the GFX9 DPP path (all VALU, no LDS), one readlane.
template <bool MAX>
__device__ inline int wave_reduce_pk_i16(int v) {
  auto op = [](int a, int b) { return MAX ? pk_max_i16(a, b) : pk_min_i16(a, b); };
  v = op(v, dpp_exchange<0xB1>(v));          // quad_perm [1,0,3,2]
  v = op(v, dpp_exchange<0x4E>(v));          // quad_perm [2,3,0,1]
  v = op(v, dpp_exchange<0x141>(v));         // row_half_mirror
  v = op(v, dpp_exchange<0x140>(v));         // row_mirror: every lane holds its row's result
  v = op(v, dpp_exchange<0x142, 0xa>(v));    // row_bcast15 into rows 1 and 3
  v = op(v, dpp_exchange<0x143, 0xc>(v));    // row_bcast31 into rows 2 and 3: lane 63 holds the wave's result
  return __builtin_amdgcn_readlane(v, 63);
}

// Lanes of ONE wave exchange data through its private LDS slice.  The hardware executes a wave's LDS operations in order, so no
// s_barrier and no s_waitcnt is needed between a write and another lane's read; what is needed is that the COMPILER keeps them in
// program order: a wave-scope fence plus a compiler-level memory clobber.
__device__ inline void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// The small per-frame matrices are wave-uniform: read through the constant address space they arrive as scalar loads in SGPRs (the
// vector memory pipe carries the measurement cells), and the next frame's are requested a whole frame ahead.
typedef const float __attribute__((address_space(4)))* ccfloat_p;
__device__ inline ccfloat_p as_constant(const float* p) { return (ccfloat_p)p; }

// GW x GH = 16 reference pixels per wave; CAP = cells of the dot table; CPW = chunks of 16 planes a wave works through one after the
// other (same pixels: the reference features stay in registers); WAVES = waves per SIMD the register allocation is held to.
template <int GW_, int GH_, int CAP_, int CPW_, int WAVES_, int ABLATE_ = 0, int STAGGER_ = 0, int ORDER_ = 0, int WPB_ = 1, int NBUF_ = 2, int PAIR_ = 0, int GATHER_ = 0, int GATHER_UNROLL_ = 1, int QUARTER_ = 1>
struct MfmaSweepConfig {
  static constexpr int GW = GW_, GH = GH_, CAP = CAP_, CPW = CPW_, WAVES = WAVES_;
  // Round 6.  WPB: waves per workgroup.  One-wave workgroups (round 5) are handed to the SIMDs of a CU unevenly -- 4 to 7 of a launch's waves per SIMD
  // where the mean is 5 (profiles/r06_sweep_mfma_v3_simd_histogram.txt) -- and a launch lasts as long as its busiest SIMD; a 256-thread workgroup puts
  // exactly one wave on each SIMD of its CU.  The waves of a workgroup still never synchronise (each has its own table slice): the workgroup is a
  // placement unit only.  With WPB = 4 and a 2-D grid its waves own a 2 x 2 arrangement of pixel groups (8 x 8 pixels: neighbouring footprints, similar cost).
  // NBUF: 16-cell operand tiles in flight per wave (2 or 4).  PAIR: the eight MFMAs of two tiles interleaved on two accumulators (a tile's own
  // eight form a dependent chain: 40 cycles of latency against 32 of issue each).
  static constexpr int WPB = WPB_, NBUF = NBUF_, PAIR = PAIR_;
  // GATHER: a 4-plane pass whose box holds more than this many cells (0: never) takes its 4 x 64 taps straight from the map instead of filling a dot
  // table -- under magnification (forward motion onto near planes) the box of 16 pixels x 4 planes is mostly empty: up to 50 tiles for 256 taps, and
  // one such item alone used to last 100 us.  The gather costs what ~6 tiles cost, whatever the footprint.
  static constexpr int GATHER = GATHER_, GATHER_UNROLL = GATHER_UNROLL_;      // (samples of a gather pass whose loads are in flight together)
  static constexpr int QUARTER = QUARTER_;      // persistent form: the last four queue entries of a workgroup are handed out as quarter items (else whole)
  static_assert(GATHER == 0 || GATHER >= 2 * CAP_, "quarter items and whole items must take the same path per sample: see sweep_mfma_item");
  static constexpr int STAGGER = STAGGER_;             // 1: s_setprio by wave slot (waves of a SIMD leave lockstep: one's MFMA phase beside another's VALU phase)
  static constexpr int ORDER = ORDER_;                 // 1: within an XCD far chunks (more tiles) first
  static constexpr int ABLATE = ABLATE_;               // tools only (timing experiments, wrong results): 1 no operand loads, 2 no MFMAs, 4 no interpolation
  static constexpr int PW = 16;                        // planes per chunk
  static constexpr int SPLIT = 2;                      // a 16-plane box of more than SPLIT * CAP cells is redone per 4 planes (tools/sweep_mfma_model.py)
  static constexpr int PITCH = CAP + 16 + 4;           // floats per pixel row of the table (16-byte aligned rows; the last tile may be partial)
  static constexpr int kTableFloats = 16 * PITCH;
  static constexpr int wave_lds_floats(int M) { return kTableFloats + 4 * PW * CPW * M; }
  static constexpr size_t lds_bytes(int M) { return sizeof(float) * WPB * wave_lds_floats(M); }
  static_assert(GW * GH == 16 && CAP % 16 == 0 && (CPW == 1 || CPW == 2 || CPW == 4), "group shape");
  static_assert((WPB == 1 || WPB == 2 || WPB == 4) && (NBUF == 2 || NBUF == 4), "workgroup shape / operand ring");
};

// ---- optional timeline instrumentation (tools/sweep_mfma_trace.py; built only by `make trace`) -------------------------------------
#ifdef DVMVS_SWEEP_TRACE
constexpr int kMfmaTraceWords = 16, kMfmaTraceWaves = 16384;
__device__ unsigned long long g_sweep_mfma_trace[kMfmaTraceWaves * kMfmaTraceWords];
#define MFMA_TRACE(...) __VA_ARGS__
#else
#define MFMA_TRACE(...)
#endif

constexpr int kMfmaSweepChannels = 32;   // the K extent of a tile's eight MFMAs; fewer channels are padded with zero operands

// this lane's four samples of one (chunk, frame): north-west tap (x | y << 16, int16 each), fractional position, alive bits (some tap
// can lie inside the image)
struct SweepSamples {
  int xy[4];
  float frx[4], fry[4];
  unsigned int alive;
};

#ifdef DVMVS_SWEEP_TRACE
struct MfmaTraceCounters {
  unsigned long long pos = 0, box = 0, tiles = 0, look = 0, ntiles = 0, nstrips = 0, first = 0;
};
#endif

// One work item of one wave: a group of 16 reference pixels (gx, gy) x Cfg::CPW chunks of 16 planes from plane d_wave on, all measurement frames.
// T: the wave's table slice.  ktd: K t / depth per frame and plane, [m * ktd_stride + plane - d_wave] -- the wave's own table (SHARED_KTD false: filled
// here) or the workgroup's (the persistent kernel: filled once for all planes before any item starts).  quarter < 0: the whole item; 0..3 (wave-uniform):
// only the planes 4 quarter + q of (each chunk of) the item -- the pass structure of a box that is redone per 4 planes, one pass only; same arithmetic
// per sample, so four quarter items computed by four waves give the bits of the whole item.
template <class Cfg, bool NHWC, bool FULL, bool SHARED_KTD>
__device__ __forceinline__ void sweep_mfma_item(const CostVolumeArgs& a, int b, int gx, int gy, int d_wave, int lane, float* T, float4v* ktd, int ktd_stride, int quarter
                                                MFMA_TRACE(, MfmaTraceCounters& tr)) {
  constexpr int GW = Cfg::GW, GH = Cfg::GH, PW = Cfg::PW, CAP = Cfg::CAP, CPW = Cfg::CPW, PITCH = Cfg::PITCH;
  constexpr int WP = PW * CPW;   // planes per item
  const int C = FULL ? kMfmaSweepChannels : a.C;
  const int p = lane & 15, q = lane >> 4;
  MFMA_TRACE(unsigned long long &tr_pos = tr.pos, &tr_box = tr.box, &tr_tiles = tr.tiles, &tr_look = tr.look, &tr_ntiles = tr.ntiles, &tr_nstrips = tr.nstrips, &tr_first = tr.first;)
  const int HW = a.H * a.W;
  const int x = gx * GW + p % GW, y = gy * GH + p / GW;
  const bool live = x < a.W && y < a.H;
  const int pix = live ? y * a.W + x : 0;
  const float xf = static_cast<float>(x), yf = static_cast<float>(y);
  const SweepScale sc = sweep_scale(a.W, a.H);
  const unsigned int plane_bytes = static_cast<unsigned int>(HW) * 4u;
  const unsigned int map_bytes = static_cast<unsigned int>(C) * plane_bytes;

  // B operand of every MFMA of this wave: lane (p, q) holds f1[8 q + jj][pixel p], jj = 0..7 (requested first, needed last; the channel
  // rides in the scalar offset: no per-load address arithmetic)
  const __amdgpu_buffer_rsrc_t ref_rsrc = map_resource(as_global(a.image1) + static_cast<size_t>(b) * C * HW, map_bytes);
  float f1v[8];
  {
    const unsigned int vo = (8u * q * static_cast<unsigned int>(HW) + static_cast<unsigned int>(pix)) * 4u;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) f1v[jj] = buffer_f32(ref_rsrc, (FULL || 8 * q + jj < C) ? vo : kBufferOutOfRange, static_cast<unsigned int>(jj) * plane_bytes);
  }

  // ---- K t / depth for every frame and every plane of this wave (utils.py:59-68: the depth as the reference's python-double
  // expression, then an IEEE fp32 division by the fp32-rounded depth): one pass of the wave, before any frame needs it ----
  if (!SHARED_KTD) {
    const int pl = lane & (WP - 1);
    const float depth = plane_depth(a.inv_depth_base, a.inv_depth_step, min(d_wave + pl, a.D - 1));
    gcfloat_p kt_g = as_global(a.kt) + static_cast<size_t>(b) * a.M * 3;
    for (int i = lane; i < a.M * WP; i += 64) {
      const int m = i / WP;
      float4v k = {0.0f, 0.0f, 0.0f, 0.0f};
      if (d_wave + pl < a.D) {
        k.x = kt_g[m * 3 + 0] / depth;
        k.y = kt_g[m * 3 + 1] / depth;
        k.z = kt_g[m * 3 + 2] / depth;
      }
      ktd[i] = k;
    }
    wave_lds_fence();
  }
  const ccfloat_p Hm_c = as_constant(a.Hm) + static_cast<size_t>(b) * a.M * 9;

  // sample positions of chunk c, frame m (the reference's fp32 arithmetic: sweep_sample.h)
  auto positions = [&](int c, int m, SweepSamples& S) {
    float Hm[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Hm[k] = Hm_c[m * 9 + k];
    const SweepRay ray = sweep_ray(Hm, xf, yf);
    S.alive = 0u;
    float4v kd[4];
    float ix[4], iy[4];
    if (quarter < 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) kd[j] = ktd[m * ktd_stride + c * PW + 4 * j + q];
      sweep_samples<2>(ray, kd, sc, ix, iy);   // clamped to [-1, W] x [-1, H]; NaN -> -1; two planes' chains interleaved
      __builtin_amdgcn_sched_barrier(0);
      sweep_samples<2>(ray, kd + 2, sc, ix + 2, iy + 2);
    } else {      // one sample per lane; the other three are dead (-1: outside the image)
      kd[0] = ktd[m * ktd_stride + c * PW + 4 * quarter + q];
      float tx, ty;
      sweep_samples<1>(ray, kd, sc, &tx, &ty);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ix[j] = j == quarter ? tx : -1.0f;
        iy[j] = j == quarter ? ty : -1.0f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool al = live & (d_wave + c * PW + 4 * j + q < a.D) & (ix[j] > -1.0f) & (ix[j] < sc.Wf) & (iy[j] > -1.0f) & (iy[j] < sc.Hf);   // (no short-circuit: one basic block)
      const float fx = floorf(ix[j]), fy = floorf(iy[j]);
      S.xy[j] = (static_cast<int>(fx) & 0xffff) | (static_cast<int>(fy) << 16);
      S.frx[j] = ix[j] - fx;
      S.fry[j] = iy[j] - fy;
      S.alive |= al ? (1u << j) : 0u;
    }
  };

  SweepSamples S;
  MFMA_TRACE(tr_first = __builtin_amdgcn_s_memtime();)

  for (int c = 0; c < CPW; ++c) {
    const int d_block = d_wave + c * PW;
    if (d_block >= a.D) break;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // sum over frames and taps of w <f1, f2>; plane d_block + 4 j + q

    for (int m = 0; m < a.M; ++m) {
      // (round 5, v2 computed the NEXT frame's positions behind the first operand requests: a second sample set, 13 registers, for nothing --
      // the launch is bound by instruction issue, not by the round trip it covered)
      MFMA_TRACE(const unsigned long long tr_p0 = __builtin_amdgcn_s_memtime();)
      positions(c, m, S);
      MFMA_TRACE(asm volatile("s_nop 0" :: "v"(S.xy[0]), "v"(S.xy[1]), "v"(S.xy[2]), "v"(S.xy[3])); tr_pos += __builtin_amdgcn_s_memtime() - tr_p0;)
      const __amdgpu_buffer_rsrc_t meas_rsrc = map_resource(as_global(a.image2[m]) + static_cast<size_t>(b) * C * HW, map_bytes);

      // ---- passes: the 16 planes as one box; a box of more than SPLIT x CAP cells (diagonal or fast epipolar motion: the box is
      // mostly empty) is redone per 4 planes.  A box is processed in STRIPS of CAP cells of its row-major index space: one strip
      // almost always; several under strong magnification, where a tap simply belongs to the strip that holds its cell ----
      // A quarter item is pass `quarter` of the four 4-plane passes.  Whether a 4-plane pass goes through the table or is gathered depends on ITS box
      // alone (more than GATHER cells: gather), and GATHER >= SPLIT x CAP, so a whole item that fits one 16-plane pass (<= SPLIT x CAP cells, hence every
      // 4-plane sub-box too) and its quarter items take the same path for every sample: a sample's arithmetic does not depend on who computes it.
      int n_pass = quarter < 0 ? 1 : 4;
      for (int s = quarter < 0 ? 0 : quarter; s < (quarter < 0 ? n_pass : quarter + 1); ++s) {
        const int jlo = n_pass == 1 ? 0 : s, jhi = n_pass == 1 ? 4 : s + 1;
        MFMA_TRACE(const unsigned long long tr_b0 = __builtin_amdgcn_s_memtime();)
        // bounding box of the taps of the alive samples: north-west taps lie in [-1, W-1] x [-1, H-1], so the box may reach one cell
        // beyond the image -- those cells are requested out of range and arrive as zeros: grid_sample's zeros padding as a zero
        // apron in the table, no per-tap bounds logic
        int lo = 0x7fff7fff, hi = static_cast<int>(0x80008000u);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j >= jlo && j < jhi && ((S.alive >> j) & 1u)) {
            lo = pk_min_i16(lo, S.xy[j]);
            hi = pk_max_i16(hi, S.xy[j]);
          }
        const int hi_u = wave_reduce_pk_i16<true>(hi);
        if (hi_u == static_cast<int>(0x80008000u)) continue;   // no alive sample in this pass: nothing to add
        const int lo_u = wave_reduce_pk_i16<false>(lo);
        const int x_lo = static_cast<short>(lo_u & 0xffff), y_lo = lo_u >> 16;
        const int bw = static_cast<short>(hi_u & 0xffff) + 1 - x_lo + 1, bh = (hi_u >> 16) + 1 - y_lo + 1, cells = bw * bh;
        if (n_pass == 1 && cells > Cfg::SPLIT * CAP) {
          n_pass = 4;
          s = -1;
          continue;
        }
        if (Cfg::GATHER > 0 && n_pass == 4 && cells > Cfg::GATHER) {
          // ---- gather pass: no table.  The four lanes (p, 0..3) of a pixel hold its four samples of this pass (planes 4 s + q) and one channel
          // octet each (f1v: channels 8 q ..): every lane takes ITS octet of all four samples of its pixel -- positions through the wave's LDS slice,
          // 4 taps x 8 channels per sample from the map (two 16-byte loads per tap when it is channels-last), per tap the dot with f1v, the taps
          // weighted -- and the four octet sums of a sample meet in LDS again, added in the fixed order 0, 1, 2, 3 by the sample's own lane. ----
          MFMA_TRACE(tr_box += __builtin_amdgcn_s_memtime() - tr_b0; const unsigned long long tr_g0 = __builtin_amdgcn_s_memtime();)
          float4v* X = reinterpret_cast<float4v*>(T);      // [pixel][q']: (north-west tap x | y << 16, fractional x, fractional y, alive)
          {
            const bool al = (S.alive >> s) & 1u;
            float4v mine;
            mine.x = __builtin_bit_cast(float, S.xy[0]);
#pragma unroll
            for (int j = 1; j < 4; ++j) mine.x = s == j ? __builtin_bit_cast(float, S.xy[j]) : mine.x;
            mine.y = S.frx[0]; mine.z = S.fry[0];
#pragma unroll
            for (int j = 1; j < 4; ++j) { mine.y = s == j ? S.frx[j] : mine.y; mine.z = s == j ? S.fry[j] : mine.z; }
            mine.w = al ? 1.0f : 0.0f;
            X[p * 4 + q] = mine;
          }
          wave_lds_fence();
          float* P = T + 256;      // [pixel][sample q'][octet q]: behind the 64 sample records
          const unsigned int cell_b = NHWC ? static_cast<unsigned int>(C) * 4u : 4u, row_b = static_cast<unsigned int>(a.W) * cell_b;
          const unsigned int octet_b = NHWC ? 32u * q : 8u * q * plane_bytes;
          constexpr int kGatherUnroll = NHWC ? Cfg::GATHER_UNROLL : 1;
#pragma unroll kGatherUnroll
          for (int qq = 0; qq < 4; ++qq) {      // (not fully unrolled: 8 - 32 operand registers in flight per sample are what the kernel has left; NCHW maps: one)
            const float4v smp = X[p * 4 + qq];
            const int xy = __builtin_bit_cast(int, smp.x);
            const int x0 = static_cast<short>(xy & 0xffff), y0 = xy >> 16;
            const bool al = smp.w != 0.0f;
            float2v w_n, w_s;
            tap_weights(smp.y, smp.z, &w_n, &w_s);
            const bool in_x0 = static_cast<unsigned int>(x0) < static_cast<unsigned int>(a.W), in_x1 = static_cast<unsigned int>(x0 + 1) < static_cast<unsigned int>(a.W);
            const bool in_y0 = static_cast<unsigned int>(y0) < static_cast<unsigned int>(a.H), in_y1 = static_cast<unsigned int>(y0 + 1) < static_cast<unsigned int>(a.H);
            const unsigned int base = static_cast<unsigned int>(y0) * row_b + static_cast<unsigned int>(x0) * cell_b + octet_b;      // (wraps for -1: only used when inside)
            float d = 0.0f;      // this octet's share of the sample: sum over taps of weight x <f2(tap), f1> (a tap's dot first, as the table path has it)
            auto tap = [&](bool inside, unsigned int off, float weight) {
              const unsigned int vo = (al && inside) ? off : kBufferOutOfRange;
              float A[8];
              if (NHWC) {
                const float4v v0 = buffer_f32x4(meas_rsrc, (FULL || 8 * q < C) ? vo : kBufferOutOfRange, 0u);
                const float4v v1 = buffer_f32x4(meas_rsrc, (FULL || 8 * q + 4 < C) ? vo + 16u : kBufferOutOfRange, 0u);
                A[0] = v0.x; A[1] = v0.y; A[2] = v0.z; A[3] = v0.w;
                A[4] = v1.x; A[5] = v1.y; A[6] = v1.z; A[7] = v1.w;
              } else {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                  A[jj] = buffer_f32(meas_rsrc, (FULL || 8 * q + jj < C) ? vo : kBufferOutOfRange, static_cast<unsigned int>(jj) * plane_bytes);
              }
              float t = A[0] * f1v[0];
#pragma unroll
              for (int jj = 1; jj < 8; ++jj) t = fmaf(A[jj], f1v[jj], t);
              d = fmaf(t, weight, d);
            };
            tap(in_x0 && in_y0, base, w_n.x);
            tap(in_x1 && in_y0, base + cell_b, w_n.y);
            tap(in_x0 && in_y1, base + row_b, w_s.x);
            tap(in_x1 && in_y1, base + row_b + cell_b, w_s.y);
            P[(p * 4 + qq) * 4 + q] = d;
          }
          wave_lds_fence();
          {
            const float4v o = *reinterpret_cast<const float4v*>(P + (p * 4 + q) * 4);
            const float total = ((o.x + o.y) + o.z) + o.w;
            const bool al = (S.alive >> s) & 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = (s == j && al) ? acc[j] + total : acc[j];
          }
          wave_lds_fence();
          MFMA_TRACE(tr_look += __builtin_amdgcn_s_memtime() - tr_g0; ++tr_nstrips;)
          continue;
        }
        const float rcp_bw = __builtin_amdgcn_rcpf(static_cast<float>(bw));   // (1 ulp: the quotients below are corrected / rounded with a margin)
        MFMA_TRACE(tr_box += __builtin_amdgcn_s_memtime() - tr_b0;)

        // The lane's cell of successive tiles moves 16 places through the box's row-major index space: (row, column, byte offset) are
        // kept incrementally -- per request one add, one wrap test, the bounds tests (a cell outside the image, i.e. in the zero apron, or
        // beyond the strip is requested out of range: zeros, no traffic).  The column / row steps of 16 places are wave-uniform.
        const int step_r = bw > 16 ? 0 : static_cast<int>(16.0f * rcp_bw + 1e-3f), step_c = 16 - step_r * bw;   // 16 = step_r * bw + step_c, 0 <= step_c < bw
        const int cell_bytes = NHWC ? C * 4 : 4, row_bytes = a.W * cell_bytes;
        const int step_bytes = step_r * row_bytes + step_c * cell_bytes, wrap_bytes = row_bytes - bw * cell_bytes;
        int cur_r = 0, cur_c = 0, cur_left = 0;     // row / column in the box of the next request's cell, cells left in the strip from it
        unsigned int cur_off = 0u;
        auto seek = [&](int base, int n) {   // first tile of a strip: (row, column) of box cell base + p by a float quotient (exact operands
                                             // below 2^24: a map has fewer cells) and one step of correction either way
          const int idx = base + p;
          int r = static_cast<int>(static_cast<float>(idx) * rcp_bw);
          int col = idx - __mul24(r, bw);
          if (col < 0) { col += bw; --r; }
          if (col >= bw) { col -= bw; ++r; }
          cur_r = r; cur_c = col; cur_left = n - p;
          cur_off = static_cast<unsigned int>(__mul24(y_lo + r, row_bytes) + __mul24(x_lo + col, cell_bytes) + (NHWC ? 32 * q : 8 * q * HW * 4));
        };
        float A0[8], A1[8], A2[8], A3[8];      // (A2 / A3: the deeper operand ring, Cfg::NBUF == 4; dead otherwise)
        auto issue = [&](float (&A)[8]) {   // operand request of the next tile of the strip
          bool ok = cur_left > 0 && static_cast<unsigned int>(x_lo + cur_c) < static_cast<unsigned int>(a.W);
          if (!NHWC) ok = ok && static_cast<unsigned int>(y_lo + cur_r) < static_cast<unsigned int>(a.H);   // (NHWC: a row outside the image is out of range by itself)
          const unsigned int vo = ok ? cur_off : kBufferOutOfRange;
          if (Cfg::ABLATE & 1) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) A[jj] = static_cast<float>(vo + jj) * 1e-9f;
          } else if (NHWC) {
            const float4v v0 = buffer_f32x4(meas_rsrc, (FULL || 8 * q < C) ? vo : kBufferOutOfRange, 0u);
            const float4v v1 = buffer_f32x4(meas_rsrc, (FULL || 8 * q + 4 < C) ? vo + 16u : kBufferOutOfRange, 0u);
            A[0] = v0.x; A[1] = v0.y; A[2] = v0.z; A[3] = v0.w;
            A[4] = v1.x; A[5] = v1.y; A[6] = v1.z; A[7] = v1.w;
          } else {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
              A[jj] = buffer_f32(meas_rsrc, (FULL || 8 * q + jj < C) ? vo : kBufferOutOfRange, static_cast<unsigned int>(jj) * plane_bytes);
          }
          cur_left -= 16;
          cur_c += step_c;
          cur_r += step_r;
          cur_off += static_cast<unsigned int>(step_bytes);
          if (cur_c >= bw) { cur_c -= bw; ++cur_r; cur_off += static_cast<unsigned int>(wrap_bytes); }
        };
        auto compute = [&](const float (&A)[8], int t) {   // D[cell][pixel]: this lane gets its own pixel p, cells 4 q .. 4 q + 3 of the tile
          float4v d = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            if (Cfg::ABLATE & 2) d[jj & 3] += A[jj] * f1v[jj];
            else d = __builtin_amdgcn_mfma_f32_16x16x4f32(A[jj], f1v[jj], d, 0, 0, 0);
          }
          *reinterpret_cast<float4v*>(T + p * PITCH + t * 16 + 4 * q) = d;
        };

        auto compute_pair = [&](const float (&X)[8], const float (&Y)[8], int t) {   // tiles t and t + 1
          float4v d0 = {0.0f, 0.0f, 0.0f, 0.0f}, d1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(X[jj], f1v[jj], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Y[jj], f1v[jj], d1, 0, 0, 0);
          }
          *reinterpret_cast<float4v*>(T + p * PITCH + t * 16 + 4 * q) = d0;
          *reinterpret_cast<float4v*>(T + p * PITCH + t * 16 + 16 + 4 * q) = d1;
        };

        for (int base = 0; base < cells; base += CAP) {
          const int n = min(CAP, cells - base);
          const int ntiles = (n + 15) >> 4;
          MFMA_TRACE(const unsigned long long tr_t0 = __builtin_amdgcn_s_memtime();)
          seek(base, n);
          issue(A0);
          issue(A1);
          if (Cfg::NBUF == 4) {
            issue(A2);
            issue(A3);
          }
          // two tiles: separately, or (PAIR) their MFMAs interleaved on two accumulators; the operands of both are requested again behind them
          auto two_tiles = [&](float (&X)[8], float (&Y)[8], int t) {
            if (Cfg::PAIR && t + 1 < ntiles) {
              compute_pair(X, Y, t);
              issue(X);
              issue(Y);
            } else {
              compute(X, t);
              issue(X);
              if (t + 1 < ntiles) {
                compute(Y, t + 1);
                issue(Y);
              }
            }
          };
          for (int t = 0; t < ntiles; t += Cfg::NBUF) {
            two_tiles(A0, A1, t);
            if (Cfg::NBUF == 4 && t + 2 < ntiles) two_tiles(A2, A3, t + 2);
          }
          wave_lds_fence();
          MFMA_TRACE(const unsigned long long tr_t1 = __builtin_amdgcn_s_memtime(); tr_tiles += tr_t1 - tr_t0; tr_ntiles += ntiles; ++tr_nstrips;)

          // ---- interpolation: four table entries per sample ----
          const float* Tp = T + p * PITCH;
          if (cells <= CAP) {
            // the whole box is in the table and every tap of an alive sample lies in the box: index arithmetic only
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j >= jlo && j < jhi) {
                const bool al = (S.alive >> j) & 1u;
                const int x0 = static_cast<short>(S.xy[j] & 0xffff), y0 = S.xy[j] >> 16;
                const int k = al ? __mul24(y0 - y_lo, bw) + (x0 - x_lo) : 0;
                if (Cfg::ABLATE & 4) {
                  acc[j] += Tp[k];
                } else {
                  const float t_nw = Tp[k], t_ne = Tp[k + 1], t_sw = Tp[k + bw], t_se = Tp[k + bw + 1];
                  float2v w_n, w_s;
                  tap_weights(S.frx[j], S.fry[j], &w_n, &w_s);
                  float f = acc[j];
                  f = fmaf(t_nw, w_n.x, f);
                  f = fmaf(t_ne, w_n.y, f);
                  f = fmaf(t_sw, w_s.x, f);
                  f = fmaf(t_se, w_s.y, f);
                  acc[j] = al ? f : acc[j];
                }
              }
          } else {
            // strips: a tap contributes in the strip that holds its cell
            const int last = n - 1;
            const unsigned int un = static_cast<unsigned int>(n);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j >= jlo && j < jhi) {
                const bool al = (S.alive >> j) & 1u;
                const int x0 = static_cast<short>(S.xy[j] & 0xffff), y0 = S.xy[j] >> 16;
                const int k_nw = (al ? __mul24(y0 - y_lo, bw) + (x0 - x_lo) : 0) - base, k_ne = k_nw + 1, k_sw = k_nw + bw, k_se = k_sw + 1;
                const float t_nw = Tp[min(max(k_nw, 0), last)], t_ne = Tp[min(max(k_ne, 0), last)];
                const float t_sw = Tp[min(max(k_sw, 0), last)], t_se = Tp[min(max(k_se, 0), last)];
                float2v w_n, w_s;
                tap_weights(S.frx[j], S.fry[j], &w_n, &w_s);
                float f = acc[j];
                f = fmaf(t_nw, (al && static_cast<unsigned int>(k_nw) < un) ? w_n.x : 0.0f, f);
                f = fmaf(t_ne, (al && static_cast<unsigned int>(k_ne) < un) ? w_n.y : 0.0f, f);
                f = fmaf(t_sw, (al && static_cast<unsigned int>(k_sw) < un) ? w_s.x : 0.0f, f);
                f = fmaf(t_se, (al && static_cast<unsigned int>(k_se) < un) ? w_s.y : 0.0f, f);
                acc[j] = f;
              }
          }
          wave_lds_fence();   // (the next strip / pass overwrites the table)
          MFMA_TRACE(asm volatile("s_nop 0" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3])); tr_look += __builtin_amdgcn_s_memtime() - tr_t1;)
        }
      }
    }

    // sum over frames, then / C, then / M (for power-of-two counts x * 2^-k is x / 2^k exactly: the reference's per-frame / C followed
    // by the mean over frames, bit for bit; otherwise one rounding closer to exact)
    if (live) {
      const __amdgpu_buffer_rsrc_t out_rsrc = map_resource(as_global(a.out) + static_cast<size_t>(b) * a.D * HW, static_cast<unsigned int>(a.D) * plane_bytes);
      const unsigned int vo = (static_cast<unsigned int>(q) * static_cast<unsigned int>(HW) + static_cast<unsigned int>(pix)) * 4u;
      const float Cf = static_cast<float>(C), Mf = static_cast<float>(a.M);
      if (((C & (C - 1)) | (a.M & (a.M - 1))) == 0) {   // (wave-uniform)
        const float rC = 1.0f / Cf, rM = 1.0f / Mf;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (d_block + 4 * j + q < a.D && (quarter < 0 || j == quarter))
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, (acc[j] * rC) * rM), out_rsrc, static_cast<int>(vo),
                                                  static_cast<int>(static_cast<unsigned int>(d_block + 4 * j) * plane_bytes), 0);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (d_block + 4 * j + q < a.D && (quarter < 0 || j == quarter))
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, (acc[j] / Cf) / Mf), out_rsrc, static_cast<int>(vo),
                                                  static_cast<int>(static_cast<unsigned int>(d_block + 4 * j) * plane_bytes), 0);
      }
    }
  }
}

// FULL: exactly 32 channels (the hot-path shape: no per-channel bounds tests); otherwise any C <= 32 (NHWC: a multiple of 4).
// GRID2D: one batch item whose groups divide by 8 (the frame engine's case): blockIdx.y is the chunk -- the dispatcher walks x first, so the far
// chunks, whose boxes hold more cells, start first and the launch's tail is made of the cheap near ones -- and blockIdx.x & 7 the XCD, each of
// which gets a contiguous range of groups; group -> (row, column) by a host-made reciprocal: no integer division in the wave's prologue
// (the general decode below costs ~100 scalar instructions of a wave's ~1 900).
template <class Cfg, bool NHWC, bool FULL, bool GRID2D>
__global__ __launch_bounds__(64 * Cfg::WPB, Cfg::WAVES) void sweep_mfma_kernel(CostVolumeArgs a, unsigned int groups_x_reciprocal) {
  constexpr int GW = Cfg::GW, GH = Cfg::GH, PW = Cfg::PW, CPW = Cfg::CPW, WPB = Cfg::WPB;
  constexpr int WP = PW * CPW;   // planes per wave
  extern __shared__ __attribute__((aligned(16))) float s_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = WPB == 1 ? 0 : __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));      // (wave-uniform: kept in an SGPR)
  float* T = s_lds + wave * Cfg::wave_lds_floats(a.M);                           // this wave's slice: [16 pixels][PITCH]: <f1(pixel), f2(cell)>
  float4v* ktd = reinterpret_cast<float4v*>(T + Cfg::kTableFloats);              // [M][WP]: K t / depth per frame and plane of this wave

  // ---- work item: one group x CPW chunks of 16 planes; XCD k (= blockIdx % 8) gets a contiguous range of image rows, so a
  // measurement footprint is fetched into one L2 ----
  const int groups_x = (a.W + GW - 1) / GW, groups_y = (a.H + GH - 1) / GH;
  const int wchunks = (a.D + WP - 1) / WP;
  int b, item, wchunk, gx, gy;
  if (GRID2D) {
    const int g = static_cast<int>(blockIdx.x & 7) * static_cast<int>(gridDim.x >> 3) + static_cast<int>(blockIdx.x >> 3);
    b = 0;
    wchunk = static_cast<int>(blockIdx.y);
    if (WPB == 4) {      // g = a block of 2 x 2 groups; the reciprocal is that of the blocks per row
      const int blocks_x = groups_x >> 1;
      const int by = static_cast<int>(__umulhi(static_cast<unsigned int>(g), groups_x_reciprocal));
      gy = 2 * by + (wave >> 1);
      gx = 2 * (g - by * blocks_x) + (wave & 1);
    } else {
      const int gg = g * WPB + wave;
      gy = static_cast<int>(__umulhi(static_cast<unsigned int>(gg), groups_x_reciprocal));      // gg / groups_x for gg < 2^16 (checked by the launch)
      gx = gg - gy * groups_x;
    }
    item = (wchunk * static_cast<int>(gridDim.x) + g) * WPB + wave;
  } else {
  const int per_b = groups_y * groups_x * wchunks, total = per_b * a.B;
  const int per_xcd = ((total + WPB - 1) / WPB + 7) / 8;      // workgroups per XCD
  item = (static_cast<int>(blockIdx.x & 7) * per_xcd + static_cast<int>(blockIdx.x >> 3)) * WPB + wave;
  if (item >= total) return;
  b = item / per_b;
  int rem = item - b * per_b;
  if (Cfg::ORDER == 1 && WPB == 1 && a.B == 1 && per_xcd * 8 == total && per_xcd % wchunks == 0) {
    // chunk-major within the XCD's range of groups: the far chunks, whose boxes hold more cells, start first; the launch's tail is made of
    // the cheap near chunks
    const int l = static_cast<int>(blockIdx.x >> 3), xcd = static_cast<int>(blockIdx.x & 7), groups_per_xcd = per_xcd / wchunks;
    wchunk = l / groups_per_xcd;
    const int g = xcd * groups_per_xcd + l % groups_per_xcd;
    gx = g % groups_x;
    gy = g / groups_x;
  } else {
    wchunk = rem % wchunks;
    rem /= wchunks;
    gx = rem % groups_x;
    gy = rem / groups_x;
  }
  }
  if (Cfg::STAGGER >= 2) {   // a start delay by wave slot (64 * STAGGER_UNIT cycles per slot step)
    constexpr int U = Cfg::STAGGER == 2 ? 8 : Cfg::STAGGER == 3 ? 16 : Cfg::STAGGER == 4 ? 32 : 64;
    switch (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 3) {
      case 1: __builtin_amdgcn_s_sleep(U); break;
      case 2: __builtin_amdgcn_s_sleep(2 * U); break;
      case 3: __builtin_amdgcn_s_sleep(3 * U); break;
      default: break;
    }
  }
  if (Cfg::STAGGER == 1) {
    switch (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 3) {   // HW_ID.WAVE_ID: the wave's slot on its SIMD
      case 0: __builtin_amdgcn_s_setprio(3); break;
      case 1: __builtin_amdgcn_s_setprio(2); break;
      case 2: __builtin_amdgcn_s_setprio(1); break;
      default: __builtin_amdgcn_s_setprio(0); break;
    }
  }
  const int d_wave = wchunk * WP;
  MFMA_TRACE(const unsigned long long tr_start = __builtin_amdgcn_s_memtime(), tr_real0 = __builtin_amdgcn_s_memrealtime();)
  MFMA_TRACE(MfmaTraceCounters tr;)
  sweep_mfma_item<Cfg, NHWC, FULL, false>(a, b, gx, gy, d_wave, lane, T, ktd, WP, -1 MFMA_TRACE(, tr));

#ifdef DVMVS_SWEEP_TRACE
  if (lane == 0 && item < kMfmaTraceWaves) {
    unsigned long long* t = g_sweep_mfma_trace + static_cast<size_t>(item) * kMfmaTraceWords;
    t[0] = tr_start; t[1] = __builtin_amdgcn_s_memtime(); t[2] = tr.first - tr_start; t[3] = tr.pos; t[4] = tr.box; t[5] = tr.tiles; t[6] = tr.look;
    t[7] = tr.ntiles | (tr.nstrips << 32); t[8] = __builtin_amdgcn_s_getreg((31 << 11) | 4); t[9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    t[10] = tr_real0; t[11] = __builtin_amdgcn_s_memrealtime(); t[12] = blockIdx.x; t[13] = static_cast<unsigned long long>(wchunk);
  }
#endif
}

// ---- persistent form (round 6): one 16-wave workgroup per CU, every SIMD gets the same mix of work ---------------------------------------
// What round 6's per-SIMD timeline of the kernel above showed (profiles/r06_sweep_mfma_v3_timeline.txt): a launch ends when its busiest SIMD does, a
// SIMD's end time follows the number of 16-cell TILES its waves hold (correlation 0.96 - 0.98; 0.0 - 0.2 with its wave count), and with one-item
// workgroups handed out by the dispatcher that number is the sum of ~5 random items of 4 ... 38 tiles: 22 ... 115 tiles per SIMD where the mean is 62,
// a SIMD without a resident wave for 35 % of the launch on average.  Items are too coarse (5 per SIMD, 4 of them placed at once) for the dispatcher's
// greedy placement to even that out, and finer items cost their fixed overhead again.  So the placement is made here instead:
//   * the grid is one 1024-thread workgroup per CU (the 16 table slices take the CU's LDS: exactly one is resident), wave w = (list s = w & 3, slot
//     k = w >> 2): the hardware places a workgroup's waves round-robin on the four SIMDs, so the four waves of a list share a SIMD;
//   * XCD x (= blockIdx & 7) owns a contiguous range of groups (its measurement footprint stays in one L2) and 4 x workgroups-per-XCD lists; list l gets
//     chunk c of group ((l - c L / NC) mod L) + r L, r = 0, 1, ...: every list holds every chunk index equally often, each from another group, L / NC
//     groups apart -- the far chunks' many tiles and the near chunks' few, and the spatial variation of the footprints, meet in every list.  At 160 x 128
//     (1 280 groups, 4 chunks): five items per list, one of each chunk + one more;
//   * the first four items of a list are its waves' own (slot k takes enumeration index k); further ones are pulled through an LDS counter by
//     whichever of the four finishes first (the wave that had the near chunk).  No global atomics, no cross-workgroup traffic, and which wave computes
//     an item changes nothing in what is computed: bit-identical to the one-item-per-workgroup kernel.
//   * K t / depth of ALL planes and frames is computed once per workgroup (one barrier before the first item), not per item.
// Correctness does not depend on where the hardware puts workgroups or waves; the balance does (observed behaviour, ROCm 7.2).
struct MfmaPersistentPlan {
  unsigned int groups_x_reciprocal;   // g / groups_x == mulhi(g, r) for g < 2^16
  unsigned int chunks_reciprocal;     // n / chunks likewise
  int rows_per_xcd;                   // 2 ceil(groups_y / 16): local group rows of an XCD (some beyond the image when groups_y is no multiple of 16)
  int d_pad;                          // planes rounded up to whole chunks: the frame stride of the shared K t / depth table
  // launch constants of the item enumeration (made on the host: six integer divisions per item were ~240 scalar instructions of a wave's ~900)
  int chunks, lists, g_space, full_rounds, left_items, n_full, n_end;
};

template <class Cfg, bool NHWC, bool FULL>
__global__ __launch_bounds__(1024, 4) void sweep_mfma_persistent_kernel(CostVolumeArgs a, MfmaPersistentPlan plan) {
  constexpr int GW = Cfg::GW, GH = Cfg::GH, PW = Cfg::PW;
  static_assert(Cfg::CPW == 1, "one chunk per item");
  extern __shared__ __attribute__((aligned(16))) float s_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  float* T = s_lds + wave * Cfg::kTableFloats;
  float4v* ktd_all = reinterpret_cast<float4v*>(s_lds + 16 * Cfg::kTableFloats);                    // [M][d_pad]
  int* next_index = reinterpret_cast<int*>(ktd_all + a.M * plan.d_pad);                            // the next entry of the workgroup's queue to hand out

  // ---- K t / depth for every frame and plane (utils.py:59-68: the depth as the reference's python-double expression, then an IEEE fp32 division by
  // the fp32-rounded depth), once per workgroup ----
  {
    gcfloat_p kt_g = as_global(a.kt);
    for (int i = threadIdx.x; i < a.M * plan.d_pad; i += 1024) {
      const int m = i / plan.d_pad, d = i - m * plan.d_pad;
      float4v k = {0.0f, 0.0f, 0.0f, 0.0f};
      if (d < a.D) {
        const float depth = plane_depth(a.inv_depth_base, a.inv_depth_step, d);
        k.x = kt_g[m * 3 + 0] / depth;
        k.y = kt_g[m * 3 + 1] / depth;
        k.z = kt_g[m * 3 + 2] / depth;
      }
      ktd_all[i] = k;
    }
    if (threadIdx.x == 0) next_index[0] = 0;
  }
  __syncthreads();

  // XCD x owns the group rows x, 15 - x, 16 + x, 31 - x, ...: a point-symmetric set (row r and row groups_y - 1 - r meet in the same XCD where
  // groups_y is a multiple of 16), so a footprint size that varies smoothly over the image -- it does, by up to 3 x from one corner to the other -- gives every
  // XCD the same total, and a group and its mirror image, whose costs add up to twice the mean of a linear field, can meet in one list.
  // Its group index space: local row k (even: 16 (k / 2) + x, odd: 16 (k / 2) + 15 - x) x column; a slot whose row lies beyond the image is skipped.
  // Enumeration of list l: first, for every full round r (groups v = l + r lists), the chunks of v -- the far / near ones (c & 3 in {0, 3}) of v itself, the
  // middle ones of v's mirror image; then the chunks of the groups that do not fill a round (160 = 128 + 32 at 160 x 128), dealt one by one over all lists.
  //
  // A wave's first item is its own: enumeration index `slot` of its list.  The rest of the workgroup's four lists is ONE queue, handed out through an LDS
  // counter to whichever wave is free: entry e = (list e & 3, index 4 + e / 4).  The LAST four entries are handed out as quarter items (4 planes of the 16
  // each): what ends a launch is the chain "a wave's own item, then one more" on the SIMDs that got the big ones -- an item is a latency-bound sequence
  // (~7 us + 0.5 us per tile however empty its SIMD is) --, and a quarter is about half as long as the whole.
  // (One loop, one inlined copy of the item; everything the bookkeeping needs is recomputed from the arguments per item -- ~30 scalar instructions -- so
  // that nothing of it is alive across the item, whose registers are the kernel's.)
  for (int handed = -1;;) {
    const int groups_x = (a.W + GW - 1) / GW, groups_y = (a.H + GH - 1) / GH;
    const int chunks = plan.chunks, lists = plan.lists, g_space = plan.g_space, full_rounds = plan.full_rounds, left_items = plan.left_items;
    const int n_full = plan.n_full, n_end = plan.n_end;
    const int xcd = static_cast<int>(blockIdx.x & 7), wg_in_xcd = static_cast<int>(blockIdx.x >> 3);
    const int entries = 4 * (n_end - 4), quartered = Cfg::QUARTER ? min(entries, 4) : 0, whole = entries - quartered;
    int l_, n, quarter = -1;
    if (handed < 0) {
      l_ = 4 * wg_in_xcd + (wave & 3);
      n = wave >> 2;
    } else {
      if (handed >= whole + 4 * quartered) break;
      const int e = handed < whole ? handed : whole + ((handed - whole) >> 2);
      quarter = handed < whole ? -1 : ((handed - whole) & 3);
      l_ = 4 * wg_in_xcd + (e & 3);
      n = 4 + (e >> 2);
    }
    int c, v;
    bool valid = true;
    if (n < n_full) {
      const int r = chunks == 1 ? n : static_cast<int>(__umulhi(static_cast<unsigned int>(n), plan.chunks_reciprocal));      // n / chunks
      c = n - r * chunks;
      v = l_ + r * lists;
    } else {
      const int j = l_ + (n - n_full) * lists;
      valid = j < left_items;
      const int jj = valid ? j : 0;
      const int e = chunks == 1 ? jj : static_cast<int>(__umulhi(static_cast<unsigned int>(jj), plan.chunks_reciprocal));      // j / chunks
      c = jj - e * chunks;
      v = full_rounds * lists + e;
    }
    const int gi = !valid ? 0 : ((c & 3) == 0 || (c & 3) == 3) ? v : g_space - 1 - v;
    const int k = static_cast<int>(__umulhi(static_cast<unsigned int>(gi), plan.groups_x_reciprocal));      // gi / groups_x
    const int gx = gi - k * groups_x;
    const int gy = 16 * (k >> 1) + ((k & 1) ? 15 - xcd : xcd);
    if (valid && gy < groups_y) {
      MFMA_TRACE(const unsigned long long tr_start = __builtin_amdgcn_s_memtime(), tr_real0 = __builtin_amdgcn_s_memrealtime(); MfmaTraceCounters tr;)
      sweep_mfma_item<Cfg, NHWC, FULL, true>(a, 0, gx, gy, c * PW, lane, T, ktd_all + c * PW, plan.d_pad, quarter MFMA_TRACE(, tr));
#ifdef DVMVS_SWEEP_TRACE
      // (one record per item: a quartered item's record is its quarter 0's; quarters 1..3 go to the records behind the last item)
      const int groups = groups_x * groups_y;
      const int item = quarter <= 0 ? c * groups + gy * groups_x + gx : chunks * groups + ((quarter - 1) * 1024 + static_cast<int>(blockIdx.x) * 4 + (l_ & 3));
      if (lane == 0 && item < kMfmaTraceWaves) {
        unsigned long long* t = g_sweep_mfma_trace + static_cast<size_t>(item) * kMfmaTraceWords;
        t[0] = tr_start; t[1] = __builtin_amdgcn_s_memtime(); t[2] = tr.first - tr_start; t[3] = tr.pos; t[4] = tr.box; t[5] = tr.tiles; t[6] = tr.look;
        t[7] = tr.ntiles | (tr.nstrips << 32); t[8] = __builtin_amdgcn_s_getreg((31 << 11) | 4); t[9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        t[10] = tr_real0; t[11] = __builtin_amdgcn_s_memrealtime(); t[12] = blockIdx.x; t[13] = static_cast<unsigned long long>(c); t[14] = static_cast<unsigned long long>(n);
        t[15] = static_cast<unsigned long long>(quarter + 1);
      }
#endif
    }
    int next = 0;
    if (lane == 0) next = atomicAdd(next_index, 1);
    handed = __builtin_amdgcn_readfirstlane(next);
  }
}

// ---- host-side work estimate ---------------------------------------------------------------------------------------------------
// What a launch of the shipped configuration will cost, from the HOST copies of the matrices (no HIP call): the kernel's duration follows the
// number of 16-cell tiles and of passes per wave (section 4.1b of DESIGN.md), and both follow from the sample boxes.  A box is estimated
// from the group's four corner pixels on the first and last plane of a run (a plane-induced homography maps the group to a convex
// quadrilateral, the position is monotone along the epipolar line while Z > 0), on every sixth group column of every sixth group row, staggered.
// stats[0..3]: mean tiles per wave, mean passes per wave, mean strips beyond the first per wave, waves with a corner behind the camera or a
// non-finite position (their boxes are not estimated: counted as the image-sized worst case) as a fraction.
#pragma clang fp contract(off)
template <class Cfg>
void host_mfma_estimate(const float* Hm, const float* kt, int M, int H, int W, int D, double inv_base, double inv_step, double* stats) {
  constexpr int GW = Cfg::GW, GH = Cfg::GH, PW = Cfg::PW, CAP = Cfg::CAP;
  const float Wf = static_cast<float>(W), Hf = static_cast<float>(H), wn = Wf * 0.5f, hn = Hf * 0.5f, Wm1 = static_cast<float>(W - 1), Hm1 = static_cast<float>(H - 1);
  const int groups_x = (W + GW - 1) / GW, groups_y = (H + GH - 1) / GH, chunks = (D + PW - 1) / PW;
  double tiles = 0.0, passes = 0.0, extra_strips = 0.0, wild = 0.0;
  long long waves = 0;
  // K t / depth per frame and plane, once (this function runs once per keyframe on the engine's planning thread: ~25 us)
  float ktd[DVMVS_MAX_MEASUREMENTS][DVMVS_MAX_DEPTH_LEVELS][3];
  for (int d = 0; d < D; ++d) {
    const float depth = static_cast<float>(1.0 / (inv_base + static_cast<double>(d) * inv_step));
    for (int m = 0; m < M; ++m)
      for (int k = 0; k < 3; ++k) ktd[m][d][k] = kt[m * 3 + k] / depth;
  }
  SweepRay rays[4];      // the current group's corner rays for the current frame
  // cells of the box of planes [d0, d1] of frame m for the group whose corner rays are in `rays`; < 0: not estimable
  auto box_cells = [&](int m, int d0, int d1) -> long long {
    float lo_x = 1e30f, hi_x = -1e30f, lo_y = 1e30f, hi_y = -1e30f;
    for (int corner = 0; corner < 8; ++corner) {
      const SweepRay& r = rays[corner & 3];
      const float* k = ktd[m][(corner & 4) ? d1 : d0];
      const float denom = (r.Z0 + k[2]) + 1e-8f;
      const float u = (r.X0 + k[0]) / denom, v = (r.Y0 + k[1]) / denom;
      const float ix = ((((u - wn) / wn) + 1.0f) * 0.5f) * Wm1, iy = ((((v - hn) / hn) + 1.0f) * 0.5f) * Hm1;
      if (!(denom > 1e-6f) || !(ix > -1e6f && ix < 1e6f && iy > -1e6f && iy < 1e6f)) return -1;
      lo_x = fminf(lo_x, ix); hi_x = fmaxf(hi_x, ix); lo_y = fminf(lo_y, iy); hi_y = fmaxf(hi_y, iy);
    }
    if (hi_x <= -1.0f || lo_x >= Wf || hi_y <= -1.0f || lo_y >= Hf) return 0;      // every sample dead: nothing to do
    const float cx0 = floorf(fmaxf(lo_x, -1.0f)), cx1 = floorf(fminf(hi_x, Wf - 1.0f)) + 1.0f;
    const float cy0 = floorf(fmaxf(lo_y, -1.0f)), cy1 = floorf(fminf(hi_y, Hf - 1.0f)) + 1.0f;
    return static_cast<long long>(cx1 - cx0 + 1.0f) * static_cast<long long>(cy1 - cy0 + 1.0f);
  };
  auto account = [&](long long cells, double* t, double* s) {
    if (cells < 0) cells = static_cast<long long>(H) * W / 4;      // (a footprint that cannot be bounded from its corners: a large one)
    *t += static_cast<double>((cells + 15) / 16);
    if (cells > CAP) *s += static_cast<double>((cells + CAP - 1) / CAP - 1);
  };
  constexpr int stride = 6;      // (36 of 1 280 groups at 160 x 128; strides 4 / 6 / 8 pick within 0.3 us of each other over the 285 pairs: profiles/r05_sweep_selection.md)
  for (int gy = 1; gy < groups_y; gy += stride)
    for (int gx = 1 + (stride / 2) * ((gy / stride) % 2); gx < groups_x; gx += stride) {
      const int x0 = gx * GW, y0 = gy * GH;
      const float xs[2] = {static_cast<float>(x0), static_cast<float>(x0 + GW - 1 < W - 1 ? x0 + GW - 1 : W - 1)};
      const float ys[2] = {static_cast<float>(y0), static_cast<float>(y0 + GH - 1 < H - 1 ? y0 + GH - 1 : H - 1)};
      waves += chunks;
      for (int m = 0; m < M; ++m) {
        for (int corner = 0; corner < 4; ++corner) rays[corner] = sweep_ray(Hm + m * 9, xs[corner & 1], ys[corner >> 1]);
        for (int c = 0; c < chunks; ++c) {
          const int d0 = c * PW, d1 = (d0 + PW - 1 < D - 1) ? d0 + PW - 1 : D - 1;
          const long long cells = box_cells(m, d0, d1);
          if (cells == 0) continue;
          if (cells < 0) wild += 1.0 / M;
          if (cells >= 0 && cells <= static_cast<long long>(Cfg::SPLIT) * CAP) {
            account(cells, &tiles, &extra_strips);
            passes += 1.0;
          } else {
            for (int sub = d0; sub <= d1; sub += 4) {
              const long long part = box_cells(m, sub, sub + 3 < d1 ? sub + 3 : d1);
              if (part == 0) continue;
              account(part, &tiles, &extra_strips);
              passes += 1.0;
            }
          }
        }
      }
    }
  const double n = waves > 0 ? static_cast<double>(waves) : 1.0;
  stats[0] = tiles / n; stats[1] = passes / n; stats[2] = extra_strips / n; stats[3] = wild / n;
}
#pragma clang fp contract(fast)

void sweep_mfma_estimate_host(const float* Hm, const float* kt, int M, int H, int W, int D, double inv_base, double inv_step, double* stats);

// ---- launch ------------------------------------------------------------------------------------------------------------
template <class Cfg, bool NHWC, bool FULL>
int launch_sweep_mfma_layout(const CostVolumeArgs& a, hipStream_t stream) {
  constexpr int WPB = Cfg::WPB;
  const long long groups_x = (a.W + Cfg::GW - 1) / Cfg::GW, groups_y = (a.H + Cfg::GH - 1) / Cfg::GH;
  const long long total = groups_y * groups_x * ((a.D + Cfg::PW * Cfg::CPW - 1) / (Cfg::PW * Cfg::CPW)) * a.B;
  if (total > (1LL << 30)) return DVMVS_EUNSUPPORTED;
  const long long groups = groups_x * groups_y, wchunks = (a.D + Cfg::PW * Cfg::CPW - 1) / (Cfg::PW * Cfg::CPW);
  const size_t lds = Cfg::lds_bytes(a.M);
  // 2-D grid (one batch item): x = workgroups of one chunk, XCD-contiguous; y = chunk.  Four-wave workgroups own 2 x 2 groups.
  const bool square = WPB == 4 && groups_x % 2 == 0 && groups_y % 2 == 0;
  const bool grid2d = Cfg::ORDER == 1 && a.B == 1 && groups % (8 * WPB) == 0 && groups < 65536 && wchunks < 65536 && (WPB != 4 || square);
  if (grid2d) {
    const unsigned int per_row = static_cast<unsigned int>(WPB == 4 ? groups_x / 2 : groups_x);
    const unsigned int reciprocal = static_cast<unsigned int>(0xffffffffu / per_row) + 1u;      // g / per_row == mulhi(g, r) for g < 2^16
    hipLaunchKernelGGL((sweep_mfma_kernel<Cfg, NHWC, FULL, true>), dim3(static_cast<unsigned int>(groups / WPB), static_cast<unsigned int>(wchunks)), dim3(64 * WPB),
                       lds, stream, a, reciprocal);
    return launch_status();
  }
  const unsigned int grid = static_cast<unsigned int>(((total + WPB - 1) / WPB + 7) / 8 * 8);
  hipLaunchKernelGGL((sweep_mfma_kernel<Cfg, NHWC, FULL, false>), dim3(grid), dim3(64 * WPB), lds, stream, a, 0u);   // (< 48 KB of LDS: no attribute)
  return launch_status();
}

// The persistent form takes one batch item with at most 64 K groups whose K t / depth table fits beside the 16 table slices, and enough items to
// give every wave of the chip one (else the one-item-per-workgroup kernel above).
template <class Cfg>
bool sweep_mfma_persistent_eligible(const CostVolumeArgs& a) {
  const long long groups_x = (a.W + Cfg::GW - 1) / Cfg::GW, groups_y = (a.H + Cfg::GH - 1) / Cfg::GH, groups = groups_x * groups_y;
  const long long chunks = (a.D + Cfg::PW - 1) / Cfg::PW;
  const size_t lds = sizeof(float) * 16 * Cfg::kTableFloats + sizeof(float) * 4 * static_cast<size_t>(a.M) * chunks * Cfg::PW + 16;
  return Cfg::CPW == 1 && a.B == 1 && groups < 65536 && groups * chunks >= 4096 && lds <= 160 * 1024;
}

template <class Cfg, bool NHWC, bool FULL>
int launch_sweep_mfma_persistent(const CostVolumeArgs& a, hipStream_t stream) {
  static int workgroups[64] = {};      // per device: CUs rounded down to a multiple of 8 (one workgroup per CU); 0 = not asked yet
  int device = 0;
  DVMVS_RETURN_IF_HIP(hipGetDevice(&device));
  const bool tracked = device >= 0 && device < 64;
  int n = tracked ? workgroups[device] : 0;
  const int d_pad = (a.D + Cfg::PW - 1) / Cfg::PW * Cfg::PW;
  const size_t lds = sizeof(float) * 16 * Cfg::kTableFloats + sizeof(float) * 4 * static_cast<size_t>(a.M) * d_pad + 16;
  if (n == 0) {      // (idempotent; racing threads write the same values)
    int cus = 0;
    DVMVS_RETURN_IF_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    n = cus / 8 * 8;
    if (n < 8) return DVMVS_EUNSUPPORTED;
    DVMVS_RETURN_IF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_mfma_persistent_kernel<Cfg, NHWC, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            160 * 1024));
    if (tracked) workgroups[device] = n;
  }
  const int groups_x = (a.W + Cfg::GW - 1) / Cfg::GW, groups_y = (a.H + Cfg::GH - 1) / Cfg::GH;
  MfmaPersistentPlan plan;
  plan.groups_x_reciprocal = static_cast<unsigned int>(0xffffffffu / static_cast<unsigned int>(groups_x)) + 1u;
  plan.rows_per_xcd = 2 * ((groups_y + 15) / 16);
  plan.d_pad = d_pad;
  plan.chunks = d_pad / Cfg::PW;
  plan.chunks_reciprocal = static_cast<unsigned int>(0xffffffffu / static_cast<unsigned int>(plan.chunks)) + 1u;
  plan.lists = 4 * (n / 8);                                   // per XCD
  plan.g_space = plan.rows_per_xcd * groups_x;
  plan.full_rounds = plan.g_space / plan.lists;
  plan.left_items = (plan.g_space - plan.full_rounds * plan.lists) * plan.chunks;
  plan.n_full = plan.chunks * plan.full_rounds;
  plan.n_end = plan.n_full + (plan.left_items + plan.lists - 1) / plan.lists;
  if (plan.g_space >= 65536 || plan.left_items + plan.lists >= 65536 || plan.n_end >= 65536) return DVMVS_EUNSUPPORTED;      // (the reciprocal divisions are exact below 2^16)
  hipLaunchKernelGGL((sweep_mfma_persistent_kernel<Cfg, NHWC, FULL>), dim3(static_cast<unsigned int>(n)), dim3(1024), lds, stream, a, plan);
  return launch_status();
}

template <class Cfg>
int launch_sweep_mfma_persistent_cfg(const CostVolumeArgs& a, hipStream_t stream) {
  if (a.C == kMfmaSweepChannels)
    return a.image2_nhwc ? launch_sweep_mfma_persistent<Cfg, true, true>(a, stream) : launch_sweep_mfma_persistent<Cfg, false, true>(a, stream);
  return a.image2_nhwc ? launch_sweep_mfma_persistent<Cfg, true, false>(a, stream) : launch_sweep_mfma_persistent<Cfg, false, false>(a, stream);
}

template <class Cfg>
int launch_sweep_mfma_cfg(const CostVolumeArgs& a, hipStream_t stream) {
  static_assert(Cfg::lds_bytes(DVMVS_MAX_MEASUREMENTS) <= 48 * 1024, "dynamic LDS beyond the default limit");
  if (a.C == kMfmaSweepChannels)
    return a.image2_nhwc ? launch_sweep_mfma_layout<Cfg, true, true>(a, stream) : launch_sweep_mfma_layout<Cfg, false, true>(a, stream);
  return a.image2_nhwc ? launch_sweep_mfma_layout<Cfg, true, false>(a, stream) : launch_sweep_mfma_layout<Cfg, false, false>(a, stream);
}

// up to 32 channels (channels-last measurement maps: a multiple of 4), maps and one batch item's volume below 2 GiB (32-bit buffer offsets), image sides that fit the
// packed int16 tap coordinates, fewer than 2^24 cells
bool sweep_mfma_supports(const CostVolumeArgs& a) {
  return a.C >= 1 && a.C <= kMfmaSweepChannels && (!a.image2_nhwc || a.C % 4 == 0) && static_cast<long long>(a.C) * a.H * a.W * 4 < (1LL << 31) &&
         static_cast<long long>(a.D) * a.H * a.W * 4 < (1LL << 31) && a.H < 32000 && a.W < 32000 && static_cast<long long>(a.H) * a.W < (1LL << 24);
}

// the shipped configuration (round 6: the persistent form wherever it is eligible; gather passes for 4-plane boxes of more than 256 cells = 16 tiles,
// two samples' loads in flight.  Measured on the sample scene's magnified pairs -- a gather pass is bound by the L1's 64 B / clk, 32 KB of tap lines per
// pass and frame --, thresholds 64 / 96 / 144 / 256: 52.1 / 43.7 / 39.2 / 39.7 us mean over 14 index lines (profiles/r06_sweep_mfma_gather.txt); 256 =
// SPLIT x CAP is the smallest threshold for which whole and quarter items agree on every sample's path without evaluating each other's boxes.
// Interleaved MFMA pairs (PAIR) are worth 1 % and cost the gather build 8 registers it does not have: off)
using MfmaSweepDefault = MfmaSweepConfig<4, 4, 128, 1, 4, 0, 0, 1, 1, 2, 0, 256, 2>;
void sweep_mfma_estimate_host(const float* Hm, const float* kt, int M, int H, int W, int D, double inv_base, double inv_step, double* stats) {
  host_mfma_estimate<MfmaSweepDefault>(Hm, kt, M, H, W, D, inv_base, inv_step, stats);
}
int launch_sweep_mfma(const CostVolumeArgs& a, hipStream_t stream, bool allow_persistent) {
  if (!sweep_mfma_supports(a)) return DVMVS_EUNSUPPORTED;
  if (allow_persistent && sweep_mfma_persistent_eligible<MfmaSweepDefault>(a)) return launch_sweep_mfma_persistent_cfg<MfmaSweepDefault>(a, stream);
  return launch_sweep_mfma_cfg<MfmaSweepDefault>(a, stream);
}

#ifdef DVMVS_SWEEP_TUNING   // tools-only builds: configurations for tools/cv_microbench.py (variants 96 + k)
int launch_sweep_mfma_tuning(int which, const CostVolumeArgs& a, hipStream_t stream) {
  if (!sweep_mfma_supports(a)) return DVMVS_EUNSUPPORTED;
  switch (which) {   // <GW, GH, CAP, CPW, WAVES, ABLATE>
    case 0: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4>>(a, stream);
    case 1: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 2, 3>>(a, stream);
    case 2: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 4, 2>>(a, stream);
    case 3: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 5>>(a, stream);
    case 4: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 2, 4>>(a, stream);
    case 5: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 96, 1, 5>>(a, stream);
    case 6: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 192, 2, 3>>(a, stream);
    case 7: return launch_sweep_mfma_cfg<MfmaSweepConfig<8, 2, 128, 1, 4>>(a, stream);
    case 8: return launch_sweep_mfma_cfg<MfmaSweepConfig<2, 8, 128, 1, 4>>(a, stream);
    case 9: return launch_sweep_mfma_cfg<MfmaSweepConfig<8, 2, 128, 2, 3>>(a, stream);
    case 10: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 4, 3>>(a, stream);
    case 11: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 3>>(a, stream);
    case 12: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 1, 0>>(a, stream);   // staggered priorities
    case 13: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 0, 1>>(a, stream);   // far chunks first
    case 14: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 1, 1>>(a, stream);   // both
    case 15: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 2, 3, 0, 1, 1>>(a, stream);
    case 21: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 2, 1>>(a, stream);   // start delays by wave slot
    case 22: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 3, 1>>(a, stream);
    case 23: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 4, 1>>(a, stream);
    case 24: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 5, 1>>(a, stream);
    case 25: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 2, 3, 0, 0, 1>>(a, stream);
    case 26: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 3, 0, 0, 1>>(a, stream);
    case 27: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 2, 0, 0, 1>>(a, stream);
    case 28: return launch_sweep_mfma_cfg<MfmaSweepConfig<8, 2, 128, 1, 4, 0, 0, 1>>(a, stream);
    case 29: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 96, 1, 5, 0, 0, 1>>(a, stream);    // 20 waves per CU: 96 registers, 7.9 KB of table
    case 30: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 112, 1, 5, 0, 0, 1>>(a, stream);
    case 31: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 5, 0, 0, 1>>(a, stream);
    // ablations of configuration 0 (wrong results; where does the time go)
    case 16: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 1, 0, 1>>(a, stream);   // no operand loads
    case 17: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 2, 0, 1>>(a, stream);   // no MFMAs
    case 18: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 4, 0, 1>>(a, stream);   // no interpolation
    case 19: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 3, 0, 1>>(a, stream);   // neither loads nor MFMAs
    case 20: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 7, 0, 1>>(a, stream);   // positions + boxes + table writes only
    // round 6 (cost-volume variants 224 + k - 32): workgroup shape (WPB waves: one per SIMD with 4), operand ring depth, interleaved MFMA pairs
    //                                                       <GW, GH, CAP, CPW, WAVES, ABLATE, STAGGER, ORDER, WPB, NBUF, PAIR>
    case 32: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 0, 1, 4, 2, 0>>(a, stream);   // four-wave workgroups only
    case 33: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 0, 1, 4, 2, 1>>(a, stream);   // + interleaved pairs
    case 34: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 0, 1, 4, 4, 1>>(a, stream);   // + four tiles in flight
    case 35: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 0, 1, 1, 2, 1>>(a, stream);   // one-wave workgroups, interleaved pairs
    case 36: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 96, 1, 5, 0, 0, 1, 4, 2, 1>>(a, stream);    // 5 workgroups per CU: every wave of the launch resident
    case 37: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 0, 1, 2, 2, 1>>(a, stream);   // two-wave workgroups
    case 38: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 112, 1, 5, 0, 0, 1, 4, 2, 1>>(a, stream);
    case 39: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 0, 1, 4, 4, 0>>(a, stream);
    case 40: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 3, 0, 0, 1, 4, 4, 1>>(a, stream);   // 3 workgroups per CU, 168 registers allowed
    case 42: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 96, 1, 5, 0, 0, 1, 1, 2, 0>>(a, stream);    // = round 5's case 29 (20 one-wave workgroups per CU), its own id
    // persistent form (one 16-wave workgroup per CU, balanced lists): configurations as above; the general kernel where it is not eligible
    case 43: return sweep_mfma_persistent_eligible<MfmaSweepConfig<4, 4, 128, 1, 4>>(a) ? launch_sweep_mfma_persistent_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 0, 1, 1, 2, 0>>(a, stream) : DVMVS_EUNSUPPORTED;
    case 44: return sweep_mfma_persistent_eligible<MfmaSweepConfig<4, 4, 128, 1, 4>>(a) ? launch_sweep_mfma_persistent_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 0, 1, 1, 2, 1>>(a, stream) : DVMVS_EUNSUPPORTED;
    // gather passes (thresholds in cells of a 4-plane box), persistent form
    case 54: return sweep_mfma_persistent_eligible<MfmaSweepConfig<4, 4, 128, 1, 4>>(a) ? launch_sweep_mfma_persistent_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 0, 1, 1, 2, 0, 256, 2, 0>>(a, stream) : DVMVS_EUNSUPPORTED;      // no quarter items
    case 55: return sweep_mfma_persistent_eligible<MfmaSweepConfig<4, 4, 128, 1, 4>>(a) ? launch_sweep_mfma_persistent_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 0, 1, 1, 2, 0, 256, 2, 1>>(a, stream) : DVMVS_EUNSUPPORTED;      // = shipped
    case 56: return sweep_mfma_persistent_eligible<MfmaSweepConfig<4, 4, 128, 1, 4>>(a) ? launch_sweep_mfma_persistent_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 0, 1, 1, 2, 0, 0, 1, 1>>(a, stream) : DVMVS_EUNSUPPORTED;        // no gather passes
    case 48: return sweep_mfma_persistent_eligible<MfmaSweepConfig<4, 4, 128, 1, 4>>(a) ? launch_sweep_mfma_persistent_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, 0, 0, 1, 1, 2, 1, 256>>(a, stream) : DVMVS_EUNSUPPORTED;
    default: return DVMVS_EINVAL;
  }
}
#else
int launch_sweep_mfma_tuning(int, const CostVolumeArgs&, hipStream_t) { return DVMVS_EINVAL; }
#endif

}  // namespace dvmvs

#ifdef DVMVS_SWEEP_TRACE
extern "C" int dvmvs_debug_sweep_mfma_trace(unsigned long long* host, int waves) {
  if (waves > dvmvs::kMfmaTraceWaves) waves = dvmvs::kMfmaTraceWaves;
  return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(dvmvs::g_sweep_mfma_trace), sizeof(unsigned long long) * dvmvs::kMfmaTraceWords * waves));
}
#endif
