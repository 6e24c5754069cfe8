// LDS-staged plane sweep (dot-product mode) for gfx950: the fused warp + correlation kernel of the hot path.
// Semantics: /root/reference/dvmvs/utils.py:45-107; CPU restatement: oracle/dvmvs_oracle.py.
//
// A workgroup owns a TW x TH tile of reference pixels and DP consecutive sweep planes, one thread per pixel.  For one
// measurement frame the samples of the tile over a run of planes ("segment") lie inside the bounding box of 8 points
// (4 tile corners x first / last plane of the run): a plane-induced homography maps the tile to a convex quadrilateral
// and the position moves monotonically along the epipolar line with inverse depth while Z stays positive.  That box,
// with a zero apron that implements grid_sample's zeros padding, is copied once from the measurement map into LDS as
// channel-interleaved records of CCH floats (+4 floats of padding: a record stride of an odd number of 16-byte slots
// keeps the 16 lanes of a ds_read_b128 service group on different banks).
//
// What is different from a generic "gather" formulation, and why (MI355X_MICROARCH.md, LDS section):
//   * the binding resource is the LDS read pipe (4 taps x C channels x 4 B per sample = 1.3 GB per frame against
//     ~150 TB/s of ds_read_b128 bandwidth), so everything else is arranged to stay out of its way;
//   * the box is found by every wave on its own (8 lanes evaluate the corners, three xor-shuffles reduce, readfirstlane
//     makes it scalar): no LDS round trip and no barrier per segment attempt, and the segment length adapts per tile
//     (DP planes, halved down to MINSEG while the box does not fit in CAP records);
//   * per sample the position keeps the reference's fp32 rounding exactly (see SweepScale) but without the range-scaling
//     half of IEEE division, clamping replaces per-tap bounds tests (the apron supplies the zeros), and the tap
//     accumulation is "dot per tap" in packed FMAs (2 v_pk_fma_f32 per channel quad and tap, 4 more per sample) instead of
//     "interpolate, then multiply" (5 scalar FMAs per channel);
//   * workgroups are numbered so that the 8 plane chunks of a tile AND neighbouring tiles land on the same XCD
//     (blockIdx % 8 selects the XCD): a measurement footprint is then fetched into one L2 instead of up to eight;
//   * segments that cannot be staged even at MINSEG planes (magnified / behind-camera footprints) are queued per
//     workgroup ("spill group") and finished by sweep_spill_kernel, which walks whole groups in a fixed order with plain
//     read-modify-writes: the volume is bit-reproducible run to run for any M.
#include "plane_sweep.h"
#include "sweep_sample.h"

namespace dvmvs {


// ---- wave-uniform sample box -------------------------------------------------------------------------------------------
struct SampleBox {
  int x_lo, y_lo, RW, RH;
  int pitch;   // records per staged row: >= RW, congruent to 0 or +-1 mod 16 (see sweep_pitch_residue)
  int state;   // 1: stage through LDS; 2: entirely outside the image (zeros); 0: does not fit / not well defined
};

// LDS bank conflicts of the tap reads (tools/lds_conflict_sim.py is the CPU model of what follows; numbers from it and from
// SQ_LDS_BANK_CONFLICT agree).  A wave64 ds_read_b128 is served in four groups of 16 lanes ({0-3,12-15,20-27},
// {4-11,16-19,28-31}, + 32), one LDS cycle per group when the 16 addresses fall on 16 different 16-byte slots
// (slot = byte address / 16 mod 16), one more cycle per extra distinct address on a slot.  With an odd number of slots per
// record, slot is a permutation of (record index mod 16), so a group is conflict-free iff its 16 record indices
// rx + pitch * ry are distinct mod 16 (identical records broadcast).  Two things make that the common case:
//   * sweep_lane_pixel: lanes are mapped to pixels so that each service group owns 16 CONSECUTIVE pixels of a tile row
//     (the natural mapping spreads a group over 28 pixels, and any scale != 1 then wraps mod 16);
//   * the row pitch of the staged box is chosen per box, congruent to 0, +1 or -1 (mod 16), from the direction in which
//     consecutive pixels move through the measurement image: pitch = 0 makes the slot a function of the column only
//     (right when every pixel step advances a column), pitch = +-1 makes column + row monotone along the run (right when
//     rows change while a column repeats, i.e. under in-plane rotation with scale < 1).
// Measured on the sample scene's keyframe pairs: 6.3-8.5 LDS cycles per ds_read_b128 -> 4.0-5.5 (4 = conflict-free).
__device__ inline int sweep_lane_pixel(int lane32) {
  // lanes {0-3,12-15,20-27} -> pixels 0-15, lanes {4-11,16-19,28-31} -> pixels 16-31
  return lane32 < 4 ? lane32 : lane32 < 12 ? lane32 + 12 : lane32 < 16 ? lane32 - 8 : lane32 < 20 ? lane32 + 8 : lane32 < 28 ? lane32 - 12 : lane32;
}

__host__ __device__ inline int sweep_pitch_residue(float ax, float ay) {
  const float aax = fabsf(ax), aay = fabsf(ay);
  const float c = ceilf(15.0f * aax), r = ceilf(15.0f * aay);
  const float cost0 = 15.0f * aay * fmaxf(0.0f, 1.0f - aax) + fmaxf(0.0f, c - 15.0f);   // rows change while a column repeats
  const float cost1 = fmaxf(0.0f, c + r - 15.0f);                                         // column + row wraps past 16
  if (!(cost1 < cost0)) return 0;
  return ((ax < 0.0f) != (ay < 0.0f)) ? 15 : 1;
}

// (the run plan that uses the box geometry above follows the tap helpers: plan_runs)


// ---- tap accumulation of one plane in one staged channel pass ------------------------------------------------------------
// 4 taps x QPR 16-byte reads; "dot per tap" over the pass's channels, two channels per v_pk_fma_f32; then the four bilinear
// weights (register pairs {nw, ne}, {sw, se}, either half broadcast through op_sel).  (Deeper software pipelining across
// planes was measured: no gain -- the launch is bound by the workgroup's chain of staging round trips, not by LDS latency.)

// LEAN: the record's quads one after the other (four 16-byte reads in flight instead of eight: 16 registers less), for the
// configurations that run five waves per SIMD on 96 registers.
template <int QPR, int REC, bool LEAN = false>
__device__ inline void tap_plane(const char* tile_bytes, int row_bytes, int addr, float2v frac, const float2v* rv, float2v* acc) {
  const char* row0 = tile_bytes + addr;
  const char* row1 = row0 + row_bytes;
  float2v t_nw = {0.0f, 0.0f}, t_ne = {0.0f, 0.0f}, t_sw = {0.0f, 0.0f}, t_se = {0.0f, 0.0f};
#pragma unroll
  for (int q = 0; q < QPR; ++q) {
    const float4v nw = *reinterpret_cast<const float4v*>(row0 + q * 16);
    const float4v ne = *reinterpret_cast<const float4v*>(row0 + REC * 4 + q * 16);
    const float4v sw = *reinterpret_cast<const float4v*>(row1 + q * 16);
    const float4v se = *reinterpret_cast<const float4v*>(row1 + REC * 4 + q * 16);
    t_nw = fma2(rv[q * 2], nw.lo, t_nw); t_nw = fma2(rv[q * 2 + 1], nw.hi, t_nw);
    t_ne = fma2(rv[q * 2], ne.lo, t_ne); t_ne = fma2(rv[q * 2 + 1], ne.hi, t_ne);
    t_sw = fma2(rv[q * 2], sw.lo, t_sw); t_sw = fma2(rv[q * 2 + 1], sw.hi, t_sw);
    t_se = fma2(rv[q * 2], se.lo, t_se); t_se = fma2(rv[q * 2 + 1], se.hi, t_se);
    if (LEAN) __builtin_amdgcn_sched_barrier(0);   // the next quad's reads are not hoisted above these FMAs
  }
  asm volatile("" : "+v"(frac));   // opaque: keeps the four products out of registers between passes (they are loop-invariant
                                    // and would be hoisted right back into the 16 VGPRs this formulation saves)
  float2v w_n, w_s;
  tap_weights(frac.x, frac.y, &w_n, &w_s);
  float2v f = *acc;
  f = fma2(t_nw, __builtin_shufflevector(w_n, w_n, 0, 0), f);   // either half broadcast through op_sel
  f = fma2(t_ne, __builtin_shufflevector(w_n, w_n, 1, 1), f);
  f = fma2(t_sw, __builtin_shufflevector(w_s, w_s, 0, 0), f);
  f = fma2(t_se, __builtin_shufflevector(w_s, w_s, 1, 1), f);
  *acc = f;
}

// ---- run plan ------------------------------------------------------------------------------------------------------------
// A "run" is a stretch of consecutive planes of one measurement frame whose sample footprint is handled as one unit: staged
// through LDS as one box (state 1), skipped because the footprint lies entirely outside the image (state 2), or handed to the
// gather path because even MINSEG planes do not fit (state 0).  The plan of a workgroup -- every run of every frame of its
// (tile, chunk) -- is made up front by wave 0 and left in LDS, so that the execution loop below knows the NEXT run's box while
// the current one is being tapped and can have its staging loads in flight.
//
// One evaluation serves TWO measurement frames (the common M = 2 needs a single one per workgroup): lane = corner (bits 0-2:
// right edge, bottom edge, last plane of the run) + 8 * candidate run length (bits 3-4: len0, then halved, rounding up, not
// below MINSEG, once per candidate) + 32 * frame of the pair.  Three xor-shuffles leave each candidate's extrema in its eight
// lanes, every lane derives its candidate's box, one ballot per evaluation picks, per frame, the longest run that can be staged
// (or lies entirely outside the image), and the picked lane writes its own values as the run's entry -- no LDS round trip, no
// retry loop (1.3 us per evaluation in the s_memtime timeline of round 2, two to six of them per workgroup then).
constexpr int kRunWords = 8;   // [0] m | seg_lo << 8 | seg_len << 16 | state << 24, [1] x_lo, [2] y_lo, [3] RW | RH << 16, [4] pitch

template <int DP, int MINSEG>
__host__ __device__ constexpr int max_runs() { return DVMVS_MAX_MEASUREMENTS * ((DP + MINSEG - 1) / MINSEG); }

template <int TW, int TH, int DP, int CAP, int MINSEG>
__device__ inline int plan_runs(const CostVolumeArgs& a, const float* s_H, const float4v* s_ktd, int tile_x, int tile_y, int planes,
                                const SweepScale& sc, int lane, int* s_runs) {
  int n_runs = 0;
  const int half = lane >> 5, candidate = (lane >> 3) & 3;
  const int x_first = tile_x * TW, x_last = min(tile_x * TW + TW - 1, a.W - 1);
  const int cx = (lane & 1) ? x_last : x_first;
  const int cy = (lane & 2) ? min(tile_y * TH + TH - 1, a.H - 1) : tile_y * TH;
  const float edge = static_cast<float>(max(1, x_last - x_first));
  const float Wf = sc.Wf, Hf = sc.Hf;
  for (int m0 = 0; m0 < a.M; m0 += 2) {
    const int m = min(m0 + half, a.M - 1);
    const SweepRay ray = sweep_ray(s_H + m * 9, static_cast<float>(cx), static_cast<float>(cy));
    int lo_a = 0, lo_b = (m0 + 1 < a.M) ? 0 : planes;   // wave-uniform progress of the two frames
    int hint_a = DP, hint_b = DP;                        // planes per run that fitted last time: parallax per plane is uniform
    while (lo_a < planes || lo_b < planes) {
      const int j_lo = half ? lo_b : lo_a;
      const int len0 = max(1, min(planes - j_lo, half ? hint_b : hint_a));   // (a finished frame evaluates a dummy run)
      int len = len0;
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < candidate) len = max((len + 1) >> 1, MINSEG);
      len = min(len, len0);   // (len0 itself may be below MINSEG at the end of a chunk)
      const float4v k = s_ktd[m * DP + min((lane & 4) ? j_lo + len - 1 : j_lo, DP - 1)];
      // un-clamped position (the box test has to see how far outside the image the corner is)
      float ux, uy, denom;
      sweep_position_exact(ray, k.x, k.y, k.z, sc, &ux, &uy, &denom);
      // direction of a one-pixel step along the tile's top edge (the first two lanes of a frame's half hold its ends on plane j_lo)
      const float step_x = (__shfl(ux, (lane & 32) + 1) - __shfl(ux, lane & 32)) / edge;
      const float step_y = (__shfl(uy, (lane & 32) + 1) - __shfl(uy, lane & 32)) / edge;
      float lo_x = ux, hi_x = ux, lo_y = uy, hi_y = uy;
#pragma unroll
      for (int off = 1; off < 8; off <<= 1) {
        lo_x = fminf(lo_x, __shfl_xor(lo_x, off));
        hi_x = fmaxf(hi_x, __shfl_xor(hi_x, off));
        lo_y = fminf(lo_y, __shfl_xor(lo_y, off));
        hi_y = fmaxf(hi_y, __shfl_xor(hi_y, off));
      }
      // NaN-safe: v_min/v_max drop NaNs, so test the corner values themselves as well
      const bool corner_ok = (ux > -1e6f) && (ux < 1e6f) && (uy > -1e6f) && (uy < 1e6f) && (denom > 1e-6f);
      const bool finite = ((__ballot(corner_ok) >> (lane & 56)) & 0xffull) == 0xffull;   // all eight corners of this candidate
      // 0.05 px of slack for round-off between the corner samples and interior pixels
      const bool outside = (hi_x + 0.05f <= -1.0f) || (lo_x - 0.05f >= Wf) || (hi_y + 0.05f <= -1.0f) || (lo_y - 0.05f >= Hf);
      const int x_lo = max(-1, static_cast<int>(floorf(fmaxf(lo_x - 0.05f, -1.0f))));
      const int y_lo = max(-1, static_cast<int>(floorf(fmaxf(lo_y - 0.05f, -1.0f))));
      const int x_hi = min(a.W, static_cast<int>(floorf(fminf(hi_x + 0.05f, Wf)))) + 1;
      const int y_hi = min(a.H, static_cast<int>(floorf(fminf(hi_y + 0.05f, Hf)))) + 1;
      const int RW = x_hi - x_lo + 1, RH = y_hi - y_lo + 1;
      const int pitch = RW + ((sweep_pitch_residue(step_x, step_y) - RW) & 15);
      const int state = !finite ? 0 : outside ? 2 : (pitch * RH <= CAP) ? 1 : 0;
      // per frame: the first candidate that needs no further halving -- stageable, outside, or already at the minimum run length
      // (the last candidate always is)
      const unsigned long long settled = __ballot(state != 0 || len <= MINSEG);
      const int pick_a = __builtin_ctz(static_cast<unsigned int>(settled) | 0x80000000u);
      const int pick_b = 32 + __builtin_ctz(static_cast<unsigned int>(settled >> 32) | 0x80000000u);
      const bool act_a = lo_a < planes, act_b = lo_b < planes;
      if ((act_a && lane == pick_a) || (act_b && lane == pick_b)) {
        int* run = s_runs + (n_runs + ((half && act_a) ? 1 : 0)) * kRunWords;
        const bool staged = state == 1;
        run[0] = m | (j_lo << 8) | (len << 16) | (state << 24);
        run[1] = staged ? x_lo : 0;
        run[2] = staged ? y_lo : 0;
        run[3] = staged ? (RW | (RH << 16)) : 0;
        run[4] = staged ? pitch : 0;
      }
      const int len_a = __builtin_amdgcn_readlane(len, pick_a), len_b = __builtin_amdgcn_readlane(len, pick_b);
      if (act_a) { lo_a += len_a; hint_a = max(len_a, MINSEG); ++n_runs; }
      if (act_b) { lo_b += len_b; hint_b = max(len_b, MINSEG); ++n_runs; }
    }
  }
  return n_runs;
}

// ---- spill groups ------------------------------------------------------------------------------------------------------
// Spill workspace words: [0] = number of registered groups, [1] = finished workgroups of the second pass, [2..3] unused,
// then `groups` group ids, then one slot per workgroup of (1 + M * DP) words: item count and items.  An item packs
// (m, seg_lo, seg_len); a workgroup queues its items in plan order.
// The header must be zero when a call starts; the second pass restores that (its last workgroup to finish clears words 0 and
// 1), so the caller zero-fills a workspace once, when it allocates it, and never again.
constexpr int kSpillHeaderWords = 4;

// Work items (see host_build_work_list below): a launch has at most twice as many as there are (tile, chunk) pairs -- the static
// ones plus the pieces cut off long ones --, and the spill workspace has one slot per possible item.
constexpr int kWorkListHeaderWords = 2;
__host__ __device__ inline size_t work_list_capacity_items(int B, int H, int W, int D, int TW, int TH, int DP) {
  const size_t tiles = static_cast<size_t>((W + TW - 1) / TW) * ((H + TH - 1) / TH);
  const size_t total = tiles * ((D + DP - 1) / DP) * B;
  return 2 * ((total + 7) / 8 * 8);
}

__host__ __device__ inline int spill_slot_words(int M, int DP) { return 1 + M * DP; }
__device__ inline unsigned int spill_pack(int m, int seg_lo, int seg_len) {
  return (static_cast<unsigned int>(m) << 16) | (static_cast<unsigned int>(seg_lo) << 8) | static_cast<unsigned int>(seg_len);
}

// ---- optional timeline instrumentation (tools/sweep_trace.py; built only by `make trace`) -------------------------------------
#ifdef DVMVS_SWEEP_TRACE
constexpr int kTraceWords = 16, kTraceGroups = 8192;
__device__ unsigned long long g_sweep_trace[kTraceGroups * kTraceWords];
#define SWEEP_TRACE(...) __VA_ARGS__
#else
#define SWEEP_TRACE(...)
#endif

// ---- the kernel ----------------------------------------------------------------------------------------------------------
// PRE: staging pieces of the NEXT channel pass (or of the next run's first pass) whose loads are issued before the taps of the
// current one and held in registers until the LDS tile is free again -- the global-load round trip of a pass (0.65 us of the
// workgroup's dependency chain per pass in round 2's timeline, eight passes per workgroup) then overlaps the LDS-bound tap phase.
// Pieces beyond PRE (boxes larger than PRE * NT records) are loaded after the taps as before.
// PSPLIT: thread groups per tile.  With PSPLIT = 2 a workgroup is 512 threads: both halves own the tile's 256 pixels, the first
// the lower half of the chunk's planes, the second the upper half, and they share ONE staged box.  The tap phase -- the longest
// part of a workgroup's dependency chain (eight passes of ~2 us each at 256 threads) -- is then spread over twice as many waves,
// and a thread carries half the per-plane state (<= 128 registers: 4 waves / SIMD, two 8-wave workgroups per CU).
template <int TW_, int TH_, int DP_, int CCH_, int CAP_, int MINSEG_, int WAVES_ = 3, int ORDER_ = 2, int PRE_ = 2, bool PLAN0_ = true,
          bool FASTFULL_ = false, int PSPLIT_ = 1, bool LEAN_ = false>
struct SweepConfig {
  static constexpr int TW = TW_, TH = TH_, DP = DP_, CCH = CCH_, CAP = CAP_, MINSEG = MINSEG_;
  static constexpr int WAVES = WAVES_;   // waves per SIMD the register allocation is held to
  static constexpr int ORDER = ORDER_;    // workgroup numbering, see decode_work
  static constexpr int PRE = PRE_;       // prefetched staging pieces per thread (NCHW; x2 for channels-last quads), 0 = none
  static constexpr bool PLAN0 = PLAN0_;  // the run plan is made by wave 0 only (the other waves wait at the barrier)
  static constexpr bool FASTFULL = FASTFULL_;   // extra straight-line tap block for runs that cover the whole chunk (measured: 14 more
                                                // registers, 0.5 us slower on the mean: off)
  static constexpr int PSPLIT = PSPLIT_;
  static constexpr bool LEAN = LEAN_;    // register diet for 5 waves / SIMD: quads tapped one after the other, reference features loaded per pass
  static constexpr int NPIX = TW * TH;                                 // pixels of a tile = threads of one plane group
  static constexpr int NT = NPIX * PSPLIT;
  static constexpr int DPT = DP / PSPLIT;                              // planes per thread
  static constexpr int REC = CCH + 4;                                  // floats per LDS record
  static constexpr size_t kLdsBytes = sizeof(float) * static_cast<size_t>(REC) * CAP;
  static_assert(CCH % 4 == 0 && ((REC / 4) % 2) == 1, "record stride must be an odd number of 16-byte slots");
  static_assert(NPIX % 64 == 0 && MINSEG >= 1 && MINSEG <= DP && DP <= 32 && DP % PSPLIT == 0 && NT <= 1024, "workgroup shape");
};

// Decodes the linear workgroup number into a (batch, tile, chunk) work item; `group` is the item's number in (batch, tile, chunk)
// order (it indexes the spill slots).
//   ORDER 0: linear.
//   ORDER >= 1, XCD-aware: XCD = blockIdx % 8 gets a contiguous range of items (tile-major), so the 8 plane chunks of a tile and its
//     row neighbours are fetched into ONE L2.  Within the range any permutation keeps that property; what it decides is which work
//     items share a CU: the dispatcher deals an XCD's workgroups round its 32 CUs (number k lands with k + 32 and k + 64, all
//     resident from the first microsecond), and a workgroup's cost is set by its chunk -- far chunks tap all four channel passes of
//     both frames (the LDS-bound part), near chunks often see their footprint leave the image and finish in a third of the time.
//   ORDER 1: tile-major within the XCD (round 2): with 8 chunks per tile, k, k + 32 and k + 64 are the SAME chunk of three tiles, so
//     a quarter of the CUs get three far chunks each and decide the launch time while others idle (timeline: 33 us span, 22 us mean).
//   ORDER 2: chunk-major within the XCD, far chunks first: every CU gets a far, a middle and (half of them) a near chunk.
//   ORDER 3: tile-major with the chunk rotated by 3 per dispatch round (k, k + 32, k + 64 get chunks c, c + 3, c + 6).
struct SweepWork {
  int b, tile, chunk, group;
  bool valid;
};

template <int ORDER>
__host__ __device__ inline SweepWork decode_work(int block, int tiles, int chunks, int B) {
  SweepWork w;
  const int per_b = tiles * chunks, total = per_b * B;
  int v = block;
  if (ORDER >= 1) {
    const int per_xcd = (total + 7) / 8;
    int l = block >> 3;   // position within the XCD's range
    if (ORDER == 2 && per_xcd % chunks == 0) {
      const int rows = per_xcd / chunks;                      // tiles in this XCD's range
      l = (l % rows) * chunks + (chunks - 1 - l / rows);      // chunk index l / rows counts from the far end
    } else if (ORDER == 3 && per_xcd % chunks == 0) {
      l = (l / chunks) * chunks + (l % chunks + 3 * (l / 32)) % chunks;
    }
    v = (block & 7) * per_xcd + l;
  }
  w.valid = v < total;
  w.group = v;
  w.b = v / per_b;
  const int rem = v - w.b * per_b;
  w.tile = rem / chunks;
  w.chunk = chunks - 1 - (rem - w.tile * chunks);
  return w;
}

// GATHER: there is no second pass (no spill workspace, or the single-pass variants 4 / 5 that the host's plan picks when it expects
// nothing to be queued): runs of planes that cannot be staged are gathered by the same workgroup behind its output write -- slow, but
// then never taken.  A cold loop at the end of the kernel: the staged path is the two-pass kernel's, instruction for instruction.
template <class Cfg, bool NHWC, bool GATHER>
__global__ __launch_bounds__(Cfg::NT, Cfg::WAVES) void sweep_tiled_kernel(CostVolumeArgs a) {
  constexpr int TW = Cfg::TW, TH = Cfg::TH, DP = Cfg::DP, CCH = Cfg::CCH, CAP = Cfg::CAP, NT = Cfg::NT, REC = Cfg::REC;
  constexpr int NPIX = Cfg::NPIX, DPT = Cfg::DPT;
  constexpr int QPR = CCH / 4;   // 16-byte quads per record
  constexpr int kMaxRuns = max_runs<DP, Cfg::MINSEG>();
  extern __shared__ __attribute__((aligned(16))) float s_tile[];   // [CAP][REC]
  __shared__ float s_H[DVMVS_MAX_MEASUREMENTS * 9];
  __shared__ float4v s_ktd[DVMVS_MAX_MEASUREMENTS * DP];
  __shared__ int s_runs[kMaxRuns * kRunWords];
  __shared__ int s_n_runs;

  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const int chunks = (a.D + DP - 1) / DP;
  // the work item: from the caller's work list (plane sub-ranges cut on the host so that no workgroup has a long chain of runs,
  // see host_build_work_list), or the static (tile, chunk) numbering
  SweepWork work;
  int d_block, planes;
  if (a.items != nullptr) {
    const guint_p items = as_global(const_cast<unsigned int*>(a.items));
    if (blockIdx.x >= items[0]) return;
    const unsigned int w0 = items[kWorkListHeaderWords + 2 * blockIdx.x], w1 = items[kWorkListHeaderWords + 2 * blockIdx.x + 1];
    work.b = static_cast<int>(w0 >> 16);
    work.tile = static_cast<int>(w0 & 0xffffu);
    work.group = static_cast<int>(blockIdx.x);
    work.chunk = 0;
    d_block = static_cast<int>(w1 & 0xffffu);
    planes = static_cast<int>(w1 >> 16);
    if (planes == 0) return;
  } else {
    work = decode_work<Cfg::ORDER>(blockIdx.x, tiles_x * tiles_y, chunks, a.B);
    if (!work.valid) return;
    d_block = work.chunk * DP;
    planes = min(DP, a.D - d_block);
  }
  const int b = work.b;
  const int tile_y = work.tile / tiles_x, tile_x = work.tile - tile_y * tiles_x;
  const int tid = threadIdx.x, lane = tid & 63;
  SWEEP_TRACE(unsigned long long tr_stage = 0, tr_taps = 0, tr_passes = 0, tr_records = 0, tr_switch = 0, tr_setup = 0;)
  SWEEP_TRACE(const unsigned long long tr_start = __builtin_amdgcn_s_memtime(); const unsigned long long tr_real0 = __builtin_amdgcn_s_memrealtime();)

  // ---- per-workgroup tables (the caller's Hm = K R K^-1 and K t per measurement frame, K t / depth per plane: utils.py:66-68) and
  // the run plan.  With PLAN0 both are wave 0's job -- an in-order wave reads back its own LDS writes without a barrier -- and
  // the other waves go straight to the one barrier below ----
  const SweepScale sc = sweep_scale(a.W, a.H);
  if (!Cfg::PLAN0 || tid < 64) {
    const int first = Cfg::PLAN0 ? lane : tid, stride = Cfg::PLAN0 ? 64 : NT;
    gcfloat_p Hm_g = as_global(a.Hm) + static_cast<size_t>(b) * a.M * 9;
    gcfloat_p kt_g = as_global(a.kt) + static_cast<size_t>(b) * a.M * 3;
    for (int i = first; i < a.M * 9; i += stride) s_H[i] = Hm_g[i];
    for (int i = first; i < a.M * DP; i += stride) {
      const int m = i / DP, j = i - m * DP;
      float4v k = {0.0f, 0.0f, 0.0f, 0.0f};
      if (j < planes) {
        const float depth = plane_depth(a.inv_depth_base, a.inv_depth_step, d_block + j);
        k.x = kt_g[m * 3 + 0] / depth;
        k.y = kt_g[m * 3 + 1] / depth;
        k.z = kt_g[m * 3 + 2] / depth;
      }
      s_ktd[i] = k;
    }
    if (!Cfg::PLAN0) __syncthreads();   // (every wave plans: identical values, benign identical writes)
    SWEEP_TRACE(tr_setup = __builtin_amdgcn_s_memtime();)
    const int n = plan_runs<TW, TH, DP, CAP, Cfg::MINSEG>(a, s_H, s_ktd, tile_x, tile_y, planes, sc, lane, s_runs);
    if (tid == 0) s_n_runs = n;
  }
  __syncthreads();
  const int n_runs = __builtin_amdgcn_readfirstlane(s_n_runs);
  SWEEP_TRACE(const unsigned long long tr_plan = __builtin_amdgcn_s_memtime();)

  const int HW = a.H * a.W;
  // lane -> pixel: each 16-lane ds_read_b128 service group owns 16 consecutive pixels of one tile row (see sweep_lane_pixel)
  const int lane_pixel = sweep_lane_pixel(tid & 31);
  const int ptid = Cfg::PSPLIT == 1 ? tid : tid % NPIX;         // this thread's pixel within the tile
  const int j0 = Cfg::PSPLIT == 1 ? 0 : (tid / NPIX) * DPT;     // its first plane within the chunk (wave-uniform)
  const int x = tile_x * TW + (TW == 32 ? lane_pixel : TW == 16 ? (lane_pixel & 15) : ptid % TW);
  const int y = tile_y * TH + (TW == 32 ? ptid / 32 : TW == 16 ? (ptid >> 5) * 2 + (lane_pixel >> 4) : ptid / TW);
  const bool live = x < a.W && y < a.H;
  const float xf = static_cast<float>(x), yf = static_cast<float>(y);
  const int pix = live ? y * a.W + x : 0;
  gcfloat_p ref = as_global(a.image1) + static_cast<size_t>(b) * a.C * HW + pix;
  const char* tile_bytes = reinterpret_cast<const char*>(s_tile);
  const unsigned int plane_bytes = static_cast<unsigned int>(HW) * 4u;
  const unsigned int map_bytes = static_cast<unsigned int>(a.C) * plane_bytes;
  const __amdgpu_buffer_rsrc_t ref_rsrc = map_resource(as_global(a.image1) + static_cast<size_t>(b) * a.C * HW, map_bytes);
  const unsigned int ref_voffset = static_cast<unsigned int>(pix) * 4u;

  guint_p slot = nullptr;   // this workgroup's spill slot
  if (!GATHER) {
    const size_t groups = work_list_capacity_items(a.B, a.H, a.W, a.D, TW, TH, DP);
    slot = as_global(a.spill) + kSpillHeaderWords + groups + static_cast<size_t>(work.group) * spill_slot_words(a.M, DP);
  }
  int n_spilled = 0;      // workgroup-uniform
  int violated = 0;       // this thread saw a tap outside its staged box (round-off beyond the slack: not expected)

  // sum_m sum_c ref[c] * warped_m[c]: one accumulator over all measurement frames (sum over frames, then / C, then / M; for the
  // usual power-of-two C this is bit-identical to the reference's per-frame / C followed by the sum, otherwise it is one
  // rounding closer to exact).  Even- and odd-channel partial sums ride in the two halves of packed FMAs.
  float2v acc2[DPT];
#pragma unroll
  for (int j = 0; j < DPT; ++j) acc2[j] = float2v{0.0f, 0.0f};

  // ---- runs that cannot be staged: queue them for the second pass, or gather them here when there is none ----
  int first_staged = n_runs;
  for (int e = n_runs - 1; e >= 0; --e)
    if ((__builtin_amdgcn_readfirstlane(s_runs[e * kRunWords]) >> 24) == 1) first_staged = e;
  for (int e = 0; e < n_runs; ++e) {
    const int w0 = __builtin_amdgcn_readfirstlane(s_runs[e * kRunWords]);
    if ((w0 >> 24) != 0) continue;
    const int m = w0 & 0xff, seg_lo = (w0 >> 8) & 0xff, seg_len = (w0 >> 16) & 0xff;
    if (!GATHER && tid == 0) slot[1 + n_spilled] = spill_pack(m, seg_lo, seg_len);
    ++n_spilled;      // (GATHER: gathered behind the output write, see the end of the kernel)
  }

  // ---- staged runs, software-pipelined over (run, channel pass) stages ----
  constexpr int kPieces = NHWC ? (CAP * QPR + NT - 1) / NT : (CAP + NT - 1) / NT;
  // (channels-last quads: twice as many, 16 bytes each -- except in the 512-thread configuration, which is held to 128 registers: with 2 PRE
  // quads in flight its channels-last instantiations spilled 24 - 28 bytes to scratch; PRE / PRE - 1 fit)
  constexpr int kPreWanted = !NHWC ? Cfg::PRE : (Cfg::PSPLIT > 1 ? (GATHER ? Cfg::PRE - 1 : Cfg::PRE) : 2 * Cfg::PRE);
  constexpr int kPre = kPreWanted < kPieces ? kPreWanted : kPieces;
  constexpr int kPreRegs = NHWC ? 1 : QPR;   // float4 per piece
  if (first_staged < n_runs) {
    // state of the run being tapped
    int m = 0, seg_lo = 0, seg_hi = 0, row_bytes = 0, n_pieces = 0;
    int addr[DPT];
    float2v frac[DPT];
    // state of the run being staged (the same run, or the next one while the last pass of the current run is tapped)
    __amdgpu_buffer_rsrc_t meas_rsrc;
    unsigned int goff[kPieces];
    int st_n_pieces = 0;
    SampleBox st_box;
    int st_m = 0, st_lo = 0, st_hi = 0;

    auto read_run = [&](int e) {   // entry e -> the staging state: box, frame, planes, byte offset of every LDS piece in the measurement map
      const int* run = s_runs + e * kRunWords;
      const int w0 = __builtin_amdgcn_readfirstlane(run[0]), w3 = __builtin_amdgcn_readfirstlane(run[3]);
      st_m = w0 & 0xff;
      st_lo = (w0 >> 8) & 0xff;
      st_hi = st_lo + ((w0 >> 16) & 0xff);
      st_box.x_lo = __builtin_amdgcn_readfirstlane(run[1]);
      st_box.y_lo = __builtin_amdgcn_readfirstlane(run[2]);
      st_box.RW = w3 & 0xffff;
      st_box.RH = w3 >> 16;
      st_box.pitch = __builtin_amdgcn_readfirstlane(run[4]);
      st_box.state = 1;
      const int P = st_box.pitch, RS = P * st_box.RH;
      // NHWC: a piece is one 16-byte channel quad of one box position; NCHW: a piece is one box position (CCH dword loads).
      // Positions outside the image (zero apron), pad columns and pieces past the box get kBufferOutOfRange: the load
      // then returns zeros by itself.
      const unsigned int magic = 0xffffffffu / static_cast<unsigned int>(P) + 1u;   // r / P == mulhi(r, magic) for r < 2^16
      st_n_pieces = NHWC ? RS * QPR : RS;
#pragma unroll
      for (int k = 0; k < kPieces; ++k) {
        const int piece = tid + k * NT;
        const int r = NHWC ? piece / QPR : piece;
        const int ry = static_cast<int>(__umulhi(static_cast<unsigned int>(r), magic));
        const int rx = r - ry * P;
        const int gx = st_box.x_lo + rx, gy = st_box.y_lo + ry;
        const bool in = (piece < st_n_pieces) && (rx < st_box.RW) && (gx >= 0) && (gx < a.W) && (gy >= 0) && (gy < a.H);
        goff[k] = in ? static_cast<unsigned int>(NHWC ? (gy * a.W + gx) * a.C + (piece % QPR) * 4 : gy * a.W + gx) * 4u : kBufferOutOfRange;
      }
      meas_rsrc = map_resource(as_global(a.image2[st_m]) + static_cast<size_t>(b) * a.C * HW, map_bytes);
      SWEEP_TRACE(tr_records += static_cast<unsigned long long>(RS);)
    };
    auto adopt_run = [&]() {   // the staged run becomes the tapped one: this thread's tap addresses and fractional positions on its planes
      m = st_m; seg_lo = st_lo; seg_hi = st_hi; n_pieces = st_n_pieces;
      const int P = st_box.pitch;
      row_bytes = P * REC * 4;
      const SweepRay ray = sweep_ray(s_H + m * 9, xf, yf);
      const float4v* ktd_m = s_ktd + m * DP + j0;
#pragma unroll
      for (int j = 0; j < DPT; ++j) {  // all of this thread's planes in one basic block (those outside the run get addresses that are never used)
        const float4v kd = ktd_m[j];   // zeros beyond the last plane of a ragged chunk
        float ix, iy;
        sweep_sample(ray, kd.x, kd.y, kd.z, sc, &ix, &iy);
        const bool in_run = j0 + j >= seg_lo && j0 + j < seg_hi;   // wave-uniform
        const float fx = floorf(ix), fy = floorf(iy);
        int rx = static_cast<int>(fx) - st_box.x_lo, ry = static_cast<int>(fy) - st_box.y_lo;
        if (live && in_run) violated |= (static_cast<unsigned int>(rx) > static_cast<unsigned int>(st_box.RW - 2)) |
                                        (static_cast<unsigned int>(ry) > static_cast<unsigned int>(st_box.RH - 2));
        rx = min(max(rx, 0), st_box.RW - 2);
        ry = min(max(ry, 0), st_box.RH - 2);
        addr[j] = __mul24(__mul24(ry, P) + rx, REC * 4);   // full-rate 24-bit multiplies: ry, P, rx < 2^11
        frac[j] = float2v{ix - fx, iy - fy};
      }
    };
    auto load_ref = [&](int c0, float2v* rv) {
      // channels beyond C (last pass of a ragged channel count) re-read channel C-1 on both sides and are cancelled by rv = 0
#pragma unroll
      for (int c = 0; c < CCH; ++c) {
        const float v = buffer_f32(ref_rsrc, ref_voffset, static_cast<unsigned int>(min(c0 + c, a.C - 1)) * plane_bytes);
        rv[c / 2][c % 2] = (c0 + c < a.C) ? v : 0.0f;
      }
    };
    auto load_piece = [&](int k, int c0, float4v* v) {   // piece k of pass c0 of the run being staged into kPreRegs float4
      if (NHWC) {
        const int piece = tid + k * NT;
        unsigned int vo = goff[k];
        if (c0 + CCH > a.C && c0 + (piece % QPR) * 4 >= a.C) vo = kBufferOutOfRange;
        v[0] = buffer_f32x4(meas_rsrc, vo, static_cast<unsigned int>(c0) * 4u);
      } else {
#pragma unroll
        for (int c = 0; c < CCH; ++c)
          v[c / 4][c % 4] = buffer_f32(meas_rsrc, goff[k], static_cast<unsigned int>(min(c0 + c, a.C - 1)) * plane_bytes);
      }
    };
    auto store_piece = [&](int k, const float4v* v, int limit) {
      const int piece = tid + k * NT;
      if (piece < limit) {
        if (NHWC) {
          *reinterpret_cast<float4v*>(s_tile + (piece / QPR) * REC + (piece % QPR) * 4) = v[0];
        } else {
#pragma unroll
          for (int q = 0; q < QPR; ++q) *reinterpret_cast<float4v*>(s_tile + piece * REC + q * 4) = v[q];
        }
      }
    };

    float4v pre[kPre > 0 ? kPre * kPreRegs : 1];
    float2v rv_next[CCH / 2];
    auto prefetch = [&](int c0) {   // the staged run's pass c0: reference features and the first kPre pieces, requests only
      if (!Cfg::LEAN) load_ref(c0, rv_next);
#pragma unroll
      for (int k = 0; k < kPre; ++k)
        if (k * NT < st_n_pieces) load_piece(k, c0, pre + k * kPreRegs);   // workgroup-uniform
    };

    int e = first_staged, c0 = 0;
    read_run(e);
    prefetch(0);
    adopt_run();   // (overlaps the round trip of the first requests)
    for (;;) {
      SWEEP_TRACE(const unsigned long long tr_s0 = __builtin_amdgcn_s_memtime();)
      // ---- stage (run e, pass c0): pieces beyond the prefetched ones, a few in flight at a time, then the prefetched ones ----
      float2v rv[CCH / 2];
      if (Cfg::LEAN) {
        load_ref(c0, rv);   // (its round trip overlaps the stores and the barrier below)
      } else {
#pragma unroll
        for (int c = 0; c < CCH / 2; ++c) rv[c] = rv_next[c];
      }
      constexpr int kBatch = NHWC ? 4 : 1;
#pragma unroll
      for (int k0 = kPre; k0 < kPieces; k0 += kBatch) {
        if (k0 * NT < n_pieces) {   // workgroup-uniform
          float4v v[kBatch * kPreRegs];
#pragma unroll
          for (int kk = 0; kk < kBatch; ++kk)
            load_piece(k0 + kk < kPieces ? k0 + kk : kPieces - 1, c0, v + kk * kPreRegs);
#pragma unroll
          for (int kk = 0; kk < kBatch; ++kk)
            if (k0 + kk < kPieces) store_piece(k0 + kk, v + kk * kPreRegs, n_pieces);
        }
      }
#pragma unroll
      for (int k = 0; k < kPre; ++k)
        if (k * NT < n_pieces) store_piece(k, pre + k * kPreRegs, n_pieces);
      __syncthreads();
      SWEEP_TRACE(const unsigned long long tr_s1 = __builtin_amdgcn_s_memtime(); tr_stage += tr_s1 - tr_s0;)

      // ---- the next stage's requests go out before the taps of this one ----
      int e_next = e, c_next = c0 + CCH;
      bool next_run = false;
      if (c_next >= a.C) {
        c_next = 0;
        next_run = true;
        e_next = n_runs;
        for (int i = n_runs - 1; i > e; --i)
          if ((__builtin_amdgcn_readfirstlane(s_runs[i * kRunWords]) >> 24) == 1) e_next = i;
      }
      const bool more = e_next < n_runs;
      if (more) {
        if (next_run) read_run(e_next);
        prefetch(c_next);
      }

      // ---- taps of (run e, pass c0) ----
      if (Cfg::FASTFULL && seg_lo == 0 && seg_hi == DP) {   // the common case as one straight-line block
#pragma unroll
        for (int j = 0; j < DPT; ++j) tap_plane<QPR, REC, Cfg::LEAN>(tile_bytes, row_bytes, addr[j], frac[j], rv, &acc2[j]);
      } else {
#pragma unroll
        for (int j = 0; j < DPT; ++j)
          if (j0 + j >= seg_lo && j0 + j < seg_hi)   // wave-uniform
            tap_plane<QPR, REC, Cfg::LEAN>(tile_bytes, row_bytes, addr[j], frac[j], rv, &acc2[j]);
      }
      __syncthreads();
      SWEEP_TRACE(const unsigned long long tr_s2 = __builtin_amdgcn_s_memtime(); tr_taps += tr_s2 - tr_s1; ++tr_passes;)
      if (!more) break;
      if (next_run) adopt_run();
      SWEEP_TRACE(tr_switch += __builtin_amdgcn_s_memtime() - tr_s2;)
      e = e_next;
      c0 = c_next;
    }
  }

  // A tap outside its staged box can only come from round-off beyond the 0.05 px slack of the corner test.  It has never
  // been observed, but the result must not depend on it: the whole workgroup is redone through the gather path (second
  // pass, or inline when there is none).
  SWEEP_TRACE(const unsigned long long tr_loop_end = __builtin_amdgcn_s_memtime();)
  const int any_violated = __syncthreads_or(violated);
  if (any_violated) {
#pragma unroll
    for (int j = 0; j < DPT; ++j) acc2[j] = float2v{0.0f, 0.0f};
    n_spilled = 0;
    for (int m = 0; m < a.M; ++m) {
      if (!GATHER && tid == 0) slot[1 + n_spilled] = spill_pack(m, 0, planes);
      ++n_spilled;
    }
  }

  if (live) {
    gfloat_p out = as_global(a.out) + (static_cast<size_t>(b) * a.D + d_block + j0) * HW + pix;
    const float Cf = static_cast<float>(a.C), Mf = static_cast<float>(a.M);
    if (((a.C & (a.C - 1)) | (a.M & (a.M - 1))) == 0) {
      // both counts are powers of two (the usual 32 channels, 1 or 2 frames): x * 2^-k is x / 2^k, correctly rounded either
      // way, without the 2 x 10 instructions of an IEEE division per output
      const float rC = 1.0f / Cf, rM = 1.0f / Mf;
#pragma unroll
      for (int j = 0; j < DPT; ++j)
        if (j0 + j < planes) out[static_cast<size_t>(j) * HW] = ((acc2[j].x + acc2[j].y) * rC) * rM;
    } else {
#pragma unroll
      for (int j = 0; j < DPT; ++j)
        if (j0 + j < planes) out[static_cast<size_t>(j) * HW] = ((acc2[j].x + acc2[j].y) / Cf) / Mf;
    }
  }
#ifdef DVMVS_SWEEP_TRACE
  if (tid == 0 && work.group < kTraceGroups) {
    unsigned long long* t = g_sweep_trace + static_cast<size_t>(work.group) * kTraceWords;
    t[0] = tr_start; t[1] = tr_setup; t[2] = __builtin_amdgcn_s_memtime(); t[3] = tr_plan - tr_setup; t[4] = tr_stage; t[5] = tr_taps;
    t[6] = tr_passes; t[7] = tr_records; t[8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
    t[9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                                       // XCC_ID
    t[10] = tr_real0; t[11] = __builtin_amdgcn_s_memrealtime(); t[12] = static_cast<unsigned long long>(n_spilled) | (static_cast<unsigned long long>(blockIdx.x) << 32);
    t[13] = tr_switch; t[14] = static_cast<unsigned long long>(n_runs); t[15] = tr_loop_end - tr_start;
  }
#endif
  if (GATHER && n_spilled > 0) {
    // No second pass: the runs that could not be staged (none when the host's plan promised so: variants 4 / 5) are gathered here,
    // behind the output write, one plane at a time like the second pass does -- a cold loop that shares no register with the staged
    // path above (inlined into it, round 4's first form, it cost the sweep 29 registers and 1 us per launch).
    gcfloat_p kt_g = as_global(a.kt) + static_cast<size_t>(b) * a.M * 3;
    gfloat_p out = as_global(a.out) + (static_cast<size_t>(b) * a.D + d_block) * HW + pix;
    const int n_items = any_violated ? a.M : n_runs;
    for (int e = 0; e < n_items; ++e) {
      int m = e, seg_lo = 0, seg_len = planes;
      if (!any_violated) {
        const int w0 = __builtin_amdgcn_readfirstlane(s_runs[e * kRunWords]);
        if ((w0 >> 24) != 0) continue;
        m = w0 & 0xff, seg_lo = (w0 >> 8) & 0xff, seg_len = (w0 >> 16) & 0xff;
      }
      const SweepRay ray = sweep_ray(s_H + m * 9, xf, yf);
      gcfloat_p meas = as_global(a.image2[m]) + static_cast<size_t>(b) * a.C * HW;
#pragma unroll 1
      for (int j = max(seg_lo, j0); j < min(seg_lo + seg_len, j0 + DPT); ++j) {
        if (!live) continue;
        const float depth = plane_depth(a.inv_depth_base, a.inv_depth_step, d_block + j);
        const float part = gather_plane<NHWC>(a, meas, ref, HW, ray, kt_g[m * 3 + 0] / depth, kt_g[m * 3 + 1] / depth, kt_g[m * 3 + 2] / depth, sc);
        out[static_cast<size_t>(j) * HW] += (part / static_cast<float>(a.C)) / static_cast<float>(a.M);   // as the second pass scales
      }
    }
  }
  if (!GATHER && n_spilled > 0) {
    if (tid == 0) {
      slot[0] = static_cast<unsigned int>(n_spilled);
      const unsigned int at = atomicAdd(a.spill, 1u);
      as_global(a.spill)[kSpillHeaderWords + at] = static_cast<unsigned int>(work.group);
    }
  }
}

// Second pass.  A unit of work is one plane of one registered spill group: the workgroup's 256 threads are the tile's
// pixels, and each walks the group's items in the order the first pass queued them (measurement frame order), adding the
// frames whose queued run contains this plane with plain read-modify-writes.  A (pixel, plane) output has exactly one
// writer that applies its contributions in a fixed order, so the volume does not depend on scheduling, while the critical
// path of a unit is at most M single-plane gathers.
template <class Cfg, bool NHWC, int KCH = 8>
__global__ __launch_bounds__(Cfg::NPIX) void sweep_spill_kernel(CostVolumeArgs a) {
  constexpr int TW = Cfg::TW, TH = Cfg::TH, DP = Cfg::DP;
  const guint_p spill = as_global(a.spill);
  const unsigned int units = spill[0] * DP;
  const int tid = threadIdx.x;
  const int HW = a.H * a.W;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const int chunks = (a.D + DP - 1) / DP;
  const int per_b = tiles_x * tiles_y * chunks;
  const size_t groups = work_list_capacity_items(a.B, a.H, a.W, a.D, TW, TH, DP);
  const SweepScale sc = sweep_scale(a.W, a.H);
  for (unsigned int u = blockIdx.x; u < units; u += gridDim.x) {
    const int j = static_cast<int>(u % DP);
    const int group = static_cast<int>(spill[kSpillHeaderWords + u / DP]);
    const guint_p slot = spill + kSpillHeaderWords + groups + static_cast<size_t>(group) * spill_slot_words(a.M, DP);
    int b, tile, d;
    if (a.items != nullptr) {      // the group is an item of the work list: its own plane range
      const guint_p items = as_global(const_cast<unsigned int*>(a.items));
      const unsigned int w0 = items[kWorkListHeaderWords + 2 * group], w1 = items[kWorkListHeaderWords + 2 * group + 1];
      b = static_cast<int>(w0 >> 16);
      tile = static_cast<int>(w0 & 0xffffu);
      d = j < static_cast<int>(w1 >> 16) ? static_cast<int>(w1 & 0xffffu) + j : a.D;      // (planes beyond the item: nothing to do)
    } else {
      b = group / per_b;
      const int rem = group - b * per_b;
      tile = rem / chunks;
      d = (chunks - 1 - (rem - tile * chunks)) * DP + j;
    }
    const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const int x = tile_x * TW + tid % TW, y = tile_y * TH + tid / TW;
    const bool live = d < a.D && x < a.W && y < a.H;
    const float xf = static_cast<float>(x), yf = static_cast<float>(y);
    const int pix = live ? y * a.W + x : 0;
    gcfloat_p ref = as_global(a.image1) + static_cast<size_t>(b) * a.C * HW + pix;
    gfloat_p out = as_global(a.out) + (static_cast<size_t>(b) * a.D + min(d, a.D - 1)) * HW + pix;
    const float depth = plane_depth(a.inv_depth_base, a.inv_depth_step, min(d, a.D - 1));
    const int n_items = static_cast<int>(slot[0]);
    float value = 0.0f;
    bool touched = false;
    for (int it = 0; it < n_items; ++it) {
      const unsigned int w = slot[1 + it];
      const int m = static_cast<int>(w >> 16), seg_lo = static_cast<int>((w >> 8) & 0xffu), seg_len = static_cast<int>(w & 0xffu);
      if (j < seg_lo || j >= seg_lo + seg_len) continue;   // workgroup-uniform
      if (!live) continue;
      gcfloat_p Hm_g = as_global(a.Hm) + (static_cast<size_t>(b) * a.M + m) * 9;   // the matrices the first pass used
      gcfloat_p kt_g = as_global(a.kt) + (static_cast<size_t>(b) * a.M + m) * 3;
      float Hm[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) Hm[k] = Hm_g[k];
      const SweepRay ray = sweep_ray(Hm, xf, yf);
      const float part = gather_plane<NHWC, KCH>(a, as_global(a.image2[m]) + static_cast<size_t>(b) * a.C * HW, ref, HW, ray,
                                                 kt_g[0] / depth, kt_g[1] / depth, kt_g[2] / depth, sc);
      if (!touched) value = *out;
      touched = true;
      value += (part / static_cast<float>(a.C)) / static_cast<float>(a.M);   // same scaling order as the first pass
    }
    if (touched) *out = value;
  }
  // Leave the header as the next call expects it (nothing to do for an empty list).  Only the workgroups that HAD a unit take a
  // ticket, and the last of them clears both words: a workgroup without one (blockIdx >= units) has read word 0 before it got here
  // and never reads it again, and one that starts so late that it already sees the cleared header finds units == 0 -- it would not
  // have had a unit anyway.  (Round 3 let all 1024 workgroups of the grid take a ticket: 1024 atomics on one address, serialised in
  // L2 -- the fixed ~20 us a non-empty second pass cost however few units it had, tools/sweep_select_fit.py.)
  __syncthreads();
  if (tid == 0 && blockIdx.x < units) {
    const unsigned int participants = units < gridDim.x ? units : gridDim.x;
    const unsigned int ticket = atomicAdd(a.spill + 1, 1u);
    if (ticket == participants - 1) {
      spill[0] = 0u;
      spill[1] = 0u;
    }
  }
}

// ---- host-side model of the run plan --------------------------------------------------------------------------------------
// plan_runs restated for the HOST (no HIP call, no device memory): the caller's matrices are on the host before the launch (ABI 3),
// so which sweep configuration suits a keyframe pair can be decided there, without a device round trip.  Same candidates, same box
// rule, same greedy choice; the sample positions use IEEE fp32 division, which is what the kernel's refined-reciprocal sequence
// produces for normal operands, so the model's plan is the kernel's plan (it only has to be close: it selects a configuration, it
// does not change any configuration's results).  stats[0..5]: staged runs, LDS records of all staged runs, runs entirely outside
// the image, runs queued for the second pass, their planes (second-pass gathers per pixel), workgroups with at least one queued run,
// the largest number of staged runs any ONE workgroup has (its chain of (run x channel pass) stages is the launch's span: all workgroups
// are resident at once, so the slowest one decides), the largest number of queued planes of one workgroup.
#pragma clang fp contract(off)
inline void host_sweep_position(const SweepRay& r, float kx, float ky, float kz, const SweepScale& s, float* ix, float* iy, float* denom_out) {
  const float denom = (r.Z0 + kz) + 1e-8f;
  *denom_out = denom;
  const float u = (r.X0 + kx) / denom, v = (r.Y0 + ky) / denom;
  *ix = ((((u - s.wn) / s.wn) + 1.0f) * 0.5f) * s.Wm1;
  *iy = ((((v - s.hn) / s.hn) + 1.0f) * 0.5f) * s.Hm1;
}

// One (batch item, tile, chunk) of the sweep on the host: the corner rays per measurement frame and the K t / depth table of the
// chunk's planes; plan(lo, hi) walks planes [lo, hi) of every frame exactly as plan_runs does for a work item with those planes.
struct HostRunCounts {
  long long staged_runs, staged_records, empty_runs, queued_runs, queued_planes;
};

template <class Cfg>
struct HostTilePlanner {
  static constexpr int TW = Cfg::TW, TH = Cfg::TH, DP = Cfg::DP, CAP = Cfg::CAP, MINSEG = Cfg::MINSEG;
  int M, H, W;
  SweepScale sc;
  float edge;
  SweepRay ray[DVMVS_MAX_MEASUREMENTS][4];
  float ktd[DVMVS_MAX_MEASUREMENTS][DP][3];

  void set_shape(int M_, int H_, int W_) {
    M = M_; H = H_; W = W_;
    sc.Wf = static_cast<float>(W); sc.Hf = static_cast<float>(H);
    sc.wn = sc.Wf * 0.5f; sc.hn = sc.Hf * 0.5f;
    sc.r_wn = 1.0f / sc.wn; sc.r_hn = 1.0f / sc.hn;
    sc.Wm1 = static_cast<float>(W - 1); sc.Hm1 = static_cast<float>(H - 1);
  }
  void set_chunk(const float* kt_b, int d_block, int planes, double inv_base, double inv_step) {
    for (int m = 0; m < M; ++m)
      for (int j = 0; j < planes; ++j) {
        const float depth = static_cast<float>(1.0 / (inv_base + static_cast<double>(d_block + j) * inv_step));
        for (int k = 0; k < 3; ++k) ktd[m][j][k] = kt_b[m * 3 + k] / depth;
      }
  }
  void set_tile(const float* Hm_b, int tile_x, int tile_y) {
    const int x_first = tile_x * TW, x_last = x_first + TW - 1 < W - 1 ? x_first + TW - 1 : W - 1;
    const int y_first = tile_y * TH, y_last = y_first + TH - 1 < H - 1 ? y_first + TH - 1 : H - 1;
    edge = static_cast<float>(x_last - x_first > 1 ? x_last - x_first : 1);
    for (int m = 0; m < M; ++m)
      for (int c = 0; c < 4; ++c)
        ray[m][c] = sweep_ray(Hm_b + m * 9, static_cast<float>((c & 1) ? x_last : x_first), static_cast<float>((c & 2) ? y_last : y_first));
  }
  // planes [first, last) of the chunk (indices into ktd), all frames
  HostRunCounts plan(int first, int last) const {
    HostRunCounts n = {0, 0, 0, 0, 0};
    for (int m = 0; m < M; ++m) {
      int lo = first, hint = DP;
      while (lo < last) {
        const int len0 = last - lo < hint ? last - lo : hint;
        int picked_len = len0, picked_state = 0, picked_records = 0;
        for (int candidate = 0; candidate < 4; ++candidate) {
          int len = len0;
          for (int i = 0; i < candidate; ++i) len = (len + 1) / 2 > MINSEG ? (len + 1) / 2 : MINSEG;
          if (len > len0) len = len0;
          // the eight corners (4 tile corners x first / last plane of the run) in array form: the loops below vectorise (the chain of
          // dependent fp32 divisions of one corner after the other was most of the host cost of a plan)
          float X[8], Y[8], den[8], ux[8], uy[8];
          for (int corner = 0; corner < 8; ++corner) {
            const float* k = ktd[m][(corner & 4) ? lo + len - 1 : lo];
            const SweepRay& r = ray[m][corner & 3];
            X[corner] = r.X0 + k[0];
            Y[corner] = r.Y0 + k[1];
            den[corner] = (r.Z0 + k[2]) + 1e-8f;
          }
          for (int corner = 0; corner < 8; ++corner) {
            const float u = X[corner] / den[corner], v = Y[corner] / den[corner];
            ux[corner] = ((((u - sc.wn) / sc.wn) + 1.0f) * 0.5f) * sc.Wm1;
            uy[corner] = ((((v - sc.hn) / sc.hn) + 1.0f) * 0.5f) * sc.Hm1;
          }
          float lo_x = ux[0], hi_x = ux[0], lo_y = uy[0], hi_y = uy[0];
          bool finite = true;
          for (int corner = 0; corner < 8; ++corner) {
            finite = finite && (ux[corner] > -1e6f) && (ux[corner] < 1e6f) && (uy[corner] > -1e6f) && (uy[corner] < 1e6f) && (den[corner] > 1e-6f);
            lo_x = fminf(lo_x, ux[corner]); hi_x = fmaxf(hi_x, ux[corner]);
            lo_y = fminf(lo_y, uy[corner]); hi_y = fmaxf(hi_y, uy[corner]);
          }
          const float top[2][2] = {{ux[0], uy[0]}, {ux[1], uy[1]}};
          const bool outside = (hi_x + 0.05f <= -1.0f) || (lo_x - 0.05f >= sc.Wf) || (hi_y + 0.05f <= -1.0f) || (lo_y - 0.05f >= sc.Hf);
          int state = 0, records = 0;
          if (finite && outside) state = 2;
          else if (finite) {
            const int bx_lo = static_cast<int>(floorf(fmaxf(lo_x - 0.05f, -1.0f))), by_lo = static_cast<int>(floorf(fmaxf(lo_y - 0.05f, -1.0f)));
            const int x_lo = bx_lo > -1 ? bx_lo : -1, y_lo = by_lo > -1 ? by_lo : -1;
            const int bx_hi = static_cast<int>(floorf(fminf(hi_x + 0.05f, sc.Wf))), by_hi = static_cast<int>(floorf(fminf(hi_y + 0.05f, sc.Hf)));
            const int x_hi = (bx_hi < W ? bx_hi : W) + 1, y_hi = (by_hi < H ? by_hi : H) + 1;
            const int RW = x_hi - x_lo + 1, RH = y_hi - y_lo + 1;
            const int pitch = RW + ((sweep_pitch_residue((top[1][0] - top[0][0]) / edge, (top[1][1] - top[0][1]) / edge) - RW) & 15);
            records = pitch * RH;
            state = records <= CAP ? 1 : 0;
          }
          if (state != 0 || len <= MINSEG || candidate == 3) {
            picked_len = len; picked_state = state; picked_records = records;
            break;
          }
        }
        if (picked_state == 1) { ++n.staged_runs; n.staged_records += picked_records; }
        else if (picked_state == 2) ++n.empty_runs;
        else { ++n.queued_runs; n.queued_planes += picked_len; }
        lo += picked_len;
        hint = picked_len > MINSEG ? picked_len : MINSEG;
      }
    }
    return n;
  }
};

template <class Cfg>
void host_plan_stats(const float* Hm, const float* kt, int B, int M, int H, int W, int D, double inv_base, double inv_step, long long* stats) {
  constexpr int TW = Cfg::TW, TH = Cfg::TH, DP = Cfg::DP;
  HostTilePlanner<Cfg> planner;
  planner.set_shape(M, H, W);
  for (int i = 0; i < 8; ++i) stats[i] = 0;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, chunks = (D + DP - 1) / DP;
  for (int b = 0; b < B; ++b)
    for (int chunk = 0; chunk < chunks; ++chunk) {
      const int d_block = chunk * DP, planes = D - d_block < DP ? D - d_block : DP;
      planner.set_chunk(kt + static_cast<size_t>(b) * M * 3, d_block, planes, inv_base, inv_step);
      for (int tile_y = 0; tile_y < tiles_y; ++tile_y)
        for (int tile_x = 0; tile_x < tiles_x; ++tile_x) {
          planner.set_tile(Hm + static_cast<size_t>(b) * M * 9, tile_x, tile_y);
          const HostRunCounts n = planner.plan(0, planes);
          stats[0] += n.staged_runs; stats[1] += n.staged_records; stats[2] += n.empty_runs; stats[3] += n.queued_runs; stats[4] += n.queued_planes;
          if (n.queued_runs > 0) ++stats[5];
          if (n.staged_runs > stats[6]) stats[6] = n.staged_runs;
          if (n.queued_planes > stats[7]) stats[7] = n.queued_planes;
        }
    }
}

// ---- work list ------------------------------------------------------------------------------------------------------------
// All workgroups of a launch are resident at once, so the launch lasts as long as its slowest workgroup: one whose sample boxes had
// to be halved works through (runs x channel passes) stages -- 5 to 8 runs instead of 2 on wide-baseline and forward-motion pairs,
// 55-135 us instead of 35 (tools/sweep_select_fit.py on all 285 keyframe pairs of the sample scene).  The host knows the plan before
// the launch, so it hands the kernel a WORK LIST in which such a (tile, chunk) is cut into plane sub-ranges with at most
// kMaxRunsPerItem staged runs each, processed by separate workgroups in parallel.  Words: [0] = number of items, [1] = 0, then per item
// {tile | batch item << 16, first plane | number of planes << 16}.  The first `static` items sit at the positions the static numbering
// gives their (tile, chunk) -- XCD locality and the CU mix stay as decode_work arranges them --, the extra pieces follow.
constexpr int kMaxRunsPerItem = 3;

// `stats` (optional, 8 entries as host_plan_stats fills them): the plan statistics of the uncut (tile, chunk) pairs, gathered in the same walk
template <class Cfg>
int host_build_work_list(const float* Hm, const float* kt, int B, int M, int H, int W, int D, double inv_base, double inv_step,
                         unsigned int* items, size_t capacity_words, long long* stats = nullptr) {
  constexpr int TW = Cfg::TW, TH = Cfg::TH, DP = Cfg::DP;
  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, chunks = (D + DP - 1) / DP;
  const int tiles = tiles_x * tiles_y;
  const long long total = static_cast<long long>(tiles) * chunks * B;
  if (tiles > 65535 || B > 65535 || D > 65535) return DVMVS_EUNSUPPORTED;
  const size_t positions = static_cast<size_t>((total + 7) / 8 * 8);
  const size_t capacity = work_list_capacity_items(B, H, W, D, TW, TH, DP);
  if (capacity_words < kWorkListHeaderWords + 2 * capacity) return DVMVS_EINVAL;
  HostTilePlanner<Cfg> planner;
  planner.set_shape(M, H, W);
  size_t extra = positions;      // next free position behind the statically numbered ones
  int cached_b = -1, cached_chunk = -1;
  if (stats)
    for (int i = 0; i < 8; ++i) stats[i] = 0;
  for (size_t block = 0; block < positions; ++block) {
    const SweepWork work = decode_work<Cfg::ORDER>(static_cast<int>(block), tiles, chunks, B);
    unsigned int* item = items + kWorkListHeaderWords + 2 * block;
    item[0] = 0u; item[1] = 0u;      // (padding positions: no planes)
    if (!work.valid) continue;
    const int d_block = work.chunk * DP, planes = D - d_block < DP ? D - d_block : DP;
    if (work.b != cached_b || work.chunk != cached_chunk) {
      planner.set_chunk(kt + static_cast<size_t>(work.b) * M * 3, d_block, planes, inv_base, inv_step);
      cached_b = work.b; cached_chunk = work.chunk;
    }
    planner.set_tile(Hm + static_cast<size_t>(work.b) * M * 9, work.tile % tiles_x, work.tile / tiles_x);
    // cut [0, planes) into pieces of at most kMaxRunsPerItem staged runs (halving; a piece of MINSEG planes is never cut)
    int piece_lo[DP], piece_hi[DP], n_pieces = 0;
    int stack_lo[2 * DP], stack_hi[2 * DP], top = 0;
    stack_lo[0] = 0; stack_hi[0] = planes; top = 1;
    bool whole = true;
    while (top > 0) {
      --top;
      const int lo = stack_lo[top], hi = stack_hi[top];
      const HostRunCounts n = planner.plan(lo, hi);
      if (whole && stats) {      // (the first range popped is the whole chunk: its plan is the static numbering's)
        stats[0] += n.staged_runs; stats[1] += n.staged_records; stats[2] += n.empty_runs; stats[3] += n.queued_runs; stats[4] += n.queued_planes;
        if (n.queued_runs > 0) ++stats[5];
        if (n.staged_runs > stats[6]) stats[6] = n.staged_runs;
        if (n.queued_planes > stats[7]) stats[7] = n.queued_planes;
      }
      whole = false;
      // a cut replaces this range by two; each of the `top` ranges still on the stack and of the pieces already settled ends up as at
      // least one item, all but the first behind `extra` -- the guard has to count them, or a nested cut writes past the buffer
      const bool cut = hi - lo > Cfg::MINSEG && n.staged_runs > kMaxRunsPerItem && extra + n_pieces + top + 1 < capacity;
      if (cut) {
        const int mid = lo + (hi - lo + 1) / 2;
        stack_lo[top] = mid; stack_hi[top] = hi; ++top;     // (popped second: pieces come out in plane order)
        stack_lo[top] = lo; stack_hi[top] = mid; ++top;
      } else {
        piece_lo[n_pieces] = lo; piece_hi[n_pieces] = hi; ++n_pieces;
      }
    }
    const unsigned int w0 = static_cast<unsigned int>(work.tile) | (static_cast<unsigned int>(work.b) << 16);
    for (int i = 0; i < n_pieces; ++i) {
      unsigned int* dst = i == 0 ? item : items + kWorkListHeaderWords + 2 * extra++;
      dst[0] = w0;
      dst[1] = static_cast<unsigned int>(d_block + piece_lo[i]) | (static_cast<unsigned int>(piece_hi[i] - piece_lo[i]) << 16);
    }
  }
  if (extra > capacity) return DVMVS_EINVAL;      // (unreachable: the cut rule counts every pending range)
  items[0] = static_cast<unsigned int>(extra);
  items[1] = 0u;
  return static_cast<int>(kWorkListHeaderWords + 2 * extra);
}
#pragma clang fp contract(fast)

// ---- launch ------------------------------------------------------------------------------------------------------------
constexpr int kMaxDevices = 64;
constexpr int kSpillGrid = 1024;  // second-pass workgroups, grid-stride over the queued units: four per CU hide the gather's latency (one per
                                  // CU: +10-15 us on the geometries that spill); an empty pass costs ~3 us either way
constexpr int kSpillChannels = 8;  // channels x 4 taps of one sample in flight per thread in the second pass

template <class Kernel>
int raise_dynamic_lds_limit(Kernel kernel, size_t bytes, bool* configured) {
  // the limit is a per-device function attribute; setting it is idempotent, racing threads write the same value
  int device = 0;
  DVMVS_RETURN_IF_HIP(hipGetDevice(&device));
  const bool tracked = device >= 0 && device < kMaxDevices;
  if (!tracked || !configured[device]) {
    DVMVS_RETURN_IF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            static_cast<int>(bytes)));
    if (tracked) configured[device] = true;
  }
  return 0;
}

template <class Cfg, bool NHWC>
int launch_sweep_tiled_layout(const CostVolumeArgs& a, hipStream_t stream, int spill_grid = kSpillGrid, int spill_kch = kSpillChannels) {
  const long long tiles = static_cast<long long>((a.W + Cfg::TW - 1) / Cfg::TW) * ((a.H + Cfg::TH - 1) / Cfg::TH);
  const long long total = tiles * ((a.D + Cfg::DP - 1) / Cfg::DP) * a.B;
  if (total > (1LL << 30)) return DVMVS_EUNSUPPORTED;
  // with a work list: one workgroup per possible item (those beyond the list's count return at once)
  const unsigned int grid = a.items != nullptr ? static_cast<unsigned int>(work_list_capacity_items(a.B, a.H, a.W, a.D, Cfg::TW, Cfg::TH, Cfg::DP))
                                               : static_cast<unsigned int>(Cfg::ORDER >= 1 ? (total + 7) / 8 * 8 : total);
  if (a.spill == nullptr) {
    static bool configured[kMaxDevices] = {};
    auto kernel = sweep_tiled_kernel<Cfg, NHWC, true>;
    const int rc = raise_dynamic_lds_limit(kernel, Cfg::kLdsBytes, configured);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(Cfg::NT), Cfg::kLdsBytes, stream, a);
    return launch_status();
  }
  static bool configured[kMaxDevices] = {};
  auto kernel = sweep_tiled_kernel<Cfg, NHWC, false>;
  const int rc = raise_dynamic_lds_limit(kernel, Cfg::kLdsBytes, configured);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(Cfg::NT), Cfg::kLdsBytes, stream, a);
  const int rc2 = launch_status();
  if (rc2 != 0) return rc2;
  if (spill_kch >= 32) hipLaunchKernelGGL((sweep_spill_kernel<Cfg, NHWC, 32>), dim3(spill_grid), dim3(Cfg::NPIX), 0, stream, a);
  else if (spill_kch >= 16) hipLaunchKernelGGL((sweep_spill_kernel<Cfg, NHWC, 16>), dim3(spill_grid), dim3(Cfg::NPIX), 0, stream, a);
  else hipLaunchKernelGGL((sweep_spill_kernel<Cfg, NHWC, 8>), dim3(spill_grid), dim3(Cfg::NPIX), 0, stream, a);
  return launch_status();
}

template <class Cfg>
int launch_sweep_tiled(const CostVolumeArgs& a, hipStream_t stream, int spill_grid = kSpillGrid, int spill_kch = kSpillChannels) {
  return a.image2_nhwc ? launch_sweep_tiled_layout<Cfg, true>(a, stream, spill_grid, spill_kch)
                       : launch_sweep_tiled_layout<Cfg, false>(a, stream, spill_grid, spill_kch);
}

// the shipped configuration; the spill workspace is sized for it
using SweepDefault = SweepConfig<32, 8, 8, 8, 1024, 2>;   // <TW, TH, DP, CCH, CAP, MINSEG>: 3 workgroups / CU, chunk-major numbering, 2 pieces prefetched

template <class Cfg>
size_t spill_words_for(int B, int M, int H, int W, int D) {
  const size_t groups = work_list_capacity_items(B, H, W, D, Cfg::TW, Cfg::TH, Cfg::DP);
  return kSpillHeaderWords + groups + groups * spill_slot_words(M, Cfg::DP);
}

// one slot per possible work item of the product configurations (both use the 32x8x8 tiling); the tools-only tuning build also sizes it
// for its finest tiling (each launch indexes it with its own tiling)
size_t sweep_spill_words(int B, int M, int H, int W, int D) {
  const size_t a = spill_words_for<SweepDefault>(B, M, H, W, D);
#ifdef DVMVS_SWEEP_TUNING
  const size_t b = spill_words_for<SweepConfig<32, 4, 8, 8, 640, 2>>(B, M, H, W, D);
  return a > b ? a : b;
#else
  return a;
#endif
}

int launch_sweep_default(const CostVolumeArgs& a, hipStream_t stream) { return launch_sweep_tiled<SweepDefault>(a, stream); }

// the wide-baseline configuration: the same 32x8x8 tiling (same spill workspace), 72 KB boxes, planes split over two thread groups of
// one 512-thread workgroup (two per CU).  Runs that the default configuration has to halve or queue for the second pass are staged
// whole: -30 us on wide-baseline / forward-motion pairs, +7 us on easy sideways pairs (profiles/r03_sweep_experiment_log.md section 3),
// so the caller picks per keyframe pair (dvmvs_sweep_select_variant: the host-side plan model below).
using SweepWide = SweepConfig<32, 8, 8, 8, 1536, 2, 4, 2, 2, true, true, 2>;
int launch_sweep_wide(const CostVolumeArgs& a, hipStream_t stream) { return launch_sweep_tiled<SweepWide>(a, stream); }

size_t sweep_work_list_words(int B, int H, int W, int D) {
  return kWorkListHeaderWords + 2 * work_list_capacity_items(B, H, W, D, SweepDefault::TW, SweepDefault::TH, SweepDefault::DP);
}

int sweep_work_list_host(int configuration, const float* Hm, const float* kt, int B, int M, int H, int W, int D, double inv_base, double inv_step,
                         unsigned int* items, size_t capacity_words, long long* stats) {
  if (configuration == 1) return host_build_work_list<SweepWide>(Hm, kt, B, M, H, W, D, inv_base, inv_step, items, capacity_words, stats);
  return host_build_work_list<SweepDefault>(Hm, kt, B, M, H, W, D, inv_base, inv_step, items, capacity_words, stats);
}

void sweep_plan_stats_host(int configuration, const float* Hm, const float* kt, int B, int M, int H, int W, int D, double inv_base, double inv_step,
                           long long* stats) {
  if (configuration == 1) host_plan_stats<SweepWide>(Hm, kt, B, M, H, W, D, inv_base, inv_step, stats);
  else host_plan_stats<SweepDefault>(Hm, kt, B, M, H, W, D, inv_base, inv_step, stats);
}

#ifdef DVMVS_SWEEP_TUNING   // tools-only builds (`make tuning`, `make trace`); the product library carries the shipped configuration only
// tuning configurations for tools/cv_microbench.py: <TW, TH, DP, CCH, CAP, MINSEG, WAVES, ORDER, PRE, PLAN0, FASTFULL>.  All use the 32x8x8
// tiling the spill workspace is sized for.
int launch_sweep_tuning(int which_and_grid, const CostVolumeArgs& a, hipStream_t stream) {
  const int which = which_and_grid & 15, g0 = 256 << ((which_and_grid >> 4) & 3);   // + 16 / 32 / 48: second-pass grid of 512 / 1024 / 2048 (else 256)
  const int kch = 8 << ((which_and_grid >> 6) & 3);                                         // + 64 / 128: 16 / 32 channels in flight there
#define g g0, kch
  switch (which) {
    case 0: return launch_sweep_tiled<SweepConfig<32, 8, 8, 8, 1024, 2, 3, 2, 2, true, false>>(a, stream, g);   // the shipped configuration
    case 1: return launch_sweep_tiled<SweepConfig<32, 8, 8, 8, 1024, 2, 3, 1, 2, true, true>>(a, stream, g);    // round 2's tile-major numbering
    case 2: return launch_sweep_tiled<SweepConfig<32, 8, 8, 8, 1024, 2, 3, 3, 2, true, true>>(a, stream, g);    // rotated chunks
    case 3: return launch_sweep_tiled<SweepConfig<32, 8, 8, 8, 1024, 2, 3, 2, 3, true, true>>(a, stream, g);    // three pieces prefetched
    case 4: return launch_sweep_tiled<SweepConfig<32, 8, 8, 8, 1024, 2, 3, 2, 0, true, true>>(a, stream, g);    // no prefetch
    case 5: return launch_sweep_tiled<SweepConfig<32, 8, 8, 8, 1024, 2, 3, 2, 2, true, true>>(a, stream, g);    // with the straight-line full-chunk tap block
    case 6: return launch_sweep_tiled<SweepConfig<32, 8, 8, 8, 1024, 2, 2, 2, 4, true, true>>(a, stream, g);    // 2 waves / SIMD of registers, whole pass prefetched
    case 7: return launch_sweep_tiled<SweepConfig<32, 8, 8, 8, 1280, 2, 2, 2, 3, true, true>>(a, stream, g);    // 60 KB: 2 / CU
    case 8: return launch_sweep_tiled<SweepConfig<32, 8, 8, 8, 1024, 2, 3, 0, 2, true, true>>(a, stream, g);    // linear numbering (no XCD awareness)
    case 9: return launch_sweep_tiled<SweepConfig<32, 8, 8, 8, 1024, 2, 4, 2, 2, true, true, 2>>(a, stream, g);   // 512 threads: planes split over two thread groups, 2 / CU
    case 10: return launch_sweep_tiled<SweepConfig<32, 8, 8, 8, 1536, 2, 4, 2, 2, true, true, 2>>(a, stream, g);  // ... with 72 KB boxes
    case 11: return launch_sweep_tiled<SweepConfig<32, 8, 8, 8, 1536, 2, 4, 2, 3, true, true, 2>>(a, stream, g);  // ... whole pass prefetched
    case 12: return launch_sweep_tiled<SweepConfig<32, 8, 8, 8, 1024, 2, 6, 2, 1, true, true, 2>>(a, stream, g);  // ... 3 / CU (<= 80 registers)
    case 13: return launch_sweep_tiled<SweepConfig<32, 8, 8, 8, 1280, 2, 4, 2, 2, true, true, 2>>(a, stream, g);  // ... 60 KB boxes
    // 32x4-pixel tiles, planes split over two 128-thread groups: 1280 workgroups of 4 waves, five per CU, all resident
    case 14: return launch_sweep_tiled<SweepConfig<32, 4, 8, 8, 576, 2, 5, 2, 1, true, true, 2, true>>(a, stream, g);   // 27 KB + tables: five fit in 160 KB
    case 15: return launch_sweep_tiled<SweepConfig<32, 4, 8, 8, 576, 2, 5, 2, 1, true, false, 2, true>>(a, stream, g);
    default: return DVMVS_EINVAL;
  }
#undef g
}
#else
int launch_sweep_tuning(int, const CostVolumeArgs&, hipStream_t) { return DVMVS_EINVAL; }
#endif

}  // namespace dvmvs

#ifdef DVMVS_SWEEP_TRACE
extern "C" int dvmvs_debug_sweep_trace(unsigned long long* host, int groups) {
  if (groups > dvmvs::kTraceGroups) groups = dvmvs::kTraceGroups;
  return static_cast<int>(hipMemcpyFromSymbol(host, HIP_SYMBOL(dvmvs::g_sweep_trace), sizeof(unsigned long long) * dvmvs::kTraceWords * groups));
}
#endif
