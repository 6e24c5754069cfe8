// EXPERIMENT (tools only): the sweep kernel with adaptive plane counts and claimable half units -- DESIGN.md section 7, "next".
// Not part of libdvmvs_hip.so: `make trace` compiles this file INSTEAD of sweep_tiled.hip (which it includes unchanged) into
// libdvmvs_hip_trace.so, and tools/sweep_adaptive_bench.py drives it through dvmvs_debug_sweep_adaptive.
//
// A workgroup whose first fit says that its 8 planes need several runs keeps planes [0, 4) and publishes planes [4, 8) of its
// unit (all measurement frames of them, so every output keeps exactly one writer and its summation order).  Workgroups that
// are done claim published halves; an owner claims its own half back when nobody did.  Nobody waits for anybody.
// The kernel body below is sweep_tiled_kernel<Cfg, NCHW, two-pass> with a task loop around it; what differs is marked.
#include <type_traits>

#include "sweep_tiled.hip"

namespace dvmvs {

// timeline of the experiment (wall clock, 100 MHz): per workgroup and per published half
__device__ unsigned long long g_adaptive_wg[8192 * 8];
__device__ unsigned long long g_adaptive_half[8192 * 4];

// The timeline costs registers the kernel does not have (80 B of scratch per lane, +12 us per launch): it is compiled in only
// with -DDVMVS_ADAPTIVE_TIMELINE, for tools/sweep_adaptive_bench.py --timeline; timings are taken without it.
#ifdef DVMVS_ADAPTIVE_TIMELINE
#define ADAPTIVE_TL(...) __VA_ARGS__
#else
#define ADAPTIVE_TL(...)
#endif

constexpr int kStealHeaderWords = 16;   // [0] published halves, [1] claim cursor, [2] error flag; then the list, then one flag per group

// ADAPTIVE: publish / claim half units.  PREPOS: the next frame's positions are computed inside the tap blocks of the current
// frame's first pass -- measured: that form needs 232 B of scratch per lane and runs 51 us where the plain kernel runs 34-38
// (any scratch at all costs this kernel >= 10 us); it needs ~30 registers freed elsewhere first.  <false, false> is the shipped
// first pass without its trace instrumentation.
template <class Cfg, bool ADAPTIVE, bool PREPOS>
__global__ __launch_bounds__(Cfg::NT, Cfg::WAVES) void sweep_adaptive_kernel(CostVolumeArgs a, unsigned int* steal_raw, int claim_once) {
  constexpr bool NHWC = false, GATHER = false;
  constexpr int TW = Cfg::TW, TH = Cfg::TH, DP = Cfg::DP, CCH = Cfg::CCH, CAP = Cfg::CAP, NT = Cfg::NT, REC = Cfg::REC;
  constexpr int QPR = CCH / 4;   // 16-byte quads per record
  extern __shared__ __attribute__((aligned(16))) float s_tile[];   // [CAP][REC]
  __shared__ float s_H[DVMVS_MAX_MEASUREMENTS * 9];
  __shared__ float s_kt[DVMVS_MAX_MEASUREMENTS * 3];
  __shared__ float4v s_ktd[DVMVS_MAX_MEASUREMENTS * DP];
  __shared__ int s_next[2];   // next task of this workgroup: group (or -1), found by thread 0
  if (threadIdx.x == 0) s_next[1] = 0;
  unsigned int* const steal = steal_raw;   // (generic pointers: the atomics have no overloads for address-space-qualified ones)

  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  const int chunks = (a.D + DP - 1) / DP;
  const SweepWork work = decode_work<Cfg::XCD>(blockIdx.x, tiles_x * tiles_y, chunks, a.B);
  if (!work.valid) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int per_b = tiles_x * tiles_y * chunks;
  const int total_groups = per_b * a.B;
  unsigned int* const steal_list = steal + kStealHeaderWords;
  unsigned int* const steal_flag = steal + kStealHeaderWords + total_groups;
  // ---- task loop: the workgroup's own unit (all planes, or the lower half if it turns out to need several runs), then the
  // upper half of its own unit if nobody claimed it, then upper halves other workgroups published ----
  int task_group = work.group, p_lo = 0, p_hi = DP;
  bool own_upper_pending = false;   // this workgroup published the upper half of its unit and has not yet tried to claim it back
  int cur_b = -1;
  ADAPTIVE_TL(const unsigned long long tx_start = __builtin_amdgcn_s_memrealtime(); unsigned long long tx_first = 0, tx_publish = 0;)
  ADAPTIVE_TL(int tx_tasks = 0, tx_first_runs = 0, tx_published = 0;)
  bool light = false;   // this workgroup's own unit was light: it may claim published halves
  for (;;) {
  const int b = task_group / per_b;
  const int rem_g = task_group - b * per_b;
  const int tile = rem_g / chunks, chunk = chunks - 1 - (rem_g - tile * chunks);   // as decode_work numbers them
  const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
  const int d_block = chunk * DP;
  const int planes = min(DP, a.D - d_block);
  p_hi = min(p_hi, planes);

  // ---- per-workgroup tables: Hm = K R K^-1 and K t per measurement frame (fp64 on the first M lanes), K t / depth per plane ----
  if (b != cur_b) {   // (one batch item = one set of matrices for every task)
    if (tid < a.M) sweep_matrices(a.pose1 + b * 16, a.pose2[tid] + b * 16, a.K + b * 9, s_H + tid * 9, s_kt + tid * 3);
    cur_b = b;
  }
  __syncthreads();
  for (int i = tid; i < a.M * DP; i += NT) {
    const int m = i / DP, j = i - m * DP;
    float4v k = {0.0f, 0.0f, 0.0f, 0.0f};
    if (j < planes) {
      const float depth = plane_depth(a.inv_depth_base, a.inv_depth_step, d_block + j);
      k.x = s_kt[m * 3 + 0] / depth;
      k.y = s_kt[m * 3 + 1] / depth;
      k.z = s_kt[m * 3 + 2] / depth;
    }
    s_ktd[i] = k;
  }
  __syncthreads();
  const int HW = a.H * a.W;
  // lane -> pixel: each 16-lane ds_read_b128 service group owns 16 consecutive pixels of one tile row (see sweep_lane_pixel)
  const int lane_pixel = sweep_lane_pixel(tid & 31);
  const int x = tile_x * TW + (TW == 32 ? lane_pixel : TW == 16 ? (lane_pixel & 15) : tid % TW);
  const int y = tile_y * TH + (TW == 32 ? tid / 32 : TW == 16 ? (tid >> 5) * 2 + (lane_pixel >> 4) : tid / TW);
  const bool live = x < a.W && y < a.H;
  const float xf = static_cast<float>(x), yf = static_cast<float>(y);
  const int pix = live ? y * a.W + x : 0;
  gcfloat_p ref = as_global(a.image1) + static_cast<size_t>(b) * a.C * HW + pix;
  const SweepScale sc = sweep_scale(a.W, a.H);
  const char* tile_bytes = reinterpret_cast<const char*>(s_tile);
  const unsigned int plane_bytes = static_cast<unsigned int>(HW) * 4u;
  const unsigned int map_bytes = static_cast<unsigned int>(a.C) * plane_bytes;
  const __amdgpu_buffer_rsrc_t ref_rsrc = map_resource(as_global(a.image1) + static_cast<size_t>(b) * a.C * HW, map_bytes);
  const unsigned int ref_voffset = static_cast<unsigned int>(pix) * 4u;

  guint_p slot = nullptr;   // this workgroup's spill slot
  if (!GATHER) {
    const size_t groups = static_cast<size_t>(tiles_x) * tiles_y * chunks * a.B;
    slot = as_global(a.spill) + kSpillHeaderWords + groups + static_cast<size_t>(task_group) * spill_slot_words(a.M, DP);
  }
  int violated = 0;       // this thread saw a tap outside its staged box (round-off beyond the slack: not expected)
  int staged_runs = 0;    // (differs) runs this task staged: a task with fewer than M of them was light

  // even- and odd-channel partial sums of sum_m sum_c ref[c] * warped_m[c]: one accumulator over all measurement frames
  // (sum over frames, then / C, then / M; for the usual power-of-two C this is bit-identical to the reference's
  // per-frame / C followed by the sum, otherwise it is one rounding closer to exact)
  float2v acc2[DP];
#pragma unroll
  for (int j = 0; j < DP; ++j) acc2[j] = float2v{0.0f, 0.0f};

  // (differs) positions of the NEXT frame, computed inside the tap blocks of the current frame's first pass, where the wave
  // otherwise sits out LDS round trips; pre_mask: planes for which pos_pre holds the coming frame's position
  float2v pos_pre[DP];
  unsigned int pre_mask = 0;
  for (int m = 0; m < a.M; ++m) {
    const float* Hm = s_H + m * 9;
    const float4v* ktd_m = s_ktd + m * DP;
    const bool has_next = m + 1 < a.M;
    const SweepRay ray_n = sweep_ray(s_H + (has_next ? m + 1 : m) * 9, xf, yf);
    const float4v* ktd_n = s_ktd + (has_next ? m + 1 : m) * DP;
    const unsigned int pre_mask_cur = pre_mask;   // what pos_pre holds for THIS frame
    pre_mask = 0;
    bool frame_first_staged = true;
    gcfloat_p meas = as_global(a.image2[m]) + static_cast<size_t>(b) * a.C * HW;
    const SweepRay ray = sweep_ray(Hm, xf, yf);
    // This thread's sample positions on the chunk's planes do not depend on the box: with NCHW maps they are computed once per
    // frame and kept (16 registers) for workgroups that need several runs; the channels-last instantiation has no registers to
    // spare and recomputes them per run.  Either way all DP planes are evaluated without branches, so that the DP chains of
    // dependent operations (three exact divisions each) sit in one basic block.
    constexpr bool kKeepPositions = !ADAPTIVE && !PREPOS;   // (the task loop's state / pos_pre take the 16 registers)
    float2v pos[DP];
    auto sample_positions = [&]() {
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const float4v kd = ktd_m[j];   // zeros beyond the last plane of a ragged chunk
        float ix, iy;
        sweep_sample(ray, kd.x, kd.y, kd.z, sc, &ix, &iy);
        pos[j] = float2v{ix, iy};
      }
    };
    if (kKeepPositions) sample_positions();
    int seg_hint = DP;   // planes per segment that fitted last time: parallax per plane is uniform along the sweep
    int seg_lo = p_lo;
    while (seg_lo < p_hi) {
      int seg_len = min(p_hi - seg_lo, seg_hint);
      const SampleBox box = wave_sample_box<TW, TH, CAP, Cfg::MINSEG>(a, Hm, ktd_m, tile_x, tile_y, seg_lo, seg_len, sc, lane, &seg_len);
      if (ADAPTIVE && m == 0 && seg_lo == 0 && p_lo == 0 && p_hi == DP && seg_len < DP && task_group == work.group && !own_upper_pending) {
        // the first fit of the own unit says "several runs": keep planes [0, DP/2) and publish [DP/2, DP) (all frames of them)
        p_hi = DP / 2;
        seg_len = min(seg_len, p_hi);
        own_upper_pending = true;
        ADAPTIVE_TL(tx_published = 1; tx_publish = __builtin_amdgcn_s_memrealtime();)
        if (tid == 0) {
          const unsigned int at = atomicAdd(steal, 1u);
          atomicExch(steal_list + at, static_cast<unsigned int>(task_group) + 1u);
        }
      }
      seg_hint = max(seg_len, Cfg::MINSEG);
      const int seg_hi = seg_lo + seg_len;

      if (box.state == 1) {
        ++staged_runs;
        const int P = box.pitch, RS = box.pitch * box.RH;
        const int row_bytes = P * REC * 4;
        // ---- this thread's taps: byte address of the north-west record and the fractional position (ix - floor, iy - floor),
        // per plane of the run; the four bilinear weights are re-formed from the fractions in every channel pass (16 registers
        // instead of 32: what keeps this kernel at 3 waves per SIMD without scratch traffic) ----
        int addr[DP];
        float2v frac[DP];
        const unsigned int run_mask = ((1u << seg_len) - 1u) << seg_lo;
        const bool use_pre = PREPOS && frame_first_staged && (pre_mask_cur & run_mask) == run_mask;
        const bool fill_next = PREPOS && frame_first_staged && has_next;
        frame_first_staged = false;
        if (PREPOS) {
          if (!use_pre) {   // one array: filled here or during the previous frame's first pass, consumed below, refilled in pass 0
#pragma unroll
            for (int j = 0; j < DP; ++j) {
              const float4v kd = ktd_m[j];
              float ix, iy;
              sweep_sample(ray, kd.x, kd.y, kd.z, sc, &ix, &iy);
              pos_pre[j] = float2v{ix, iy};
            }
          }
        } else if (!kKeepPositions) {
          sample_positions();
        }
#pragma unroll
        for (int j = 0; j < DP; ++j) {   // all planes (those outside the run get addresses that are never used)
          const bool in_run = j >= seg_lo && j < seg_hi;   // workgroup-uniform
          const float ix = PREPOS ? pos_pre[j].x : pos[j].x, iy = PREPOS ? pos_pre[j].y : pos[j].y;
          const float fx = floorf(ix), fy = floorf(iy);
          int rx = static_cast<int>(fx) - box.x_lo, ry = static_cast<int>(fy) - box.y_lo;
          if (live && in_run) violated |= (static_cast<unsigned int>(rx) > static_cast<unsigned int>(box.RW - 2)) |
                                          (static_cast<unsigned int>(ry) > static_cast<unsigned int>(box.RH - 2));
          rx = min(max(rx, 0), box.RW - 2);
          ry = min(max(ry, 0), box.RH - 2);
          addr[j] = __mul24(__mul24(ry, P) + rx, REC * 4);   // full-rate 24-bit multiplies: ry, P, rx < 2^11
          frac[j] = float2v{ix - fx, iy - fy};
        }
        // ---- staging plan: byte offset into the measurement map of each of this thread's LDS pieces ----
        // NHWC: a piece is one 16-byte channel quad of one box position; NCHW: a piece is one box position (CCH dword loads).
        // Positions outside the image (zero apron), pad columns and pieces past the box get kBufferOutOfRange: the load
        // then returns zeros by itself.
        constexpr int kPieces = NHWC ? (CAP * QPR + NT - 1) / NT : (CAP + NT - 1) / NT;
        const unsigned int magic = 0xffffffffu / static_cast<unsigned int>(P) + 1u;   // r / P == mulhi(r, magic) for r < 2^16
        const int n_pieces = NHWC ? RS * QPR : RS;
        unsigned int goff[kPieces];
#pragma unroll
        for (int k = 0; k < kPieces; ++k) {
          const int piece = tid + k * NT;
          const int r = NHWC ? piece / QPR : piece;
          const int ry = static_cast<int>(__umulhi(static_cast<unsigned int>(r), magic));
          const int rx = r - ry * P;
          const int gx = box.x_lo + rx, gy = box.y_lo + ry;
          const bool in = (piece < n_pieces) && (rx < box.RW) && (gx >= 0) && (gx < a.W) && (gy >= 0) && (gy < a.H);
          goff[k] = in ? static_cast<unsigned int>(NHWC ? (gy * a.W + gx) * a.C + (piece % QPR) * 4 : gy * a.W + gx) * 4u : kBufferOutOfRange;
        }
        const __amdgpu_buffer_rsrc_t meas_rsrc = map_resource(meas, map_bytes);

        // ---- channel passes.  All workgroups of a frame are resident at once, so the launch lasts about as long as one
        // workgroup's dependency chain, and the global-load round trip of every pass sits on it (load -> wait -> ds_write ->
        // barrier -> taps -> barrier).  PREFETCH (a tuning option, off) requests the first kPre pieces of the NEXT pass and its
        // reference features before the taps of the current pass and holds them in registers until the buffer is free.
        // Measured on MI355X: 39.6 us against 36.0 us without (sideways pair), 57 against 52 (index line 117): the 16-24
        // extra live registers push the kernel past 168 VGPRs (scratch traffic), which costs more than the overlap gains.
        constexpr int kPre = Cfg::PREFETCH ? (NHWC ? (kPieces < 4 ? kPieces : 4) : (kPieces < 2 ? kPieces : 2)) : 0;
        constexpr int kPreRegs = NHWC ? 1 : QPR;   // float4 per piece
        float4v pre[kPre > 0 ? kPre * kPreRegs : 1];
        float2v rv_next[CCH / 2];
        auto load_ref = [&](int c0, float2v* rv) {
          // channels beyond C (last pass of a ragged channel count) re-read channel C-1 on both sides and are cancelled by rv = 0
#pragma unroll
          for (int c = 0; c < CCH; ++c) {
            const float v = buffer_f32(ref_rsrc, ref_voffset, static_cast<unsigned int>(min(c0 + c, a.C - 1)) * plane_bytes);
            rv[c / 2][c % 2] = (c0 + c < a.C) ? v : 0.0f;
          }
        };
        auto load_piece = [&](int k, int c0, float4v* v) {   // piece k of pass c0 into kPreRegs float4
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
        auto store_piece = [&](int k, const float4v* v) {
          const int piece = tid + k * NT;
          if (piece < n_pieces) {
            if (NHWC) {
              *reinterpret_cast<float4v*>(s_tile + (piece / QPR) * REC + (piece % QPR) * 4) = v[0];
            } else {
#pragma unroll
              for (int q = 0; q < QPR; ++q) *reinterpret_cast<float4v*>(s_tile + piece * REC + q * 4) = v[q];
            }
          }
        };
        if (kPre > 0) {
          load_ref(0, rv_next);
#pragma unroll
          for (int k = 0; k < kPre; ++k)
            if (k * NT < n_pieces) load_piece(k, 0, pre + k * kPreRegs);   // workgroup-uniform
        }
        for (int c0 = 0; c0 < a.C; c0 += CCH) {
          const bool fill = fill_next && c0 == 0;   // workgroup-uniform
          float2v rv[CCH / 2];
          if (kPre > 0) {
#pragma unroll
            for (int c = 0; c < CCH / 2; ++c) rv[c] = rv_next[c];
          } else {
            load_ref(c0, rv);   // first, so that its latency overlaps the copy
          }
          // pieces that were not prefetched: load now, a few in flight at a time
          constexpr int kBatch = Cfg::BATCH > 0 ? Cfg::BATCH : (NHWC ? 4 : 1);
#pragma unroll
          for (int k0 = kPre; k0 < kPieces; k0 += kBatch) {
            if (k0 * NT < n_pieces) {   // workgroup-uniform
              float4v v[kBatch * kPreRegs];
#pragma unroll
              for (int kk = 0; kk < kBatch; ++kk)
                load_piece(k0 + kk < kPieces ? k0 + kk : kPieces - 1, c0, v + kk * kPreRegs);
#pragma unroll
              for (int kk = 0; kk < kBatch; ++kk)
                if (k0 + kk < kPieces) store_piece(k0 + kk, v + kk * kPreRegs);
            }
          }
#pragma unroll
          for (int k = 0; k < kPre; ++k)
            if (k * NT < n_pieces) store_piece(k, pre + k * kPreRegs);
          __syncthreads();
          if (kPre > 0 && c0 + CCH < a.C) {   // the next pass's requests go out before the taps of this one
            load_ref(c0 + CCH, rv_next);
#pragma unroll
            for (int k = 0; k < kPre; ++k)
              if (k * NT < n_pieces) load_piece(k, c0 + CCH, pre + k * kPreRegs);
          }
#pragma unroll
          for (int j = 0; j < DP; ++j)
            if (j >= seg_lo && j < seg_hi) {   // workgroup-uniform
              tap_plane<QPR, REC>(tile_bytes, row_bytes, addr[j], frac[j], rv, &acc2[j]);
              if (fill) {   // VALU work moved into the LDS-bound half of the pass
                const float4v kd = ktd_n[j];
                float nx, ny, kz = kd.z;
                asm volatile("" : "+v"(kz));   // opaque: the computation is loop-invariant and would be hoisted out of the pass loop
                sweep_sample(ray_n, kd.x, kd.y, kz, sc, &nx, &ny);
                pos_pre[j] = float2v{nx, ny};
              }
            }
          __syncthreads();
        }
        if (fill_next) pre_mask |= run_mask;
      } else if (box.state == 0 && !GATHER) {
        // cannot be staged: queue the run for the second pass.  The two halves of a unit may both append to the unit's slot
        // (their planes are disjoint, so the per-plane order of the items stays the frame order); the first item registers
        // the group, and every appender writes the group's constants (identical values).
        if (tid < a.M * 12) {
          const int mm = tid / 12, kk = tid - mm * 12;
          slot[1 + a.M * DP + tid] = __float_as_uint(kk < 9 ? s_H[mm * 9 + kk] : s_kt[mm * 3 + (kk - 9)]);
        }
        if (tid == 0) {
          const unsigned int it = atomicAdd(a.spill + (slot - as_global(a.spill)), 1u);
          slot[1 + it] = spill_pack(m, seg_lo, seg_len);
          if (it == 0) {
            const unsigned int at = atomicAdd(a.spill, 1u);
            as_global(a.spill)[kSpillHeaderWords + at] = static_cast<unsigned int>(task_group);
          }
        }
      } else if (box.state == 0) {
        const float* kt = s_kt + m * 3;
#pragma unroll
        for (int j = 0; j < DP; ++j)
          if (j >= seg_lo && j < seg_hi) {
            const float depth = plane_depth(a.inv_depth_base, a.inv_depth_step, d_block + j);
            acc2[j].x += live ? gather_plane<NHWC>(a, meas, ref, HW, ray, kt[0] / depth, kt[1] / depth, kt[2] / depth, sc) : 0.0f;
          }
      }
      // state 2: the whole footprint of the run lies outside the image -> zeros
      seg_lo = seg_hi;
    }
  }

  // (experiment: a tap outside its staged box is only reported, through the error word)
  if (__syncthreads_or(violated) && tid == 0) atomicExch(steal + 2, 1u);

  if (live) {
    gfloat_p out = as_global(a.out) + (static_cast<size_t>(b) * a.D + d_block) * HW + pix;
    const float Cf = static_cast<float>(a.C), Mf = static_cast<float>(a.M);
    if (((a.C & (a.C - 1)) | (a.M & (a.M - 1))) == 0) {
      // both counts are powers of two (the usual 32 channels, 1 or 2 frames): x * 2^-k is x / 2^k, correctly rounded either
      // way, without the 2 x 10 instructions of an IEEE division per output
      const float rC = 1.0f / Cf, rM = 1.0f / Mf;
#pragma unroll
      for (int j = 0; j < DP; ++j)
        if (j >= p_lo && j < p_hi) out[static_cast<size_t>(j) * HW] = ((acc2[j].x + acc2[j].y) * rC) * rM;
    } else {
#pragma unroll
      for (int j = 0; j < DP; ++j)
        if (j >= p_lo && j < p_hi) out[static_cast<size_t>(j) * HW] = ((acc2[j].x + acc2[j].y) / Cf) / Mf;
    }
  }
  if (!ADAPTIVE) break;
  // ---- next task ----
  ADAPTIVE_TL(++tx_tasks; if (tx_tasks == 1) { tx_first = __builtin_amdgcn_s_memrealtime(); tx_first_runs = staged_runs; })
  ADAPTIVE_TL(if (tid == 0 && p_lo != 0 && task_group < 8192) g_adaptive_half[task_group * 4 + 2] = __builtin_amdgcn_s_memrealtime();)
  __syncthreads();   // every thread is done with s_ktd and the tile before they are rebuilt
  if (task_group == work.group && p_lo == 0) light = staged_runs < a.M;   // judged on the workgroup's own unit, kept afterwards
  if (tid == 0) {
    int next = -1;
    if (own_upper_pending && atomicExch(steal_flag + work.group, 1u) == 0u) {   // nobody took it: do it here
      next = work.group;
      ADAPTIVE_TL(if (next < 8192) { g_adaptive_half[next * 4 + 0] = __builtin_amdgcn_s_memrealtime(); g_adaptive_half[next * 4 + 1] = 1; })
    }
    if (next < 0 && light && !(claim_once && s_next[1])) {
      int stolen = 0;
      // claim upper halves other workgroups published, in publication order.  Only workgroups whose own task was light look
      // (the ones that end last must not pay a round trip to memory for nothing), and they look with plain device-scope
      // loads first: read-modify-writes on one word from 640 workgroups serialise (measured: +17 us per launch).
      for (;;) {
        const unsigned int published = __hip_atomic_load(steal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_load(steal + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= published) break;
        const unsigned int i = atomicAdd(steal + 1, 1u);
        if (i >= published) break;   // nothing (more) published at this moment: owners look after later ones
        unsigned int entry = 0;
        for (int spin = 0; spin < 100000 && entry == 0u; ++spin) entry = __hip_atomic_load(steal_list + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (entry == 0u) break;
        if (atomicExch(steal_flag + (entry - 1u), 1u) == 0u) {
          next = static_cast<int>(entry - 1u);
          stolen = 1;
          ADAPTIVE_TL(if (next < 8192) { g_adaptive_half[next * 4 + 0] = __builtin_amdgcn_s_memrealtime(); g_adaptive_half[next * 4 + 1] = 2; })
          break;
        }
      }
      if (stolen) s_next[1] = 1;
    }
    s_next[0] = next;
  }
  own_upper_pending = false;
  __syncthreads();
  const int next = s_next[0];
  if (next < 0) {
#ifdef DVMVS_ADAPTIVE_TIMELINE
    if (tid == 0 && work.group < 8192) {
      unsigned long long* t = g_adaptive_wg + static_cast<size_t>(work.group) * 8;
      t[0] = tx_start; t[1] = tx_first; t[2] = __builtin_amdgcn_s_memrealtime(); t[3] = static_cast<unsigned long long>(tx_tasks);
      t[4] = static_cast<unsigned long long>(tx_first_runs); t[5] = tx_publish; t[6] = static_cast<unsigned long long>(tx_published) | (light ? 2ull : 0ull);
    }
#endif
    break;
  }
  task_group = next;
  p_lo = DP / 2;
  p_hi = DP;
  }   // task loop
}

}  // namespace dvmvs
// bit 0: leave out the second pass; bit 1: launch the shipped first-pass kernel instead of the adaptive one (so that the two
// first passes can be timed alone, side by side); bit 2: a workgroup claims at most one published half; bit 3 (8): the kernel
// with PREPOS instead of the adaptive one; bit 4 (16): the plain kernel (shipped first pass, no trace instrumentation)
extern "C" int dvmvs_debug_mode = 0;
extern "C" int dvmvs_debug_adaptive_trace(unsigned long long* wg, unsigned long long* half, int groups) {
  if (groups > 8192) groups = 8192;
  if (hipMemcpyFromSymbol(wg, HIP_SYMBOL(dvmvs::g_adaptive_wg), sizeof(unsigned long long) * 8 * groups) != hipSuccess) return -1;
  return hipMemcpyFromSymbol(half, HIP_SYMBOL(dvmvs::g_adaptive_half), sizeof(unsigned long long) * 4 * groups) == hipSuccess ? 0 : -1;
}
namespace dvmvs {

template <class Cfg>
int launch_sweep_adaptive(const CostVolumeArgs& a, unsigned int* steal, hipStream_t stream) {
  const long long tiles = static_cast<long long>((a.W + Cfg::TW - 1) / Cfg::TW) * ((a.H + Cfg::TH - 1) / Cfg::TH);
  const long long total = tiles * ((a.D + Cfg::DP - 1) / Cfg::DP) * a.B;
  const unsigned int grid = static_cast<unsigned int>((total + 7) / 8 * 8);
  static bool configured[kMaxDevices] = {};
  auto kernel = sweep_adaptive_kernel<Cfg, true, false>;
  const int rc = raise_dynamic_lds_limit(kernel, Cfg::kLdsBytes, configured);
  if (rc != 0) return rc;
  if (dvmvs_debug_mode & 2) {
    static bool configured2[kMaxDevices] = {};
    auto shipped = sweep_tiled_kernel<Cfg, false, false>;
    const int rc3 = raise_dynamic_lds_limit(shipped, Cfg::kLdsBytes, configured2);
    if (rc3 != 0) return rc3;
    hipLaunchKernelGGL(shipped, dim3(grid), dim3(Cfg::NT), Cfg::kLdsBytes, stream, a);
  } else if (dvmvs_debug_mode & 24) {
    static bool configured3[kMaxDevices] = {}, configured4[kMaxDevices] = {};
    auto prepos = sweep_adaptive_kernel<Cfg, false, true>;
    auto plain = sweep_adaptive_kernel<Cfg, false, false>;
    int rc4 = raise_dynamic_lds_limit(prepos, Cfg::kLdsBytes, configured3);
    if (rc4 == 0) rc4 = raise_dynamic_lds_limit(plain, Cfg::kLdsBytes, configured4);
    if (rc4 != 0) return rc4;
    hipLaunchKernelGGL((dvmvs_debug_mode & 8) ? prepos : plain, dim3(grid), dim3(Cfg::NT), Cfg::kLdsBytes, stream, a, steal, 0);
  } else {
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(Cfg::NT), Cfg::kLdsBytes, stream, a, steal, (dvmvs_debug_mode & 4) ? 1 : 0);
  }
  const int rc2 = launch_status();
  if (rc2 != 0) return rc2;
  if (dvmvs_debug_mode & 1) return 0;
  hipLaunchKernelGGL((sweep_spill_kernel<Cfg, false>), dim3(kSpillGrid), dim3(Cfg::NT), 0, stream, a);
  return launch_status();
}

}  // namespace dvmvs

// Same arguments as dvmvs_cost_volume_fwd (NCHW, dot product), plus the claim list: (16 + 2 * groups) words.  Both scratch
// buffers are cleared here, on the stream, before the launch (the experiment does not keep the "leave it zeroed" contract).
extern "C" int dvmvs_debug_sweep_adaptive(const float* image1, const float* const* image2s, const float* pose1, const float* const* pose2s,
                                          const float* K, float* cost_volume, int B, int M, int C, int H, int W, int D, double min_depth,
                                          double max_depth, float* workspace, size_t workspace_bytes, unsigned int* steal, size_t steal_bytes,
                                          dvmvs_stream_t stream) {
  using namespace dvmvs;
  CostVolumeArgs a;
  const int rc = fill_sweep_args(&a, image1, image2s, pose1, pose2s, K, cost_volume, B, M, C, H, W, D, min_depth, max_depth, true);
  if (rc != 0) return rc;
  a.image2_nhwc = 0;
  a.spill = reinterpret_cast<unsigned int*>(workspace);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t groups = static_cast<size_t>((W + 31) / 32) * ((H + 7) / 8) * ((D + 7) / 8) * B;
  if (workspace == nullptr || steal == nullptr || steal_bytes < sizeof(unsigned int) * (kStealHeaderWords + 2 * groups)) return DVMVS_EINVAL;
  if (!(dvmvs_debug_mode & 32)) {   // bit 5 (32): the caller vouches for zeroed scratch (geometries where nothing spills)
    if (hipMemsetAsync(workspace, 0, workspace_bytes, s) != hipSuccess) return DVMVS_EINVAL;
    if (hipMemsetAsync(steal, 0, steal_bytes, s) != hipSuccess) return DVMVS_EINVAL;
  }
  return launch_sweep_adaptive<SweepDefault>(a, steal, s);
}
