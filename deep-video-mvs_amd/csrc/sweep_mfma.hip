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

template <int CTRL>
__device__ inline int dpp_exchange(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }

// maximum over the wave (returned wave-uniform): butterflies inside each row of 16 lanes on the VALU's DPP path, then four readlanes
__device__ inline int wave_max_i32(int v) {
  v = max(v, dpp_exchange<0xB1>(v));    // quad_perm [1,0,3,2]
  v = max(v, dpp_exchange<0x4E>(v));    // quad_perm [2,3,0,1]
  v = max(v, dpp_exchange<0x141>(v));   // row_half_mirror
  v = max(v, dpp_exchange<0x140>(v));   // row_mirror
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// Lanes of ONE wave exchange data through its private LDS slice.  The hardware executes a wave's LDS operations in order, so no
// s_barrier and no s_waitcnt is needed between a write and another lane's read; what is needed is that the COMPILER keeps them in
// program order: a wave-scope fence plus a compiler-level memory clobber.
__device__ inline void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// GW x GH = 16 reference pixels per wave; CAP = cells of the dot table; NW = waves per workgroup (x-adjacent groups of the same plane
// chunk: they share the CU's L1, not LDS); WAVES = waves per SIMD the register allocation is held to; PREFETCH = the operands of the
// next two tiles are requested before the MFMAs of the current two.
template <int GW_, int GH_, int CAP_, int NW_, int WAVES_, bool PREFETCH_, int TPI_ = 1, int ABLATE_ = 0>
struct MfmaSweepConfig {
  static constexpr int GW = GW_, GH = GH_, CAP = CAP_, NW = NW_, WAVES = WAVES_;
  static constexpr bool PREFETCH = PREFETCH_;
  static constexpr int TPI = TPI_;                     // tiles per operand request / MFMA block (2 = two accumulators interleaved)
  static constexpr int ABLATE = ABLATE_;               // tools only (timing experiments, wrong results): 1 no operand loads, 2 no MFMAs, 4 no interpolation
  static constexpr int PW = 16;                        // planes per wave
  static constexpr int SPLIT = 2;                      // a 16-plane box of more than SPLIT * CAP cells is redone per 4 planes (tools/sweep_mfma_model.py)
  static constexpr int PITCH = CAP + 16 + 4;           // floats per pixel row of the table (16-byte aligned rows; the odd tile of a pair may be written)
  static constexpr size_t kLdsBytes = sizeof(float) * NW * (16 * PITCH + 4 * PW);
  static_assert(GW * GH == 16 && CAP % 16 == 0 && NW >= 1 && NW <= 4, "group shape");
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

// FULL: exactly 32 channels (the hot-path shape: no per-channel bounds tests); otherwise any C <= 32 (NHWC: a multiple of 4).
template <class Cfg, bool NHWC, bool FULL>
__global__ __launch_bounds__(64 * Cfg::NW, Cfg::WAVES) void sweep_mfma_kernel(CostVolumeArgs a) {
  constexpr int GW = Cfg::GW, GH = Cfg::GH, PW = Cfg::PW, CAP = Cfg::CAP, NW = Cfg::NW, PITCH = Cfg::PITCH;
  const int C = FULL ? kMfmaSweepChannels : a.C;
  extern __shared__ __attribute__((aligned(16))) float s_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;
  float* T = s_lds + wave * (16 * PITCH + 4 * PW);                               // [16 pixels][PITCH]: <f1(pixel), f2(cell)>
  float4v* ktd = reinterpret_cast<float4v*>(s_lds + wave * (16 * PITCH + 4 * PW) + 16 * PITCH);   // [PW] K t / depth of the frame in work

  // ---- work item: NW x-adjacent groups x one chunk of PW planes; XCD k (= blockIdx % 8) gets a contiguous range of image rows, so a
  // measurement footprint is fetched into one L2 ----
  const int groups_x = (a.W + GW - 1) / GW, groups_y = (a.H + GH - 1) / GH;
  const int gblocks_x = (groups_x + NW - 1) / NW;
  const int chunks = (a.D + PW - 1) / PW;
  const int per_b = groups_y * gblocks_x * chunks, total = per_b * a.B;
  const int per_xcd = (total + 7) / 8;
  const int item = static_cast<int>(blockIdx.x & 7) * per_xcd + static_cast<int>(blockIdx.x >> 3);
  if (item >= total) return;
  const int b = item / per_b;
  int rem = item - b * per_b;
  const int chunk = rem % chunks;
  rem /= chunks;
  const int gbx = rem % gblocks_x, gy = rem / gblocks_x;
  const int gx = gbx * NW + wave;
  if (gx >= groups_x) return;   // (waves are independent of each other: no workgroup barrier anywhere below)
  const int d_block = chunk * PW;
  MFMA_TRACE(const unsigned long long tr_start = __builtin_amdgcn_s_memtime(), tr_real0 = __builtin_amdgcn_s_memrealtime();)
  MFMA_TRACE(unsigned long long tr_pos = 0, tr_box = 0, tr_tiles = 0, tr_look = 0, tr_ntiles = 0, tr_nstrips = 0, tr_first = 0;)

  const int HW = a.H * a.W;
  const int x = gx * GW + p % GW, y = gy * GH + p / GW;
  const bool live = x < a.W && y < a.H;
  const int pix = live ? y * a.W + x : 0;
  const float xf = static_cast<float>(x), yf = static_cast<float>(y);
  const SweepScale sc = sweep_scale(a.W, a.H);
  const unsigned int plane_bytes = static_cast<unsigned int>(HW) * 4u;
  const unsigned int map_bytes = static_cast<unsigned int>(C) * plane_bytes;

  // B operand of every MFMA of this wave: lane (p, q) holds f1[8 q + jj][pixel p], jj = 0..7
  gcfloat_p ref = as_global(a.image1) + static_cast<size_t>(b) * C * HW + pix;
  float f1v[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const float v = ref[static_cast<size_t>(FULL ? 8 * q + jj : min(8 * q + jj, C - 1)) * HW];
    f1v[jj] = (FULL || 8 * q + jj < C) ? v : 0.0f;
  }

  // depth of plane d_block + (lane & 15) as the reference's python-double expression (utils.py:59-66)
  const int d_lane = d_block + p;
  const float depth_lane = plane_depth(a.inv_depth_base, a.inv_depth_step, min(d_lane, a.D - 1));

  gcfloat_p Hm_g = as_global(a.Hm) + static_cast<size_t>(b) * a.M * 9;
  gcfloat_p kt_g = as_global(a.kt) + static_cast<size_t>(b) * a.M * 3;

  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // sum over frames and taps of w <f1, f2>; plane d_block + 4 j + q

  for (int m = 0; m < a.M; ++m) {
    MFMA_TRACE(const unsigned long long tr_f0 = __builtin_amdgcn_s_memtime(); if (m == 0) tr_first = tr_f0;)
    // ---- K t / depth of the chunk's planes for this frame (utils.py:66-68: IEEE fp32 division by the fp32-rounded depth) ----
    if (lane < PW) {
      float4v k = {0.0f, 0.0f, 0.0f, 0.0f};
      if (d_lane < a.D) {
        k.x = kt_g[m * 3 + 0] / depth_lane;
        k.y = kt_g[m * 3 + 1] / depth_lane;
        k.z = kt_g[m * 3 + 2] / depth_lane;
      }
      ktd[lane] = k;
    }
    wave_lds_fence();
    float Hm[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Hm[k] = Hm_g[m * 9 + k];
    const SweepRay ray = sweep_ray(Hm, xf, yf);
    const __amdgpu_buffer_rsrc_t meas_rsrc = map_resource(as_global(a.image2[m]) + static_cast<size_t>(b) * C * HW, map_bytes);

    // ---- this lane's four samples: north-west tap, fractional position, alive = some tap can be inside the image ----
    int x0s[4], y0s[4];
    float frx[4], fry[4];
    unsigned int alive = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4v kd = ktd[4 * j + q];
      float ix, iy;
      sweep_sample(ray, kd.x, kd.y, kd.z, sc, &ix, &iy);   // clamped to [-1, W] x [-1, H]; NaN -> -1
      const bool al = live && (d_block + 4 * j + q < a.D) && (ix > -1.0f) && (ix < sc.Wf) && (iy > -1.0f) && (iy < sc.Hf);
      const float fx = floorf(ix), fy = floorf(iy);
      x0s[j] = static_cast<int>(fx);
      y0s[j] = static_cast<int>(fy);
      frx[j] = ix - fx;
      fry[j] = iy - fy;
      alive |= al ? (1u << j) : 0u;
    }

    MFMA_TRACE(asm volatile("s_nop 0" :: "v"(x0s[0]), "v"(x0s[1]), "v"(x0s[2]), "v"(x0s[3]));)   // (the positions are complete here)
    MFMA_TRACE(tr_pos += __builtin_amdgcn_s_memtime() - tr_f0;)
    // ---- passes: the 16 planes as one box; a box of more than SPLIT x CAP cells (diagonal or fast epipolar motion: the box is mostly
    // empty) is redone per 4 planes.  A box is processed in STRIPS of CAP cells of its row-major index space: one strip almost always;
    // several under strong magnification, where a tap simply belongs to the strip that holds its cell -- no footprint is too large ----
    int n_pass = 1;
    for (int s = 0; s < n_pass; ++s) {
      const int jlo = n_pass == 1 ? 0 : s, jhi = n_pass == 1 ? 4 : s + 1;
      MFMA_TRACE(const unsigned long long tr_b0 = __builtin_amdgcn_s_memtime();)
      // bounding box of the in-image taps of the alive samples (clipped to the image: taps outside it get weight 0 below)
      int nlx = -100000, hx = -100000, nly = -100000, hy = -100000;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j >= jlo && j < jhi && ((alive >> j) & 1u)) {
          nlx = max(nlx, -max(x0s[j], 0));
          hx = max(hx, min(x0s[j] + 1, a.W - 1));
          nly = max(nly, -max(y0s[j], 0));
          hy = max(hy, min(y0s[j] + 1, a.H - 1));
        }
      const int x_hi = wave_max_i32(hx);
      if (x_hi < 0) continue;   // no alive sample in this pass: nothing to add
      const int x_lo = -wave_max_i32(nlx), y_lo = -wave_max_i32(nly), y_hi = wave_max_i32(hy);
      const int bw = x_hi - x_lo + 1, bh = y_hi - y_lo + 1, cells = bw * bh;
      if (n_pass == 1 && cells > Cfg::SPLIT * CAP) {
        n_pass = 4;
        s = -1;
        continue;
      }
      const float rcp_bw = 1.0f / static_cast<float>(bw);
      MFMA_TRACE(tr_box += __builtin_amdgcn_s_memtime() - tr_b0;)

      for (int base = 0; base < cells; base += CAP) {
        const int n = min(CAP, cells - base);
        MFMA_TRACE(const unsigned long long tr_t0 = __builtin_amdgcn_s_memtime();)
        // ---- dot table of the strip: cells 16 at a time through the matrix core ----
        const int ntiles = (n + 15) >> 4;
        constexpr int TPI = Cfg::TPI;
        auto issue = [&](float (&A)[TPI][8], int t0) {   // operand requests of tiles t0 .. t0 + TPI - 1 (beyond the strip: no traffic, zeros)
#pragma unroll
          for (int u = 0; u < TPI; ++u) {
            const int li = (t0 + u) * 16 + p;
            // (row, column) of box cell base + li: float quotient (exact operands below 2^24: a map has fewer cells), then one step of
            // correction either way
            const int idx = base + li;
            int r = static_cast<int>(static_cast<float>(idx) * rcp_bw);
            int c = idx - r * bw;
            if (c < 0) { c += bw; --r; }
            if (c >= bw) { c -= bw; ++r; }
            const unsigned int cell = static_cast<unsigned int>((y_lo + r) * a.W + x_lo + c);
            const bool ok = li < n;
            if (Cfg::ABLATE & 1) {
#pragma unroll
              for (int jj = 0; jj < 8; ++jj) A[u][jj] = static_cast<float>(cell + jj) * 1e-6f;
            } else if (NHWC) {
              const unsigned int vo = ok ? (cell * C + 8u * q) * 4u : kBufferOutOfRange;
              const float4v lo = buffer_f32x4(meas_rsrc, (FULL || 8 * q < C) ? vo : kBufferOutOfRange, 0u);
              const float4v hi = buffer_f32x4(meas_rsrc, (FULL || 8 * q + 4 < C) ? vo + 16u : kBufferOutOfRange, 0u);
              A[u][0] = lo.x; A[u][1] = lo.y; A[u][2] = lo.z; A[u][3] = lo.w;
              A[u][4] = hi.x; A[u][5] = hi.y; A[u][6] = hi.z; A[u][7] = hi.w;
            } else {
              const unsigned int vo = ok ? (8u * q * static_cast<unsigned int>(HW) + cell) * 4u : kBufferOutOfRange;
#pragma unroll
              for (int jj = 0; jj < 8; ++jj)
                A[u][jj] = buffer_f32(meas_rsrc, (FULL || 8 * q + jj < C) ? vo : kBufferOutOfRange, static_cast<unsigned int>(jj) * plane_bytes);
            }
          }
        };
        auto compute = [&](const float (&A)[TPI][8], int t0) {   // D[cell][pixel]: this lane gets its own pixel p, cells 4 q .. 4 q + 3 of each tile
          float4v c[TPI];
#pragma unroll
          for (int u = 0; u < TPI; ++u) c[u] = float4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int jj = 0; jj < 8; ++jj)
#pragma unroll
            for (int u = 0; u < TPI; ++u) {
              if (Cfg::ABLATE & 2) c[u][jj & 3] += A[u][jj] * f1v[jj];
              else c[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[u][jj], f1v[jj], c[u], 0, 0, 0);
            }
#pragma unroll
          for (int u = 0; u < TPI; ++u)
            if (u == 0 || t0 + u < ntiles) *reinterpret_cast<float4v*>(T + p * PITCH + (t0 + u) * 16 + 4 * q) = c[u];
        };
        if (Cfg::PREFETCH) {
          float A0[TPI][8], A1[TPI][8];
          issue(A0, 0);
          for (int t = 0; t < ntiles; t += 2 * TPI) {
            issue(A1, t + TPI);
            compute(A0, t);
            issue(A0, t + 2 * TPI);
            if (t + TPI < ntiles) compute(A1, t + TPI);
          }
        } else {
          for (int t = 0; t < ntiles; t += TPI) {
            float A0[TPI][8];
            issue(A0, t);
            compute(A0, t);
          }
        }
        wave_lds_fence();
        MFMA_TRACE(const unsigned long long tr_t1 = __builtin_amdgcn_s_memtime(); tr_tiles += tr_t1 - tr_t0; tr_ntiles += ntiles; ++tr_nstrips;)

        // ---- interpolation: four table entries per sample.  Tap indices live in the box's row-major index space, clamped into the box (a
        // tap outside the image -- the only way to fall outside the box -- has weight 0: grid_sample's zeros padding); a tap whose cell
        // lies in another strip contributes there ----
        const float* Tp = T + p * PITCH;
        const int last = n - 1;
        const unsigned int un = static_cast<unsigned int>(n);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if ((Cfg::ABLATE & 4) && j >= jlo && j < jhi) {
            acc[j] += Tp[min(x0s[j] & 15, last)];
          } else if (j >= jlo && j < jhi) {
            const int xa = x0s[j], ya = y0s[j];
            const bool al = (alive >> j) & 1u;
            const bool in_w = al && xa >= 0, in_e = al && xa + 1 <= a.W - 1, in_n = ya >= 0, in_s = ya + 1 <= a.H - 1;
            const int rxa = min(max(xa - x_lo, 0), bw - 1), rxb = min(max(xa + 1 - x_lo, 0), bw - 1);
            const int rya = min(max(ya - y_lo, 0), bh - 1) * bw - base, ryb = min(max(ya + 1 - y_lo, 0), bh - 1) * bw - base;
            const int k_nw = rya + rxa, k_ne = rya + rxb, k_sw = ryb + rxa, k_se = ryb + rxb;
            const float t_nw = Tp[min(max(k_nw, 0), last)], t_ne = Tp[min(max(k_ne, 0), last)];
            const float t_sw = Tp[min(max(k_sw, 0), last)], t_se = Tp[min(max(k_se, 0), last)];
            float2v w_n, w_s;
            tap_weights(frx[j], fry[j], &w_n, &w_s);
            float f = acc[j];
            f = fmaf(t_nw, (in_w && in_n && static_cast<unsigned int>(k_nw) < un) ? w_n.x : 0.0f, f);
            f = fmaf(t_ne, (in_e && in_n && static_cast<unsigned int>(k_ne) < un) ? w_n.y : 0.0f, f);
            f = fmaf(t_sw, (in_w && in_s && static_cast<unsigned int>(k_sw) < un) ? w_s.x : 0.0f, f);
            f = fmaf(t_se, (in_e && in_s && static_cast<unsigned int>(k_se) < un) ? w_s.y : 0.0f, f);
            acc[j] = f;
          }
        wave_lds_fence();   // (the next strip / pass overwrites the table)
        MFMA_TRACE(asm volatile("s_nop 0" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3])); tr_look += __builtin_amdgcn_s_memtime() - tr_t1;)
      }
    }
  }

  // sum over frames, then / C, then / M (for power-of-two counts x * 2^-k is x / 2^k exactly: the reference's per-frame / C followed
  // by the mean over frames, bit for bit; otherwise one rounding closer to exact)
  if (live) {
    gfloat_p out = as_global(a.out) + (static_cast<size_t>(b) * a.D + d_block + q) * HW + pix;
    const float Cf = static_cast<float>(C), Mf = static_cast<float>(a.M);
    const bool pow2 = ((C & (C - 1)) | (a.M & (a.M - 1))) == 0;
    const float rC = 1.0f / Cf, rM = 1.0f / Mf;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (d_block + 4 * j + q < a.D) out[static_cast<size_t>(4 * j) * HW] = pow2 ? (acc[j] * rC) * rM : (acc[j] / Cf) / Mf;
  }
#ifdef DVMVS_SWEEP_TRACE
  {
    const int wid = item * NW + wave;
    if (lane == 0 && wid < kMfmaTraceWaves) {
      unsigned long long* t = g_sweep_mfma_trace + static_cast<size_t>(wid) * kMfmaTraceWords;
      t[0] = tr_start; t[1] = __builtin_amdgcn_s_memtime(); t[2] = tr_first - tr_start; t[3] = tr_pos; t[4] = tr_box; t[5] = tr_tiles; t[6] = tr_look;
      t[7] = tr_ntiles | (tr_nstrips << 32); t[8] = __builtin_amdgcn_s_getreg((31 << 11) | 4); t[9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      t[10] = tr_real0; t[11] = __builtin_amdgcn_s_memrealtime(); t[12] = blockIdx.x; t[13] = static_cast<unsigned long long>(chunk);
    }
  }
#endif
}

// ---- launch ------------------------------------------------------------------------------------------------------------
template <class Cfg, bool NHWC, bool FULL>
int launch_sweep_mfma_layout(const CostVolumeArgs& a, hipStream_t stream) {
  const long long groups_x = (a.W + Cfg::GW - 1) / Cfg::GW, groups_y = (a.H + Cfg::GH - 1) / Cfg::GH;
  const long long total = groups_y * ((groups_x + Cfg::NW - 1) / Cfg::NW) * ((a.D + Cfg::PW - 1) / Cfg::PW) * a.B;
  if (total > (1LL << 30)) return DVMVS_EUNSUPPORTED;
  const unsigned int grid = static_cast<unsigned int>((total + 7) / 8 * 8);
  auto kernel = sweep_mfma_kernel<Cfg, NHWC, FULL>;
  if (Cfg::kLdsBytes > 48 * 1024) {
    static bool configured[64] = {};
    int device = 0;
    DVMVS_RETURN_IF_HIP(hipGetDevice(&device));
    const bool tracked = device >= 0 && device < 64;
    if (!tracked || !configured[device]) {   // idempotent per-device function attribute; racing threads write the same value
      DVMVS_RETURN_IF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(Cfg::kLdsBytes)));
      if (tracked) configured[device] = true;
    }
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * Cfg::NW), Cfg::kLdsBytes, stream, a);
  return launch_status();
}

template <class Cfg>
int launch_sweep_mfma_cfg(const CostVolumeArgs& a, hipStream_t stream) {
  if (a.C == kMfmaSweepChannels)
    return a.image2_nhwc ? launch_sweep_mfma_layout<Cfg, true, true>(a, stream) : launch_sweep_mfma_layout<Cfg, false, true>(a, stream);
  return a.image2_nhwc ? launch_sweep_mfma_layout<Cfg, true, false>(a, stream) : launch_sweep_mfma_layout<Cfg, false, false>(a, stream);
}

// up to 32 channels (channels-last measurement maps: a multiple of 4), maps below 2 GiB (32-bit buffer offsets), fewer than 2^24 cells
bool sweep_mfma_supports(const CostVolumeArgs& a) {
  return a.C >= 1 && a.C <= kMfmaSweepChannels && (!a.image2_nhwc || a.C % 4 == 0) && static_cast<long long>(a.C) * a.H * a.W * 4 < (1LL << 31) &&
         static_cast<long long>(a.H) * a.W < (1LL << 24);
}

// the shipped configuration
using MfmaSweepDefault = MfmaSweepConfig<4, 4, 128, 1, 4, true, 1>;
int launch_sweep_mfma(const CostVolumeArgs& a, hipStream_t stream) {
  if (!sweep_mfma_supports(a)) return DVMVS_EUNSUPPORTED;
  return launch_sweep_mfma_cfg<MfmaSweepDefault>(a, stream);
}

#ifdef DVMVS_SWEEP_TUNING   // tools-only builds: configurations for tools/cv_microbench.py (variants 96 + k)
int launch_sweep_mfma_tuning(int which, const CostVolumeArgs& a, hipStream_t stream) {
  if (!sweep_mfma_supports(a)) return DVMVS_EUNSUPPORTED;
  switch (which) {   // <GW, GH, CAP, NW, WAVES, PREFETCH, TPI>
    case 0: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, true, 1>>(a, stream);
    case 1: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, false, 1>>(a, stream);
    case 2: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, false, 2>>(a, stream);
    case 3: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 3, true, 2>>(a, stream);
    case 4: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 2, 4, true, 1>>(a, stream);
    case 5: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 4, 4, true, 1>>(a, stream);
    case 6: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 256, 1, 3, true, 2>>(a, stream);
    case 7: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 256, 1, 4, true, 1>>(a, stream);
    case 8: return launch_sweep_mfma_cfg<MfmaSweepConfig<8, 2, 128, 1, 4, true, 1>>(a, stream);
    case 9: return launch_sweep_mfma_cfg<MfmaSweepConfig<2, 8, 128, 1, 4, true, 1>>(a, stream);
    case 10: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 5, false, 1>>(a, stream);
    case 11: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 4, 4, false, 1>>(a, stream);
    case 12: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 64, 1, 4, true, 1>>(a, stream);
    case 13: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 2, true, 2>>(a, stream);
    // ablations of configuration 0 (wrong results; where does the time go)
    case 16: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, true, 1, 1>>(a, stream);   // no operand loads
    case 17: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, true, 1, 2>>(a, stream);   // no MFMAs
    case 18: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, true, 1, 4>>(a, stream);   // no interpolation
    case 19: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, true, 1, 3>>(a, stream);   // neither loads nor MFMAs
    case 20: return launch_sweep_mfma_cfg<MfmaSweepConfig<4, 4, 128, 1, 4, true, 1, 7>>(a, stream);   // positions + boxes + table writes only
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
