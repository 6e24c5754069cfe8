// Sample geometry of the plane sweep shared by the gfx950 forward kernels (csrc/sweep_tiled.hip, csrc/sweep_mfma.hip): the reference's
// fp32 position arithmetic rounding for rounding, buffer-descriptor loads with hardware zero fill, the gather path of one sample.
// Semantics: /root/reference/dvmvs/utils.py:45-86.
#pragma once

#include "plane_sweep.h"

namespace dvmvs {

typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));

// Raw buffer descriptors for the feature maps: a load whose byte offset is >= num_records returns 0 without touching
// memory, which is exactly the zero apron of grid_sample(padding_mode='zeros') -- no exec-mask branches around the
// staging loads -- and the channel-plane offset rides in the scalar offset operand, so a staged element costs no VALU
// address arithmetic at all.
constexpr unsigned int kBufferOutOfRange = 0x80000000u;   // > any offset inside a map (maps are < 2 GiB, checked on the host)
__device__ inline __amdgpu_buffer_rsrc_t map_resource(gcfloat_p base, unsigned int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, static_cast<int>(bytes), 0x00020000);
}
__device__ inline float buffer_f32(__amdgpu_buffer_rsrc_t r, unsigned int voffset, unsigned int soffset) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, static_cast<int>(voffset), static_cast<int>(soffset), 0));
}
__device__ inline float4v buffer_f32x4(__amdgpu_buffer_rsrc_t r, unsigned int voffset, unsigned int soffset) {
  return __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(voffset), static_cast<int>(soffset), 0));
}
__device__ inline float2v fma2(float2v a, float2v b, float2v c) { return __builtin_elementwise_fma(a, b, c); }   // v_pk_fma_f32

// ---- geometry of one sample --------------------------------------------------------------------------------------------
// [X0, Y0, Z0] = Hm * [x, y, 1] in the reference's op order (utils.py:68); per plane [X, Y, Z] = [X0, Y0, Z0] + K t / depth.
struct SweepRay {
  float X0, Y0, Z0;
};

__host__ __device__ inline SweepRay sweep_ray(const float* Hm, float xf, float yf) {
  SweepRay r;
  r.X0 = fmaf(Hm[2], 1.0f, fmaf(Hm[1], yf, Hm[0] * xf));
  r.Y0 = fmaf(Hm[5], 1.0f, fmaf(Hm[4], yf, Hm[3] * xf));
  r.Z0 = fmaf(Hm[8], 1.0f, fmaf(Hm[7], yf, Hm[6] * xf));
  return r;
}

// Sample positions are computed with the REFERENCE's fp32 arithmetic, rounding for rounding: u = X / (Z + 1e-8),
// g = (u - W/2) / (W/2) (utils.py:70-73), pixel = ((g + 1) / 2) * (W - 1) (ATen's align_corners un-normalisation).  The
// position is where fp32 round-off enters the volume (1e-5 px times the feature gradient), and the network downstream
// amplifies it: with "mathematically equal, differently rounded" positions the hybrid pipeline of
// tests/test_hybrid_parity.py sits 9e-5 (depth rel-L1) from the oracle, with identical positions ~1e-6.  Divisions are
// IEEE-exact for finite normal operands without v_div_scale / v_div_fixup: a refined reciprocal, then two residual
// corrections (the sequence the compiler emits for '/', minus its range scaling -- sample coordinates never need it).
struct SweepScale {
  float wn, hn;       // W / 2, H / 2
  float r_wn, r_hn;   // their refined reciprocals
  float Wm1, Hm1;     // W - 1, H - 1
  float Wf, Hf;
};

#pragma clang fp contract(off)
__device__ inline float refined_rcp(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}

__device__ inline float exact_div(float n, float d, float rcp_d) {   // n / d, correctly rounded (finite, normal range)
  float q = n * rcp_d;
  q = fmaf(fmaf(-d, q, n), rcp_d, q);
  return fmaf(fmaf(-d, q, n), rcp_d, q);
}

__device__ inline SweepScale sweep_scale(int W, int H) {
  SweepScale s;
  s.Wf = static_cast<float>(W);
  s.Hf = static_cast<float>(H);
  s.wn = s.Wf * 0.5f;
  s.hn = s.Hf * 0.5f;
  s.r_wn = refined_rcp(s.wn);
  s.r_hn = refined_rcp(s.hn);
  s.Wm1 = static_cast<float>(W - 1);
  s.Hm1 = static_cast<float>(H - 1);
  return s;
}

// un-clamped sample position (NaN / Inf when Z + 1e-8 == 0, as in the reference).  x and y go through the same operations, so
// they ride in the two halves of packed instructions (v_pk_add / v_pk_mul / v_pk_fma_f32: one issue slot for both): 20 VALU
// instructions per sample instead of 35, with the scalar version's rounding.
__device__ inline void sweep_position_exact(const SweepRay& r, float kx, float ky, float kz, const SweepScale& s, float* ix, float* iy,
                                            float* denom_out = nullptr) {
  const float Z = r.Z0 + kz;
  const float denom = Z + 1e-8f;
  const float rcp = refined_rcp(denom);
  if (denom_out) *denom_out = denom;
  const float2v n = float2v{r.X0, r.Y0} + float2v{kx, ky};          // X, Y
  const float2v d = {denom, denom}, rd = {rcp, rcp};
  float2v q = n * rd;                                                // exact_div(X, denom), exact_div(Y, denom)
  q = fma2(fma2(-d, q, n), rd, q);
  q = fma2(fma2(-d, q, n), rd, q);
  const float2v half = {s.wn, s.hn}, rhalf = {s.r_wn, s.r_hn};
  const float2v t = q - half;
  float2v g = t * rhalf;                                             // exact_div(u - W/2, W/2), exact_div(v - H/2, H/2)
  g = fma2(fma2(-half, g, t), rhalf, g);
  g = fma2(fma2(-half, g, t), rhalf, g);
  const float2v pos = ((g + 1.0f) * 0.5f) * float2v{s.Wm1, s.Hm1};
  *ix = pos.x;
  *iy = pos.y;
}
#pragma clang fp contract(fast)

// Sample position in measurement-image pixels, clamped to [-1, W] x [-1, H]: everything at or beyond those bounds has
// all four taps outside the image (zeros padding), and v_max/v_min return the non-NaN operand, so NaN (Z + 1e-8 == 0)
// lands on -1 as well, where both taps are zero -- ATen's "non-finite coordinates fail the bounds test".
__device__ inline void sweep_sample(const SweepRay& r, float kx, float ky, float kz, const SweepScale& s, float* ix, float* iy) {
  float px, py;
  sweep_position_exact(r, kx, ky, kz, s, &px, &py);
  *ix = fminf(fmaxf(px, -1.0f), s.Wf);
  *iy = fminf(fmaxf(py, -1.0f), s.Hf);
}

// N samples of one ray at once, stage by stage: per sample exactly the operations of sweep_sample (same rounding), issued so that N
// independent dependency chains are in flight -- one sample's packed-arithmetic chain is ~20 dependent instructions, each of which the
// compiler otherwise pads with a wait state (csrc/sweep_mfma.hip: a lane's four planes, two at a time: four at once cost 20 registers
// more than the kernel has).  k[j] = K t / depth of sample j's plane.
#pragma clang fp contract(off)
template <int N>
__device__ inline void sweep_samples(const SweepRay& r, const float4v* k, const SweepScale& s, float* ix, float* iy) {
  float denom[N], rcp[N];
  float2v n[N], q[N], t[N], g[N];
  const float2v half = {s.wn, s.hn}, rhalf = {s.r_wn, s.r_hn};
#pragma unroll
  for (int j = 0; j < N; ++j) denom[j] = (r.Z0 + k[j].z) + 1e-8f;
#pragma unroll
  for (int j = 0; j < N; ++j) rcp[j] = __builtin_amdgcn_rcpf(denom[j]);
#pragma unroll
  for (int j = 0; j < N; ++j) rcp[j] = fmaf(fmaf(-denom[j], rcp[j], 1.0f), rcp[j], rcp[j]);      // refined_rcp
#pragma unroll
  for (int j = 0; j < N; ++j) n[j] = float2v{r.X0, r.Y0} + float2v{k[j].x, k[j].y};
#pragma unroll
  for (int j = 0; j < N; ++j) q[j] = n[j] * float2v{rcp[j], rcp[j]};
#pragma unroll
  for (int rep = 0; rep < 2; ++rep)
#pragma unroll
    for (int j = 0; j < N; ++j) q[j] = fma2(fma2(-float2v{denom[j], denom[j]}, q[j], n[j]), float2v{rcp[j], rcp[j]}, q[j]);
#pragma unroll
  for (int j = 0; j < N; ++j) t[j] = q[j] - half;
#pragma unroll
  for (int j = 0; j < N; ++j) g[j] = t[j] * rhalf;
#pragma unroll
  for (int rep = 0; rep < 2; ++rep)
#pragma unroll
    for (int j = 0; j < N; ++j) g[j] = fma2(fma2(-half, g[j], t[j]), rhalf, g[j]);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const float2v pos = ((g[j] + 1.0f) * 0.5f) * float2v{s.Wm1, s.Hm1};
    ix[j] = fminf(fmaxf(pos.x, -1.0f), s.Wf);
    iy[j] = fminf(fmaxf(pos.y, -1.0f), s.Hf);
  }
}
#pragma clang fp contract(fast)

// ---- gather path (no staging): taps straight from global memory ----------------------------------------------------------
// One plane of one measurement frame for this thread's pixel: sum_c ref[c] * warped[c].  Used for runs of planes whose
// footprint cannot be staged: by sweep_spill_kernel (second pass) and, when the caller gave no spill workspace, inline.
// KCH channels x four taps independent loads are in flight before the first use (KCH = 32, all 128 loads of a plane, was
// measured in the second pass: no faster than 8).
template <bool NHWC, int KCH = 8>
__device__ inline float gather_plane(const CostVolumeArgs& a, gcfloat_p meas, gcfloat_p ref, int HW, const SweepRay& ray,
                                     float kx, float ky, float kz, const SweepScale& sc) {
  float ix, iy;
  sweep_sample(ray, kx, ky, kz, sc, &ix, &iy);
  const BilinearTaps t = make_taps(ix, iy, a.W, a.H);
  const int xa = t.in_x0 ? t.x0 : 0, xb = t.in_x1 ? t.x0 + 1 : 0;
  const int ya = t.in_y0 ? t.y0 : 0, yb = t.in_y1 ? t.y0 + 1 : 0;
  const int es = NHWC ? a.C : 1;
  const int off[4] = {(ya * a.W + xa) * es, (ya * a.W + xb) * es, (yb * a.W + xa) * es, (yb * a.W + xb) * es};
  const float wgt[4] = {(t.in_x0 && t.in_y0) ? t.w_nw : 0.0f, (t.in_x1 && t.in_y0) ? t.w_ne : 0.0f,
                        (t.in_x0 && t.in_y1) ? t.w_sw : 0.0f, (t.in_x1 && t.in_y1) ? t.w_se : 0.0f};
  float sum = 0.0f;
  // under strong magnification most pixels sample outside the image: a wave whose 64 pixels are all dead for this plane
  // skips its channel loop (wave-uniform branch)
  if (!__any((wgt[0] + wgt[1] + wgt[2] + wgt[3]) != 0.0f)) return 0.0f;
  constexpr int kChan = KCH;
  for (int c0 = 0; c0 < a.C; c0 += kChan) {
    float r[kChan], v[kChan][4];
    if (NHWC) {
      float4v q[kChan / 4][4];
#pragma unroll
      for (int h = 0; h < kChan / 4; ++h)
#pragma unroll
        for (int tp = 0; tp < 4; ++tp)
          q[h][tp] = (c0 + 4 * h < a.C) ? *(const float4v DVMVS_GLOBAL*)(meas + off[tp] + c0 + 4 * h) : float4v{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int cc = 0; cc < kChan; ++cc)
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) v[cc][tp] = q[cc / 4][tp][cc % 4];
    } else {
#pragma unroll
      for (int cc = 0; cc < kChan; ++cc) {
        gcfloat_p plane = meas + static_cast<size_t>(min(c0 + cc, a.C - 1)) * HW;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) v[cc][tp] = plane[off[tp]];
      }
    }
#pragma unroll
    for (int cc = 0; cc < kChan; ++cc) r[cc] = (c0 + cc < a.C) ? ref[static_cast<size_t>(min(c0 + cc, a.C - 1)) * HW] : 0.0f;
#pragma unroll
    for (int cc = 0; cc < kChan; ++cc) {
      float w = v[cc][0] * wgt[0];
      w += v[cc][1] * wgt[1];
      w += v[cc][2] * wgt[2];
      w += v[cc][3] * wgt[3];
      sum += r[cc] * w;
    }
  }
  return sum;
}

// ATen's bilinear weights (ix_se - ix)(iy_se - iy), ... from the fractional position: (fx + 1) - ix == 1 - (ix - fx) bit for bit
// wherever the tap it multiplies is inside the image (ix - fx is exact for ix >= 0, and then so is its complement).
__device__ inline void tap_weights(float frac_x, float frac_y, float2v* w_n, float2v* w_s) {
  const float2v xw = {1.0f - frac_x, frac_x};   // {west, east}
  *w_n = xw * (1.0f - frac_y);                  // {nw, ne}
  *w_s = xw * frac_y;                           // {sw, se}
}

}  // namespace dvmvs
