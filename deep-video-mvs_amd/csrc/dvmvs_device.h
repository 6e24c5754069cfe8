// Shared device helpers for the gfx950 plane-sweep kernels: small fp64 matrix algebra for the pose set-up,
// the bilinear tap decomposition used by every warp kernel, and launch plumbing.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dvmvs_hip.h"

namespace dvmvs {

constexpr int kWave = 64;  // gfx950 wavefront

// Pointers that reach a kernel inside a by-value argument struct are generic ("flat") to the compiler: every access
// becomes a flat_load / flat_store, which is counted on BOTH vmcnt and lgkmcnt and therefore serialises against the
// ds_read waits of the LDS-bound sweep kernels.  Hot paths re-type them as global (address space 1) once, at the top.
#define DVMVS_GLOBAL __attribute__((address_space(1)))
typedef const float DVMVS_GLOBAL* gcfloat_p;
typedef float DVMVS_GLOBAL* gfloat_p;
typedef unsigned int DVMVS_GLOBAL* guint_p;
__device__ inline gcfloat_p as_global(const float* p) { return (gcfloat_p)p; }
__device__ inline gfloat_p as_global(float* p) { return (gfloat_p)p; }
__device__ inline guint_p as_global(unsigned int* p) { return (guint_p)p; }

#define DVMVS_RETURN_IF_HIP(expr)                      \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return static_cast<int>(_e); \
  } while (0)

inline int launch_status() { return static_cast<int>(hipGetLastError()); }

// ---- fp64 small-matrix algebra (row-major) -------------------------------------------------------------------
// The reference evaluates inverse(pose)/inverse(K) with an fp32 LU (torch.inverse); evaluating the same algebra
// in fp64 and rounding once lands within the reference's own fp32 round-off of it, needs no library call and
// no host round trip.

__device__ inline double det3(double a, double b, double c, double d, double e, double f, double g, double h, double i) {
  return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
}

__device__ inline void inverse3(const double* m, double* o) {
  const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const double inv_det = 1.0 / (m[0] * c00 + m[1] * c01 + m[2] * c02);
  o[0] = c00 * inv_det;
  o[1] = (m[2] * m[7] - m[1] * m[8]) * inv_det;
  o[2] = (m[1] * m[5] - m[2] * m[4]) * inv_det;
  o[3] = c01 * inv_det;
  o[4] = (m[0] * m[8] - m[2] * m[6]) * inv_det;
  o[5] = (m[2] * m[3] - m[0] * m[5]) * inv_det;
  o[6] = c02 * inv_det;
  o[7] = (m[1] * m[6] - m[0] * m[7]) * inv_det;
  o[8] = (m[0] * m[4] - m[1] * m[3]) * inv_det;
}

// General 4x4 inverse by cofactors (poses are rigid, but the reference does not assume it).
__device__ inline void inverse4(const double* m, double* o) {
  double cof[16];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r0 = (r + 1) & 3, r1 = (r + 2) & 3, r2 = (r + 3) & 3;
      const int c0 = (c + 1) & 3, c1 = (c + 2) & 3, c2 = (c + 3) & 3;
      // remaining rows/cols taken in cyclic order: a rotation of three indices is an even permutation, so the
      // 3x3 determinant equals the standard minor and the cofactor sign is the usual (-1)^(r+c)
      const double minor = det3(m[r0 * 4 + c0], m[r0 * 4 + c1], m[r0 * 4 + c2],
                                m[r1 * 4 + c0], m[r1 * 4 + c1], m[r1 * 4 + c2],
                                m[r2 * 4 + c0], m[r2 * 4 + c1], m[r2 * 4 + c2]);
      cof[r * 4 + c] = ((r + c) & 1) ? -minor : minor;
    }
  }
  const double det = m[0] * cof[0] + m[1] * cof[1] + m[2] * cof[2] + m[3] * cof[3];
  const double inv_det = 1.0 / det;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) o[r * 4 + c] = cof[c * 4 + r] * inv_det;
}

__device__ inline void matmul4(const double* a, const double* b, double* o) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += a[r * 4 + k] * b[k * 4 + c];
      o[r * 4 + c] = s;
    }
}

__device__ inline void matmul3(const double* a, const double* b, double* o) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) o[r * 3 + c] = a[r * 3 + 0] * b[0 * 3 + c] + a[r * 3 + 1] * b[1 * 3 + c] + a[r * 3 + 2] * b[2 * 3 + c];
}

// out = inverse(a) * c for fp32 4x4 inputs, result in fp64.
__device__ inline void relative_pose_f64(const float* a, const float* c, double* out) {
  double A[16], C[16], Ai[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    A[i] = static_cast<double>(a[i]);
    C[i] = static_cast<double>(c[i]);
  }
  inverse4(A, Ai);
  matmul4(Ai, C, out);
}

// ---- bilinear taps ---------------------------------------------------------------------------------------------
// grid_sample(bilinear, zeros padding, align_corners=True) at pixel position (ix, iy): four taps around
// floor(ix), floor(iy); a tap contributes only if it lies inside the image; non-finite positions contribute 0.
struct BilinearTaps {
  int x0, y0;             // north-west tap (may be outside the image)
  float w_nw, w_ne, w_sw, w_se;
  bool in_x0, in_x1, in_y0, in_y1;
};

__device__ inline BilinearTaps make_taps(float ix, float iy, int W, int H) {
  BilinearTaps t;
  // Anything this far out (or NaN/Inf) has no tap inside; clamp before the int conversion so it is well defined.
  const bool sane = (ix > -2.0f) && (ix < static_cast<float>(W) + 1.0f) && (iy > -2.0f) && (iy < static_cast<float>(H) + 1.0f);
  const float fx = sane ? floorf(ix) : -2.0f;
  const float fy = sane ? floorf(iy) : -2.0f;
  const float ax = sane ? ix : -2.0f;
  const float ay = sane ? iy : -2.0f;
  t.x0 = static_cast<int>(fx);
  t.y0 = static_cast<int>(fy);
  const float ex = (fx + 1.0f) - ax;  // ix_se - ix
  const float wx = ax - fx;           // ix - ix_nw
  const float ey = (fy + 1.0f) - ay;
  const float wy = ay - fy;
  t.w_nw = ex * ey;
  t.w_ne = wx * ey;
  t.w_sw = ex * wy;
  t.w_se = wx * wy;
  t.in_x0 = (t.x0 >= 0) && (t.x0 < W);
  t.in_x1 = (t.x0 + 1 >= 0) && (t.x0 + 1 < W);
  t.in_y0 = (t.y0 >= 0) && (t.y0 < H);
  t.in_y1 = (t.y0 + 1 >= 0) && (t.y0 + 1 < H);
  return t;
}

// grid_sample's align_corners=True un-normalisation, in the op order ATen uses: ((g + 1) / 2) * (size - 1)
__device__ inline float unnormalize_ac(float g, int size) { return ((g + 1.0f) * 0.5f) * static_cast<float>(size - 1); }

}  // namespace dvmvs
