// frontend.hpp -- per-pixel arithmetic of the depth front end (depth image -> world point + normal), shared by
// frontend.hip (compacting kernels behind bnv_depth_to_points) and encode.hip (k_front_mark: front end fused with the
// voxel marking of the encode).  Reference: src/datasets/fusion_inference_dataset.py:40-90, src/utils/geometry.py:150-171,
// kornia 0.6.2 depth_to_normals as restated at geometry.py:515-527.  float64 in the reference's operation order
// (compiled with -ffp-contract=off), one rounding to float32 at the end.
#pragma once
#include "bnv_common.hpp"

namespace bnv {

struct FrontArgs {
  const void* depth;
  int dtype;  // 0: uint16 millimetres (cv2.imread(...)/1000., common.py:93), 1: float32 metres, 2: float64 metres
  int H, W;
  double fx, fy, cx, cy;
  double T[12];  // rows 0..2 of T_wc
  double max_depth;
  float fxf, fyf, cxf, cyf;  // depth2xyz builds its pixel rays in float32 (geometry.py:163-168)
};

__device__ __forceinline__ double depth_at(const FrontArgs& a, int y, int x) {
  y = y < 0 ? 0 : (y >= a.H ? a.H - 1 : y);  // replicate padding of the Sobel filter
  x = x < 0 ? 0 : (x >= a.W ? a.W - 1 : x);
  const size_t i = (size_t)y * a.W + x;
  double d;
  if (a.dtype == 0) d = (double)((const uint16_t*)a.depth)[i] / 1000.0;
  else if (a.dtype == 1) d = (double)((const float*)a.depth)[i];
  else d = ((const double*)a.depth)[i];
  // mask = depth > 0 (& depth < max_depth); depth = depth * mask   (common.py:107-110)
  return (d > 0.0 && d < a.max_depth) ? d : 0.0;
}

__device__ __forceinline__ void xyz_at(const FrontArgs& a, int y, int x, double (&p)[3]) {
  const int yc = y < 0 ? 0 : (y >= a.H ? a.H - 1 : y);
  const int xc = x < 0 ? 0 : (x >= a.W ? a.W - 1 : x);
  const double d = depth_at(a, yc, xc);
  p[0] = ((double)xc - a.cx) / a.fx * d;
  p[1] = ((double)yc - a.cy) / a.fy * d;
  p[2] = d;
}

// World point (out[0..2]) and world normal (out[3..5]) of pixel (y, x), rounded to float32; false if the pixel is
// invalid (depth 0 or >= max_depth).
__device__ __forceinline__ bool front_point(const FrontArgs& a, int y, int x, float (&out)[6]) {
  const double d = depth_at(a, y, x);
  if (!(d > 0.0)) return false;
  // ---- normal: Sobel/8 of the xyz map, cross product, L2 normalise (kornia depth_to_normals) ----
  double A[3], B[3], C[3], D[3], E[3], F[3], gx[3], gy[3];
  xyz_at(a, y - 1, x + 1, A); xyz_at(a, y, x + 1, B); xyz_at(a, y + 1, x + 1, C);
  xyz_at(a, y - 1, x - 1, D); xyz_at(a, y, x - 1, E); xyz_at(a, y + 1, x - 1, F);
#pragma unroll
  for (int c = 0; c < 3; ++c) gx[c] = (((((A[c] + 2.0 * B[c]) + C[c]) - D[c]) - 2.0 * E[c]) - F[c]) / 8.0;
  xyz_at(a, y + 1, x - 1, A); xyz_at(a, y + 1, x, B); xyz_at(a, y + 1, x + 1, C);
  xyz_at(a, y - 1, x - 1, D); xyz_at(a, y - 1, x, E); xyz_at(a, y - 1, x + 1, F);
#pragma unroll
  for (int c = 0; c < 3; ++c) gy[c] = (((((A[c] + 2.0 * B[c]) + C[c]) - D[c]) - 2.0 * E[c]) - F[c]) / 8.0;
  double nrm[3] = {gx[1] * gy[2] - gx[2] * gy[1], gx[2] * gy[0] - gx[0] * gy[2], gx[0] * gy[1] - gx[1] * gy[0]};
  const double len = sqrt((nrm[0] * nrm[0] + nrm[1] * nrm[1]) + nrm[2] * nrm[2]);
  const double den = len > 1e-12 ? len : 1e-12;
#pragma unroll
  for (int c = 0; c < 3; ++c) nrm[c] = nrm[c] / den;
  // ---- point: depth2xyz with float32 pixel rays, then T_wc ----
  const double ur = (double)__fdiv_rn(__fsub_rn((float)x, a.cxf), a.fxf);
  const double vr = (double)__fdiv_rn(__fsub_rn((float)y, a.cyf), a.fyf);
  const double pc[3] = {ur * d, vr * d, d};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    out[r] = (float)(((a.T[r * 4 + 0] * pc[0] + a.T[r * 4 + 1] * pc[1]) + a.T[r * 4 + 2] * pc[2]) + a.T[r * 4 + 3]);
    out[3 + r] = (float)((a.T[r * 4 + 0] * nrm[0] + a.T[r * 4 + 1] * nrm[1]) + a.T[r * 4 + 2] * nrm[2]);
  }
  return true;
}

// host side: fills FrontArgs from the C-ABI arguments
static inline void front_args_fill(FrontArgs& a, const void* depth, int depth_dtype, int H, int W,
                                   const double* intr_host, const double* T_wc_host, double max_depth) {
  a.depth = depth;
  a.dtype = depth_dtype;
  a.H = H;
  a.W = W;
  a.fx = intr_host[0];
  a.fy = intr_host[4];
  a.cx = intr_host[2];
  a.cy = intr_host[5];
  for (int i = 0; i < 12; ++i) a.T[i] = T_wc_host[i];
  a.max_depth = max_depth;
  a.fxf = (float)a.fx;
  a.fyf = (float)a.fy;
  a.cxf = (float)a.cx;
  a.cyf = (float)a.cy;
}

}  // namespace bnv
