// frontend.hip -- depth image -> input_pts [N, 6] (world xyz, world normal) on the GPU
// (reference src/datasets/fusion_inference_dataset.py:40-90, src/utils/geometry.py:150-171,
// kornia 0.6.2 depth_to_normals restated as in geometry.py:515-527).
//
// The reference does this in float64 numpy on DataLoader workers and casts to float32 afterwards
// (run_e2e.py:249); bit-exact voxel ids downstream therefore need float64 here too.  One thread per
// pixel, float64 arithmetic in the reference's operation order (compiled with -ffp-contract=off), one
// rounding to float32 at the end, rows compacted by the validity mask in row-major pixel order with an
// ordered prefix sum.  HBM-bound and tiny: 0.6 MB of depth in, 7.4 MB of points out per 640x480 frame.
#include "bnv_common.hpp"

namespace bnv {

constexpr int kFrontThreads = 256;
constexpr int kFrontItems = 4;
constexpr int kFrontTile = kFrontThreads * kFrontItems;

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

__global__ __launch_bounds__(kFrontThreads) void k_front_count(FrontArgs a, uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t wave_tot[kFrontThreads / 64];
  const int64_t n = (int64_t)a.H * a.W;
  const int64_t base = (int64_t)blockIdx.x * kFrontTile + (int64_t)threadIdx.x * kFrontItems;
  uint32_t s = 0;
#pragma unroll
  for (int e = 0; e < kFrontItems; ++e) {
    const int64_t i = base + e;
    if (i < n) s += depth_at(a, (int)(i / a.W), (int)(i % a.W)) > 0.0 ? 1u : 0u;
  }
  uint32_t total;
  block_exclusive_scan<kFrontThreads>(s, wave_tot, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// (256 threads: one wave per SIMD and 24 VGPRs fit beside the persistent MLP kernels of another stream; a 1024-thread
// workgroup does not, and stalled its stream until the MLP kernel had finished -- rocprofv3 kernel trace)
__global__ __launch_bounds__(256) void k_front_top(uint32_t* __restrict__ block_sums, int n_blocks,
                                                    int32_t* __restrict__ total_out) {
  __shared__ uint32_t wave_tot[16];
  uint32_t carry = 0;
  for (int base = 0; base < n_blocks; base += 256) {
    const int i = base + threadIdx.x;
    const uint32_t v = (i < n_blocks) ? block_sums[i] : 0;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<256>(v, wave_tot, &total);
    if (i < n_blocks) block_sums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) *total_out = (int32_t)carry;
}

// pad_total: when not null (= the valid-pixel count written by k_front_top), the rows behind the valid ones are
// filled with NaN here -- invalid pixel i goes to row n_valid + (i - valid pixels before i) -- so that the caller
// needs no separate fill of the H*W-row buffer.
__global__ __launch_bounds__(kFrontThreads) void k_front_points(FrontArgs a, const uint32_t* __restrict__ block_sums,
                                                                float* __restrict__ out,
                                                                const int32_t* __restrict__ pad_total) {
  __shared__ uint32_t wave_tot[kFrontThreads / 64];
  const int64_t n = (int64_t)a.H * a.W;
  const int64_t base = (int64_t)blockIdx.x * kFrontTile + (int64_t)threadIdx.x * kFrontItems;
  double dd[kFrontItems];
  uint32_t s = 0;
#pragma unroll
  for (int e = 0; e < kFrontItems; ++e) {
    const int64_t i = base + e;
    dd[e] = (i < n) ? depth_at(a, (int)(i / a.W), (int)(i % a.W)) : 0.0;
    s += dd[e] > 0.0 ? 1u : 0u;
  }
  uint32_t total;
  uint32_t run = block_exclusive_scan<kFrontThreads>(s, wave_tot, &total) + block_sums[blockIdx.x];
#pragma unroll
  for (int e = 0; e < kFrontItems; ++e) {
    const int64_t i = base + e;
    if (!(dd[e] > 0.0)) {
      if (pad_total && i < n) {
        float* o = out + ((size_t)*pad_total + (size_t)(i - run)) * 6;
#pragma unroll
        for (int r = 0; r < 6; ++r) o[r] = __builtin_nanf("");
      }
      continue;
    }
    const int y = (int)(i / a.W), x = (int)(i % a.W);
    const double d = dd[e];
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
    float* o = out + (size_t)run * 6;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      o[r] = (float)(((a.T[r * 4 + 0] * pc[0] + a.T[r * 4 + 1] * pc[1]) + a.T[r * 4 + 2] * pc[2]) + a.T[r * 4 + 3]);
      o[3 + r] = (float)((a.T[r * 4 + 0] * nrm[0] + a.T[r * 4 + 1] * nrm[1]) + a.T[r * 4 + 2] * nrm[2]);
    }
    ++run;
  }
}

}  // namespace bnv

using namespace bnv;

extern "C" {

size_t bnv_depth_workspace_bytes(int H, int W) {
  const int64_t n = (int64_t)H * W;
  return (size_t)(((n + kFrontTile - 1) / kFrontTile + 1) * 4 + 256);
}

static int depth_to_points_impl(const void* depth, int depth_dtype, int H, int W, const double* intr_host,
                                const double* T_wc_host, double max_depth, void* ws, size_t ws_bytes, float* out_pts,
                                int32_t* n_out, bool pad, bnv_stream_t stream_) {
  if (!depth || !intr_host || !T_wc_host || !ws || !out_pts || !n_out || H <= 0 || W <= 0 || depth_dtype < 0 ||
      depth_dtype > 2 || (int64_t)H * W >= (1LL << 31))
    return BNV_ERR_INVALID_ARGUMENT;
  if (ws_bytes < bnv_depth_workspace_bytes(H, W)) return BNV_ERR_WORKSPACE_TOO_SMALL;
  FrontArgs a;
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
  hipStream_t stream = (hipStream_t)stream_;
  const int nb = (int)(((int64_t)H * W + kFrontTile - 1) / kFrontTile);
  uint32_t* sums = (uint32_t*)ws;
  hipLaunchKernelGGL(k_front_count, dim3(nb), dim3(kFrontThreads), 0, stream, a, sums);
  BNV_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_front_top, dim3(1), dim3(256), 0, stream, sums, nb, n_out);
  BNV_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_front_points, dim3(nb), dim3(kFrontThreads), 0, stream, a, sums, out_pts,
                     pad ? (const int32_t*)n_out : (const int32_t*)nullptr);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_depth_to_points(const void* depth, int depth_dtype, int H, int W, const double* intr_host,
                        const double* T_wc_host, double max_depth, void* ws, size_t ws_bytes, float* out_pts,
                        int32_t* n_out, bnv_stream_t stream) {
  return depth_to_points_impl(depth, depth_dtype, H, W, intr_host, T_wc_host, max_depth, ws, ws_bytes, out_pts, n_out,
                              false, stream);
}

int bnv_depth_to_points_padded(const void* depth, int depth_dtype, int H, int W, const double* intr_host,
                               const double* T_wc_host, double max_depth, void* ws, size_t ws_bytes, float* out_pts,
                               int32_t* n_out, bnv_stream_t stream) {
  return depth_to_points_impl(depth, depth_dtype, H, W, intr_host, T_wc_host, max_depth, ws, ws_bytes, out_pts, n_out,
                              true, stream);
}

}  // extern "C"
