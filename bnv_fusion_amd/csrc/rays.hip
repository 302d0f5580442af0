// Ray sampling and the SDF ray loss of the global optimiser, fused.
//
// Reference: src/utils/render_utils.py -- get_camera_params / lift (:411-458), stratified_sampling (:77-94),
// hierarchical_sampling (:191-233), render_with_rays (:461-505), compute_sdf_loss (:508-549).  The torch
// restatement of the same functions (bnv_fusion_amd/optimize.py, pinned to the reference's golden vectors)
// costs ~180 tiny launches per 1000-ray split and leaves the optimiser step bound by the HOST's launch rate
// (7.7 ms per step against 2 ms of kernels).  Here one split is
//   k_ray_samples   one thread per drawn sample: ray through the pixel, fine + coarse stratified samples, merged
//                   by distance (rank counting), world points; per sample the L1 target (signed distance to the nearest valid
//                   neighbouring surface point, truncated) and its weight (valid x ray mask);
//   k_count_optim_pts  weights[row] += 1 once per distinct corner row of the samples (count_optim, :602-622);
//   (decode_pts forward: csrc/decode.hip)
//   k_ray_loss      loss = sum w |pred - target| / n_valid and d loss / d pred;
//   (decode_pts backward).
// The uniforms of the stratified draws are an INPUT (the caller's generator), so the reference's random
// stream can be replayed bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bnv_fusion.h"
#include "bnv_common.hpp"

namespace bnv {

constexpr int kMaxSamples = 64;  // S = n_fine + n_coarse (interface bound)

struct RayArgs {
  const float* uv;        // [n, 2]
  const float* gt_pts;    // [n, 3]
  const float* ray_mask;  // [n]
  const float* nb_pts;    // [n, n_nb, 3]
  const float* nb_mask;   // [n, n_nb]
  const float* u_fine;    // [n, n_fine]   uniforms of the fine strata
  const float* u_coarse;  // [n, n_coarse]
  float T[16];            // T_wc row-major
  float K[9];             // intrinsics row-major
  int n, n_fine, n_coarse, n_nb;
  float truncated_dist;
  float* pts;             // [n, S, 3]
  float* target;          // [n, S]
  float* weight;          // [n, S]
};

// torch.linspace(0, 1, steps)[i] in float32 (symmetric evaluation about the midpoint)
__device__ __forceinline__ float linspace01(int i, int steps) {
  const float step = __fdiv_rn(1.f, (float)(steps - 1));
  return i < steps / 2 ? __fmul_rn(step, (float)i) : __fsub_rn(1.f, __fmul_rn(step, (float)(steps - i - 1)));
}

// stratified_sampling (:77-94): sample i of `steps` strata over [0, length]
__device__ __forceinline__ float stratum(int i, int steps, float length, float u) {
  const float e = __fmul_rn(linspace01(i, steps), length);
  const float lower = i == 0 ? e : __fmul_rn(0.5f, __fadd_rn(e, __fmul_rn(linspace01(i - 1, steps), length)));
  const float upper = i == steps - 1 ? e : __fmul_rn(0.5f, __fadd_rn(__fmul_rn(linspace01(i + 1, steps), length), e));
  return __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), u));
}

__device__ __forceinline__ float norm3(float x, float y, float z) {
  return sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z)));
}

// One thread per (ray, drawn sample).  The two stratified lists are ascending, so the position of a sample in
// the merged order is its index in its own list plus the number of samples of the other list in front of it
// (ties: fine first, like the sequential merge) -- no sort, no per-ray serial loop.
__global__ __launch_bounds__(256) void k_ray_samples(RayArgs A) {
  const int S = A.n_fine + A.n_coarse;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)A.n * S) return;
  const int r = (int)(t / S), e = (int)(t - (int64_t)r * S);
  const float* T = A.T;
  const float cam[3] = {T[3], T[7], T[11]};
  // lift (:411-428) with z = 1, then world = T [x y 1 1]^T, direction normalised (F.normalize: / max(|v|, 1e-12))
  const float u = A.uv[r * 2], v = A.uv[r * 2 + 1];
  const float fx = A.K[0], sk = A.K[1], cx = A.K[2], fy = A.K[4], cy = A.K[5];
  const float xl = __fdiv_rn(__fsub_rn(__fadd_rn(__fsub_rn(u, cx), __fdiv_rn(__fmul_rn(cy, sk), fy)),
                                       __fdiv_rn(__fmul_rn(sk, v), fy)), fx);
  const float yl = __fdiv_rn(__fsub_rn(v, cy), fy);
  float dir[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float w = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T[a * 4], xl), __fmul_rn(T[a * 4 + 1], yl)), T[a * 4 + 2]),
                              T[a * 4 + 3]);
    dir[a] = __fsub_rn(w, cam[a]);
  }
  const float dn = fmaxf(norm3(dir[0], dir[1], dir[2]), 1e-12f);
#pragma unroll
  for (int a = 0; a < 3; ++a) dir[a] = __fdiv_rn(dir[a], dn);
  const float gt[3] = {A.gt_pts[r * 3], A.gt_pts[r * 3 + 1], A.gt_pts[r * 3 + 2]};
  const float gt_depth = norm3(__fsub_rn(gt[0], cam[0]), __fsub_rn(gt[1], cam[1]), __fsub_rn(gt[2], cam[2]));
  // hierarchical_sampling (:191-233)
  const float off = A.truncated_dist;
  const float back = __fsub_rn(gt_depth, off) < 0.f ? gt_depth : off;
  float st[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) st[a] = __fsub_rn(__fsub_rn(gt[a], __fmul_rn(back, dir[a])), cam[a]);
  const float start_depth = norm3(st[0], st[1], st[2]);
  const float span = __fmul_rn(off, 2.f);
  const float* uf = A.u_fine + (size_t)r * A.n_fine;
  const float* uc = A.u_coarse + (size_t)r * A.n_coarse;
  float d;
  int rank;
  if (e < A.n_fine) {
    d = __fadd_rn(stratum(e, A.n_fine, span, uf[e]), start_depth);
    rank = e;
    for (int jq = 0; jq < A.n_coarse; ++jq) rank += stratum(jq, A.n_coarse, gt_depth, uc[jq]) < d;
  } else {
    const int jq = e - A.n_fine;
    d = stratum(jq, A.n_coarse, gt_depth, uc[jq]);
    rank = jq;
    for (int i = 0; i < A.n_fine; ++i) rank += __fadd_rn(stratum(i, A.n_fine, span, uf[i]), start_depth) <= d;
  }
  float p[3];
  const size_t o = (size_t)r * S + rank;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    p[a] = __fadd_rn(cam[a], __fmul_rn(d, dir[a]));
    A.pts[o * 3 + a] = p[a];
  }
  // compute_sdf_loss (:508-549)
  const float depth = norm3(__fsub_rn(p[0], cam[0]), __fsub_rn(p[1], cam[1]), __fsub_rn(p[2], cam[2]));
  const float gt_sdf = fminf(fmaxf(__fsub_rn(gt_depth, depth), -off), off);
  const bool valid = gt_sdf > fmaxf(-off * 0.5f, -0.05f);
  float nearest = 3.4e38f;
  for (int k = 0; k < A.n_nb; ++k) {
    const float* q = A.nb_pts + ((size_t)r * A.n_nb + k) * 3;
    const float dist = A.nb_mask[(size_t)r * A.n_nb + k] != 0.f
                           ? norm3(__fsub_rn(q[0], p[0]), __fsub_rn(q[1], p[1]), __fsub_rn(q[2], p[2]))
                           : 10000.f;
    nearest = fminf(nearest, dist);
  }
  const float sgn = gt_sdf > 0.f ? 1.f : -1.f;
  A.target[o] = fminf(fmaxf(__fmul_rn(nearest, sgn), -off), off);
  A.weight[o] = valid ? A.ray_mask[r] : 0.f;
}

// (split_samples > 0: sample i belongs to split i / split_samples and n_valid holds one value per split)
__global__ __launch_bounds__(256) void k_ray_loss(const float* __restrict__ pred, const float* __restrict__ target,
                                                  const float* __restrict__ weight, const float* __restrict__ n_valid,
                                                  int64_t m, int64_t split_samples, float* __restrict__ loss,
                                                  float* __restrict__ grad) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float l = 0.f;
  if (i < m) {
    const float inv = 1.f / n_valid[split_samples > 0 ? i / split_samples : 0];
    const float w = weight[i] * inv;
    const float d = pred[i] - target[i];
    l = fabsf(d) * w;
    grad[i] = d > 0.f ? w : (d < 0.f ? -w : 0.f);   // d|x|/dx with torch's sign(0) = 0
  }
  // block reduction, one atomic per block
  __shared__ float red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, red[0] + red[1] + red[2] + red[3]);
}

// count_optim on sample points (render_utils.py:491-493 + sparse_volume.py:602-622): one thread per (point, corner)
__global__ __launch_bounds__(256) void k_count_optim_pts(bnv_volume_t v, bnv_grid_t g, const float* __restrict__ pts,
                                                         int64_t m, int is_coords, float* __restrict__ weights,
                                                         int64_t row_limit, int32_t* __restrict__ stamp, int32_t epoch) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= m * 8) return;
  const int64_t q = t >> 3;
  const int cb = kCornerCeilBits[t & 7];
  int64_t c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float x = pts[q * 3 + a];
    if (!is_coords) x = __fdiv_rn(__fsub_rn(x, g.bound_min[a]), g.voxel_size);
    c[a] = (int64_t)(((cb >> a) & 1) ? ceilf(x) : floorf(x));
  }
  uint64_t key;
  if (!pack_key(c[0], c[1], c[2], &key)) return;
  const int row = volume_find(v.slot_keys, v.slot_rows, (uint32_t)(v.n_slots - 1), key);
  if (row < 0 || row >= row_limit) return;
  if (atomicExch(&stamp[row], epoch) != epoch) weights[row] = __fadd_rn(weights[row], 1.0f);
}

// The same for ALL ray splits of an optimiser step at once (sample q belongs to split q / split_samples): nothing is
// added yet -- bit s of split_mask[row] records that split s touches the row, the decode kernels rebuild from it the
// weight each split's mask decisions see (DecodeArgs::split_mask), and k_apply_split_counts adds the +1s afterwards.
__global__ __launch_bounds__(256) void k_count_optim_splits(bnv_volume_t v, bnv_grid_t g, const float* __restrict__ pts,
                                                            int64_t m, int is_coords, int64_t row_limit,
                                                            int64_t split_samples, uint32_t* __restrict__ split_mask) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= m * 8) return;
  const int64_t q = t >> 3;
  const int cb = kCornerCeilBits[t & 7];
  int64_t c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float x = pts[q * 3 + a];
    if (!is_coords) x = __fdiv_rn(__fsub_rn(x, g.bound_min[a]), g.voxel_size);
    c[a] = (int64_t)(((cb >> a) & 1) ? ceilf(x) : floorf(x));
  }
  uint64_t key;
  if (!pack_key(c[0], c[1], c[2], &key)) return;
  const int row = volume_find(v.slot_keys, v.slot_rows, (uint32_t)(v.n_slots - 1), key);
  if (row < 0 || row >= row_limit) return;
  const uint32_t bit = 1u << (uint32_t)(q / split_samples);
  if (!(split_mask[row] & bit)) atomicOr(&split_mask[row], bit);     // (most corners of a ray's samples repeat a row)
}

// weights[row] += 1 once per split that touched the row, as sequential exact fp32 additions (what the split-by-split
// calls of count_optim leave behind); the masks are cleared for the next step.
__global__ __launch_bounds__(256) void k_apply_split_counts(float* __restrict__ weights, uint32_t* __restrict__ split_mask,
                                                            int64_t n_rows) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  uint32_t m = split_mask[r];
  if (!m) return;
  float w = weights[r];
  while (m) {
    w = __fadd_rn(w, 1.0f);
    m &= m - 1u;
  }
  weights[r] = w;
  split_mask[r] = 0u;
}

}  // namespace bnv

using namespace bnv;

extern "C" {

int bnv_ray_samples(const float* uv, const float* gt_pts, const float* ray_mask, const float* nb_pts,
                    const float* nb_mask, int n_nb, const float T_wc[16], const float intr[9], const float* u_fine,
                    const float* u_coarse, int n, int n_fine, int n_coarse, float truncated_dist, float* pts,
                    float* target, float* weight, bnv_stream_t stream) {
  if (n < 0 || n_fine < 2 || n_coarse < 2 || n_fine + n_coarse > kMaxSamples || n_nb < 1) return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  if (!uv || !gt_pts || !ray_mask || !nb_pts || !nb_mask || !T_wc || !intr || !u_fine || !u_coarse || !pts || !target ||
      !weight)
    return BNV_ERR_INVALID_ARGUMENT;
  RayArgs a;
  a.uv = uv; a.gt_pts = gt_pts; a.ray_mask = ray_mask; a.nb_pts = nb_pts; a.nb_mask = nb_mask;
  a.u_fine = u_fine; a.u_coarse = u_coarse;
  for (int i = 0; i < 16; ++i) a.T[i] = T_wc[i];
  for (int i = 0; i < 9; ++i) a.K[i] = intr[i];
  a.n = n; a.n_fine = n_fine; a.n_coarse = n_coarse; a.n_nb = n_nb;
  a.truncated_dist = truncated_dist;
  a.pts = pts; a.target = target; a.weight = weight;
  const int64_t threads = (int64_t)n * (n_fine + n_coarse);
  hipLaunchKernelGGL(k_ray_samples, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_ray_loss_splits(const float* pred, const float* target, const float* weight, const float* n_valid, int64_t m,
                        int64_t split_samples, float* loss, float* grad, bnv_stream_t stream) {
  if (m < 0 || split_samples < 0 || (m > 0 && (!pred || !target || !weight || !n_valid || !loss || !grad)))
    return BNV_ERR_INVALID_ARGUMENT;
  if (m == 0) return BNV_OK;
  hipLaunchKernelGGL(k_ray_loss, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred, target,
                     weight, n_valid, m, split_samples, loss, grad);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_ray_loss(const float* pred, const float* target, const float* weight, const float* n_valid, int64_t m,
                 float* loss, float* grad, bnv_stream_t stream) {
  return bnv_ray_loss_splits(pred, target, weight, n_valid, m, 0, loss, grad, stream);
}

int bnv_volume_count_optim_splits(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* pts, int64_t m,
                                  int is_coords, int64_t row_limit, int64_t split_samples, uint32_t* split_mask,
                                  bnv_stream_t stream) {
  if (!vol || !grid || m < 0 || !split_mask || split_samples < 1 || (m + split_samples - 1) / split_samples > 31)
    return BNV_ERR_INVALID_ARGUMENT;
  if (m == 0) return BNV_OK;
  if (!pts) return BNV_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_count_optim_splits, dim3((unsigned)((m * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     *vol, *grid, pts, m, is_coords, row_limit, split_samples, split_mask);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_volume_apply_split_counts(float* weights, uint32_t* split_mask, int64_t n_rows, bnv_stream_t stream) {
  if (n_rows < 0 || (n_rows > 0 && (!weights || !split_mask))) return BNV_ERR_INVALID_ARGUMENT;
  if (n_rows == 0) return BNV_OK;
  hipLaunchKernelGGL(k_apply_split_counts, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     weights, split_mask, n_rows);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_volume_count_optim_pts(const bnv_volume_t* vol, const bnv_grid_t* grid, const float* pts, int64_t m,
                               int is_coords, float* weights, int64_t row_limit, int32_t* stamp, int32_t epoch,
                               bnv_stream_t stream) {
  if (!vol || !grid || m < 0 || !stamp || !weights) return BNV_ERR_INVALID_ARGUMENT;
  if (m == 0) return BNV_OK;
  if (!pts) return BNV_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_count_optim_pts, dim3((unsigned)((m * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *vol,
                     *grid, pts, m, is_coords, weights, row_limit, stamp, epoch);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

}  // extern "C"
