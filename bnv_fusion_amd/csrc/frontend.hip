// frontend.hip -- depth image -> input_pts [N, 6] (world xyz, world normal) on the GPU
// (reference src/datasets/fusion_inference_dataset.py:40-90, src/utils/geometry.py:150-171,
// kornia 0.6.2 depth_to_normals restated as in geometry.py:515-527).
//
// The reference does this in float64 numpy on DataLoader workers and casts to float32 afterwards
// (run_e2e.py:249); bit-exact voxel ids downstream therefore need float64 here too.  One thread per
// pixel, float64 arithmetic in the reference's operation order (compiled with -ffp-contract=off), one
// rounding to float32 at the end, rows compacted by the validity mask in row-major pixel order with an
// ordered prefix sum.  HBM-bound and tiny: 0.6 MB of depth in, 7.4 MB of points out per 640x480 frame.
#include "frontend.hpp"

namespace bnv {

constexpr int kFrontThreads = 256;
constexpr int kFrontItems = 4;
constexpr int kFrontTile = kFrontThreads * kFrontItems;

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
    float p[6];
    front_point(a, (int)(i / a.W), (int)(i % a.W), p);
    float* o = out + (size_t)run * 6;
#pragma unroll
    for (int r = 0; r < 6; ++r) o[r] = p[r];
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
  front_args_fill(a, depth, depth_dtype, H, W, intr_host, T_wc_host, max_depth);
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
