// tsdf.hip -- TSDFVolume.integrate (reference third_parties/fusion.py:68-141, the inline
// PyCUDA kernel that run_e2e.py:99-109 calls once per frame inside NeuralMap.integrate).
//
// One thread per voxel, voxel index fastest along z, so tsdf / weight / colour accesses are coalesced
// and each voxel is read-modified-written once: HBM-bound, 24 B per touched voxel (13.5 MB at the
// reference's fixed 0.025 m grid over a 2.54 m volume -> microseconds).  fp32 arithmetic in the
// reference kernel's order (compiled with -ffp-contract=off; nvcc would fuse some mul+add pairs).
#include "bnv_common.hpp"

namespace bnv {

struct TsdfArgs {
  float* tsdf;
  float* weight;
  float* color;        // may be null together with color_im
  int dim[3];
  float origin[3];
  float intr[9];
  float pose[16];
  float voxel_size, trunc_margin, obs_weight;
  int im_h, im_w;
  const float* color_im;  // folded b*65536 + g*256 + r, or null
  const float* depth_im;  // metres, 0 = invalid
  const uint16_t* depth_mm;  // alternative input: the dataset's uint16 millimetres (common.py:93: / 1000.)
  float max_depth;           // > 0: samples with depth >= max_depth are invalid (common.py:110-113: depth * mask)
  const int32_t* gate;       // device int32 or null: *gate == 0 -> the launch does nothing (the reference returns from
                             // NeuralMap.integrate before the TSDF fusion when the encode found no point, run_e2e.py:91-92)
};

__global__ __launch_bounds__(256) void k_tsdf_integrate(TsdfArgs a) {
  const int64_t n = (int64_t)a.dim[0] * a.dim[1] * a.dim[2];
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  if (a.gate && *a.gate == 0) return;
  const int yz = a.dim[1] * a.dim[2];
  const int vx = (int)(idx / yz), vy = (int)((idx - (int64_t)vx * yz) / a.dim[2]);
  const int vz = (int)(idx - (int64_t)vx * yz - (int64_t)vy * a.dim[2]);
  // voxel grid -> world -> camera (fusion.py:90-101)
  const float pt_x = a.origin[0] + (float)vx * a.voxel_size;
  const float pt_y = a.origin[1] + (float)vy * a.voxel_size;
  const float pt_z = a.origin[2] + (float)vz * a.voxel_size;
  const float tx = pt_x - a.pose[3], ty = pt_y - a.pose[7], tz = pt_z - a.pose[11];
  const float cx = a.pose[0] * tx + a.pose[4] * ty + a.pose[8] * tz;
  const float cy = a.pose[1] * tx + a.pose[5] * ty + a.pose[9] * tz;
  const float cz = a.pose[2] * tx + a.pose[6] * ty + a.pose[10] * tz;
  // camera -> pixel (fusion.py:103-104)
  const int px = (int)roundf(a.intr[0] * (cx / cz) + a.intr[2]);
  const int py = (int)roundf(a.intr[4] * (cy / cz) + a.intr[5]);
  if (px < 0 || px >= a.im_w || py < 0 || py >= a.im_h || cz < 0.f) return;
  const float depth = a.depth_mm ? __fdiv_rn((float)a.depth_mm[(size_t)py * a.im_w + px], 1000.f)
                                 : a.depth_im[(size_t)py * a.im_w + px];
  if (depth == 0.f || (a.max_depth > 0.f && !(depth < a.max_depth))) return;
  const float diff = depth - cz;
  if (diff < -a.trunc_margin) return;
  const float dist = fminf(1.0f, diff / a.trunc_margin);
  const float w_old = a.weight[idx];
  const float w_new = w_old + a.obs_weight;
  a.weight[idx] = w_new;
  a.tsdf[idx] = (a.tsdf[idx] * w_old + a.obs_weight * dist) / w_new;
  if (a.color && a.color_im) {  // fusion.py:127-139
    const float oc = a.color[idx];
    const float ob = floorf(oc / 65536.f), og = floorf((oc - ob * 65536.f) / 256.f);
    const float orr = oc - ob * 65536.f - og * 256.f;
    const float nc = a.color_im[(size_t)py * a.im_w + px];
    float nb = floorf(nc / 65536.f), ng = floorf((nc - nb * 65536.f) / 256.f);
    float nr = nc - nb * 65536.f - ng * 256.f;
    nb = fminf(roundf((ob * w_old + a.obs_weight * nb) / w_new), 255.0f);
    ng = fminf(roundf((og * w_old + a.obs_weight * ng) / w_new), 255.0f);
    nr = fminf(roundf((orr * w_old + a.obs_weight * nr) / w_new), 255.0f);
    a.color[idx] = nb * 65536.f + ng * 256.f + nr;
  }
}

// Several consecutive frames in one launch (the replay loop of the frame-parallel multi-GPU mode): the voxel's
// tsdf / weight stay in registers between frames, every frame applies exactly the single-frame arithmetic above,
// in frame order -> identical results to one launch per frame.  Colour (optional, per frame) follows the same rule.
constexpr int kTsdfBatchMax = BNV_TSDF_BATCH_MAX;
struct TsdfBatchArgs {
  float* tsdf;
  float* weight;
  float* color;                            // may be null
  const float* color_im[kTsdfBatchMax];    // folded b*65536 + g*256 + r per frame, or null
  int dim[3];
  float origin[3];
  float voxel_size, trunc_margin, obs_weight, max_depth;
  int im_h, im_w, n_frames;
  const uint16_t* depth_mm[kTsdfBatchMax];
  float intr[kTsdfBatchMax][4];    // fx, cx, fy, cy
  float pose[kTsdfBatchMax][12];   // rows 0..2 of the camera-to-world matrix
};

__global__ __launch_bounds__(256) void k_tsdf_integrate_batch(TsdfBatchArgs a) {
  const int64_t n = (int64_t)a.dim[0] * a.dim[1] * a.dim[2];
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int yz = a.dim[1] * a.dim[2];
  const int vx = (int)(idx / yz), vy = (int)((idx - (int64_t)vx * yz) / a.dim[2]);
  const int vz = (int)(idx - (int64_t)vx * yz - (int64_t)vy * a.dim[2]);
  const float pt_x = a.origin[0] + (float)vx * a.voxel_size;
  const float pt_y = a.origin[1] + (float)vy * a.voxel_size;
  const float pt_z = a.origin[2] + (float)vz * a.voxel_size;
  bool loaded = false, c_dirty = false;
  float w_cur = 0.f, t_cur = 0.f, c_cur = 0.f;
  for (int f = 0; f < a.n_frames; ++f) {
    const float* P = a.pose[f];
    const float tx = pt_x - P[3], ty = pt_y - P[7], tz = pt_z - P[11];
    const float cx = P[0] * tx + P[4] * ty + P[8] * tz;
    const float cy = P[1] * tx + P[5] * ty + P[9] * tz;
    const float cz = P[2] * tx + P[6] * ty + P[10] * tz;
    const int px = (int)roundf(a.intr[f][0] * (cx / cz) + a.intr[f][1]);
    const int py = (int)roundf(a.intr[f][2] * (cy / cz) + a.intr[f][3]);
    if (px < 0 || px >= a.im_w || py < 0 || py >= a.im_h || cz < 0.f) continue;
    const float depth = __fdiv_rn((float)a.depth_mm[f][(size_t)py * a.im_w + px], 1000.f);
    if (depth == 0.f || (a.max_depth > 0.f && !(depth < a.max_depth))) continue;
    const float diff = depth - cz;
    if (diff < -a.trunc_margin) continue;
    const float dist = fminf(1.0f, diff / a.trunc_margin);
    if (!loaded) {
      w_cur = a.weight[idx];
      t_cur = a.tsdf[idx];
      if (a.color) c_cur = a.color[idx];
      loaded = true;
    }
    const float w_new = w_cur + a.obs_weight;
    t_cur = (t_cur * w_cur + a.obs_weight * dist) / w_new;
    if (a.color && a.color_im[f]) {  // fusion.py:127-139, as k_tsdf_integrate
      const float ob = floorf(c_cur / 65536.f), og = floorf((c_cur - ob * 65536.f) / 256.f);
      const float orr = c_cur - ob * 65536.f - og * 256.f;
      const float nc = a.color_im[f][(size_t)py * a.im_w + px];
      float nb = floorf(nc / 65536.f), ng = floorf((nc - nb * 65536.f) / 256.f);
      float nr = nc - nb * 65536.f - ng * 256.f;
      nb = fminf(roundf((ob * w_cur + a.obs_weight * nb) / w_new), 255.0f);
      ng = fminf(roundf((og * w_cur + a.obs_weight * ng) / w_new), 255.0f);
      nr = fminf(roundf((orr * w_cur + a.obs_weight * nr) / w_new), 255.0f);
      c_cur = nb * 65536.f + ng * 256.f + nr;
      c_dirty = true;
    }
    w_cur = w_new;
  }
  if (loaded) {
    a.weight[idx] = w_cur;
    a.tsdf[idx] = t_cur;
    if (c_dirty) a.color[idx] = c_cur;
  }
}

}  // namespace bnv

using namespace bnv;

static int tsdf_integrate_impl(float* tsdf, float* weight, float* color, const int32_t dim_host[3],
                                  const float origin_host[3], float voxel_size, float trunc_margin,
                                  const float* depth_im, const uint16_t* depth_mm, const float* color_im, int im_h, int im_w,
                                  const float intr_host[9], const float pose_host[16], float obs_weight,
                                  float max_depth, const int32_t* gate, bnv_stream_t stream) {
  if (!tsdf || !weight || !dim_host || !origin_host || (!depth_im && !depth_mm) || !intr_host || !pose_host || im_h <= 0 ||
      im_w <= 0)
    return BNV_ERR_INVALID_ARGUMENT;
  TsdfArgs a;
  a.tsdf = tsdf;
  a.weight = weight;
  a.color = color;
  for (int i = 0; i < 3; ++i) {
    a.dim[i] = dim_host[i];
    a.origin[i] = origin_host[i];
  }
  for (int i = 0; i < 9; ++i) a.intr[i] = intr_host[i];
  for (int i = 0; i < 16; ++i) a.pose[i] = pose_host[i];
  a.voxel_size = voxel_size;
  a.trunc_margin = trunc_margin;
  a.obs_weight = obs_weight;
  a.im_h = im_h;
  a.im_w = im_w;
  a.color_im = color_im;
  a.depth_im = depth_im;
  a.depth_mm = depth_mm;
  a.max_depth = max_depth;
  a.gate = gate;
  const int64_t n = (int64_t)a.dim[0] * a.dim[1] * a.dim[2];
  if (n <= 0) return BNV_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_tsdf_integrate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

extern "C" int bnv_tsdf_integrate(float* tsdf, float* weight, float* color, const int32_t dim_host[3],
                                  const float origin_host[3], float voxel_size, float trunc_margin,
                                  const float* depth_im, const float* color_im, int im_h, int im_w,
                                  const float intr_host[9], const float pose_host[16], float obs_weight,
                                  float max_depth, const int32_t* gate, bnv_stream_t stream) {
  return tsdf_integrate_impl(tsdf, weight, color, dim_host, origin_host, voxel_size, trunc_margin, depth_im, nullptr,
                             color_im, im_h, im_w, intr_host, pose_host, obs_weight, max_depth, gate, stream);
}

extern "C" int bnv_tsdf_integrate_u16(float* tsdf, float* weight, float* color, const int32_t dim_host[3],
                                      const float origin_host[3], float voxel_size, float trunc_margin,
                                      const uint16_t* depth_mm, const float* color_im, int im_h, int im_w,
                                      const float intr_host[9], const float pose_host[16], float obs_weight,
                                      float max_depth, const int32_t* gate, bnv_stream_t stream) {
  return tsdf_integrate_impl(tsdf, weight, color, dim_host, origin_host, voxel_size, trunc_margin, nullptr, depth_mm,
                             color_im, im_h, im_w, intr_host, pose_host, obs_weight, max_depth, gate, stream);
}

extern "C" int bnv_tsdf_integrate_batch_u16(float* tsdf, float* weight, float* color, const int32_t dim_host[3],
                                            const float origin_host[3], float voxel_size, float trunc_margin,
                                            int n_frames, const uint16_t* const* depth_mm,
                                            const float* const* color_im, int im_h, int im_w,
                                            const float* intr_host, const float* pose_host, float obs_weight,
                                            float max_depth, bnv_stream_t stream) {
  if (!tsdf || !weight || !dim_host || !origin_host || !depth_mm || !intr_host || !pose_host || im_h <= 0 || im_w <= 0 ||
      n_frames < 0 || n_frames > kTsdfBatchMax)
    return BNV_ERR_INVALID_ARGUMENT;
  if (n_frames == 0) return BNV_OK;
  TsdfBatchArgs a = {};
  a.tsdf = tsdf;
  a.weight = weight;
  a.color = (color && color_im) ? color : nullptr;
  for (int i = 0; i < 3; ++i) {
    a.dim[i] = dim_host[i];
    a.origin[i] = origin_host[i];
  }
  a.voxel_size = voxel_size;
  a.trunc_margin = trunc_margin;
  a.obs_weight = obs_weight;
  a.max_depth = max_depth;
  a.im_h = im_h;
  a.im_w = im_w;
  a.n_frames = n_frames;
  for (int f = 0; f < n_frames; ++f) {
    if (!depth_mm[f]) return BNV_ERR_INVALID_ARGUMENT;
    a.depth_mm[f] = depth_mm[f];
    a.color_im[f] = a.color ? color_im[f] : nullptr;
    const float* K = intr_host + 9 * f;
    a.intr[f][0] = K[0];
    a.intr[f][1] = K[2];
    a.intr[f][2] = K[4];
    a.intr[f][3] = K[5];
    for (int i = 0; i < 12; ++i) a.pose[f][i] = pose_host[16 * f + i];
  }
  const int64_t n = (int64_t)a.dim[0] * a.dim[1] * a.dim[2];
  if (n <= 0) return BNV_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_tsdf_integrate_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}
