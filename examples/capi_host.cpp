// capi_host.cpp -- the hot path driven from a plain C++ host through include/bnv_fusion.h: no Python, no PyTorch.
// Memory comes from hipMalloc, the stream is a hipStream_t, the weights are the packed arrays the Python side also
// uploads (written to a directory by `python -m bnv_fusion_amd.export_packs`).  Per frame: uint16 depth image ->
// bnv_encode_begin_depth + bnv_encode_finish_image -> bnv_volume_integrate -> bnv_decode_lattice; one line per frame
// with the counters and order-sensitive checksums of the outputs (tests/test_gpu_capi_host.py compares them with the
// Python path on the same frames).
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/capi_host.cpp -o capi_host \
//         -Lbnv_fusion_amd -l:libbnv_fusion_hip.so -Wl,-rpath,$PWD/bnv_fusion_amd
//   ./capi_host <dir>        (<dir>: meta.bin, pointnet_pack.bin, sdfmlp_pack.bin, depth_<k>.u16)
//
// Reference counterpart: the loop of run_e2e.py:243-252 (NeuralMap.integrate, :78-109) + the per-frame decode of
// sparse_volume.py:697-738.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "bnv_fusion.h"

#define HIP_OK(x)                                                                          \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      std::exit(2);                                                                        \
    }                                                                                      \
  } while (0)
#define BNV_OK_OR_DIE(x)                                                                   \
  do {                                                                                     \
    int s_ = (x);                                                                          \
    if (s_ != BNV_OK) {                                                                    \
      std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, bnv_status_string(s_));      \
      std::exit(3);                                                                        \
    }                                                                                      \
  } while (0)

// meta.bin, written by bnv_fusion_amd/export_packs.py with struct.pack (little endian, no padding surprises: every
// member is 8-byte aligned by construction)
struct Meta {
  int32_t H, W, n_frames, mlp_mode;
  double max_depth;
  double K[9];
  bnv_grid_t grid;   // the very bytes of the ctypes struct the Python side passes
};

static std::vector<char> read_file(const std::string& path) {
  std::FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) {
    std::fprintf(stderr, "cannot open %s\n", path.c_str());
    std::exit(1);
  }
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<char> b((size_t)n);
  if (n && std::fread(b.data(), 1, (size_t)n, f) != (size_t)n) std::exit(1);
  std::fclose(f);
  return b;
}

template <typename T>
static T* dev_alloc(size_t n, bool zero = true) {
  void* p = nullptr;
  HIP_OK(hipMalloc(&p, n * sizeof(T) + 256));
  if (zero) HIP_OK(hipMemset(p, 0, n * sizeof(T) + 256));
  return (T*)p;
}

template <typename T>
static T* dev_upload(const std::vector<char>& bytes) {
  T* p = dev_alloc<T>(bytes.size() / sizeof(T), false);
  HIP_OK(hipMemcpy(p, bytes.data(), bytes.size(), hipMemcpyHostToDevice));
  return p;
}

// the checksum of bnv_fusion_amd/sequence.py: sum of value * (index % 1000003 + 1) over the bit patterns, mod 2^64
static uint64_t checksum_i64(const std::vector<int64_t>& v) {
  uint64_t s = 0;
  for (size_t i = 0; i < v.size(); ++i) s += (uint64_t)v[i] * (uint64_t)(i % 1000003 + 1);
  return s;
}
static uint64_t checksum_f32(const std::vector<float>& v) {
  uint64_t s = 0;
  for (size_t i = 0; i < v.size(); ++i) {
    int32_t bits;
    __builtin_memcpy(&bits, &v[i], 4);
    s += (uint64_t)(int64_t)bits * (uint64_t)(i % 1000003 + 1);
  }
  return s;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s <dir>\n", argv[0]);
    return 1;
  }
  const std::string dir = argv[1];
  const std::vector<char> mb = read_file(dir + "/meta.bin");
  if (mb.size() < sizeof(Meta) + 0) {
    std::fprintf(stderr, "meta.bin: %zu bytes, expected >= %zu\n", mb.size(), sizeof(Meta));
    return 1;
  }
  Meta m;
  __builtin_memcpy(&m, mb.data(), sizeof(Meta));
  const double* poses = (const double*)(mb.data() + sizeof(Meta));   // n_frames x 16
  if (mb.size() != sizeof(Meta) + (size_t)m.n_frames * 16 * sizeof(double)) {
    std::fprintf(stderr, "meta.bin: size does not match %d frames\n", m.n_frames);
    return 1;
  }
  HIP_OK(hipSetDevice(0));
  BNV_OK_OR_DIE(bnv_init(0));
  // the arithmetic mode travels with every call, in the grid (no process-global switch is touched)
  m.grid.mlp_mode = BNV_GRID_MLP_MODE(m.mlp_mode);
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));

  const float* pointnet_pack = dev_upload<float>(read_file(dir + "/pointnet_pack.bin"));
  const float* sdfmlp_pack = dev_upload<float>(read_file(dir + "/sdfmlp_pack.bin"));

  const int64_t n_pts = (int64_t)m.H * m.W;
  const int64_t n_vox = (int64_t)m.grid.n_xyz[0] * m.grid.n_xyz[1] * m.grid.n_xyz[2];
  const int64_t out_cap = 8 * n_pts < n_vox ? 8 * n_pts : n_vox;   // a frame cannot touch more voxels than either

  // encode: scratch (zero-filled once; every encode leaves it clean), outputs, counters
  const size_t enc_ws_bytes = bnv_encode_workspace_bytes(n_pts, m.grid.n_xyz);
  char* enc_ws = dev_alloc<char>(enc_ws_bytes);
  uint16_t* depth = dev_alloc<uint16_t>((size_t)n_pts);
  float* pts = dev_alloc<float>((size_t)n_pts * 6);
  float* feats = dev_alloc<float>((size_t)out_cap * 8);
  int64_t* pcounts = dev_alloc<int64_t>((size_t)out_cap);
  int64_t* flat_ids = dev_alloc<int64_t>((size_t)out_cap);
  int64_t* grid_ids = dev_alloc<int64_t>((size_t)out_cap * 3);
  bnv_encode_counters_t* counters = dev_alloc<bnv_encode_counters_t>(1);

  // the volume: open-addressing slot table + row arrays, all caller-owned (sparse_volume.py:587-600)
  bnv_volume_t vol{};
  vol.row_capacity = 1 << 20;
  vol.n_slots = 1 << 22;
  vol.slot_keys = dev_alloc<uint64_t>((size_t)vol.n_slots, false);
  vol.slot_rows = dev_alloc<int32_t>((size_t)vol.n_slots, false);
  vol.row_coords = dev_alloc<int64_t>((size_t)vol.row_capacity * 3);
  vol.features = dev_alloc<float>((size_t)vol.row_capacity * 8);
  vol.weights = dev_alloc<float>((size_t)vol.row_capacity);
  vol.num_hits = dev_alloc<float>((size_t)vol.row_capacity);
  vol.n_rows = dev_alloc<int32_t>(2);
  vol.n_feats = 8;
  vol.brick = nullptr;   // no dense row index: the hash alone
  BNV_OK_OR_DIE(bnv_volume_clear(&vol, stream));
  const size_t vol_ws_bytes = bnv_volume_workspace_bytes(out_cap);
  char* vol_ws = dev_alloc<char>(vol_ws_bytes);
  const size_t lat_ws_bytes = bnv_decode_lattice_workspace_bytes(out_cap, vol.row_capacity);
  char* lat_ws = dev_alloc<char>(lat_ws_bytes);       // zero-filled: per-row stamps start at 0, epochs from 1
  float* sdf = dev_alloc<float>((size_t)out_cap * 27);
  int32_t epoch = 0;

  std::vector<uint16_t> depth_h((size_t)n_pts);
  for (int k = 0; k < m.n_frames; ++k) {
    const std::vector<char> d = read_file(dir + "/depth_" + std::to_string(k) + ".u16");
    if (d.size() != (size_t)n_pts * 2) {
      std::fprintf(stderr, "depth_%d.u16: wrong size\n", k);
      return 1;
    }
    HIP_OK(hipMemcpyAsync(depth, d.data(), d.size(), hipMemcpyHostToDevice, stream));
    // front end + voxelisation + sorted-unique | point encoder + per-voxel mean + filter + repack
    BNV_OK_OR_DIE(bnv_encode_begin_depth(depth, /*uint16 millimetres*/ 0, m.H, m.W, m.K, poses + 16 * k, m.max_depth,
                                         &m.grid, enc_ws, enc_ws_bytes, n_pts, pts, stream));
    BNV_OK_OR_DIE(bnv_encode_finish_image(pts, n_pts, m.W, &m.grid, pointnet_pack, enc_ws, enc_ws_bytes, n_pts, feats,
                                          pcounts, flat_ids, grid_ids, out_cap, 0, counters, stream));
    // running-average upsert of the frame's voxels; the count stays on the device
    BNV_OK_OR_DIE(bnv_volume_integrate(&vol, grid_ids, feats, pcounts, out_cap, &counters->n_out, vol_ws, vol_ws_bytes,
                                       stream));
    bnv_encode_counters_t c;
    HIP_OK(hipMemcpyAsync(&c, counters, sizeof(c), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    if (c.error) {
      std::fprintf(stderr, "frame %d: encode error %d\n", k, c.error);
      return 4;
    }
    uint64_t cs_ids = 0, cs_sdf = 0;
    if (c.n_out > 0) {
      // the 3x3x3 SDF lattice of every voxel the frame touched
      BNV_OK_OR_DIE(bnv_decode_lattice(&vol, &m.grid, vol.features, vol.weights, vol.row_capacity, sdfmlp_pack, grid_ids,
                                       c.n_out, nullptr, nullptr, lat_ws, lat_ws_bytes, ++epoch, sdf, stream));
      std::vector<int64_t> ids_h((size_t)c.n_out * 3);
      std::vector<float> sdf_h((size_t)c.n_out * 27);
      HIP_OK(hipMemcpyAsync(ids_h.data(), grid_ids, ids_h.size() * 8, hipMemcpyDeviceToHost, stream));
      HIP_OK(hipMemcpyAsync(sdf_h.data(), sdf, sdf_h.size() * 4, hipMemcpyDeviceToHost, stream));
      HIP_OK(hipStreamSynchronize(stream));
      cs_ids = checksum_i64(ids_h);
      cs_sdf = checksum_f32(sdf_h);
    }
    int32_t rows[2];
    HIP_OK(hipMemcpy(rows, vol.n_rows, sizeof(rows), hipMemcpyDeviceToHost));
    if (rows[1]) {
      std::fprintf(stderr, "frame %d: upsert error %d\n", k, rows[1]);
      return 4;
    }
    std::printf("frame %d n_valid %d n_unique %d n_out %d n_avg_bits %u rows %d ids %llu sdf %llu\n", k,
                c.n_valid_points, c.n_unique, c.n_out, *(const uint32_t*)&c.n_avg_pts, rows[0],
                (unsigned long long)cs_ids, (unsigned long long)cs_sdf);
  }
  std::printf("done frames %d compute_units %d\n", m.n_frames, bnv_num_compute_units());
  return 0;
}
