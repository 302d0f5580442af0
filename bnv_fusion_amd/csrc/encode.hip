// encode.hip -- LitFusionPointNet.encode_pointcloud (reference local_point_fusion.py:81-165)
// as four gfx950 kernels (bnv_encode_begin = the first two, bnv_encode_finish = the last two):
//
//   k_mark / k_front_mark  points (or depth pixels: front end fused in, frontend.hpp) -> 8 corner voxels ->
//                     flag bytes in a grid byte map, plain stores, no atomics                 (HBM)
//   k_rank            one pass over the flagged chunks (decoupled look-back): bitmap words + popcount prefix --
//                     the rank of a voxel's bit IS its position in torch.unique's ascending output
//                     (replaces sort+unique)                                                   (HBM/L2)
//   k_pointnet_scatter[_x|_t]  per (point, corner) pair: 6->128->128->128->8 MLP on MFMA
//                     (transposed chaining: layer L's D registers are layer L+1's B operands, no
//                     cross-lane traffic), then order-independent 64-bit fixed-point atomics into
//                     per-voxel accumulators                                                   (MFMA)
//   k_finalize        mean, min-points filter, ordered compaction in one pass (look-back), unflatten,
//                     scratch cleanup, the frame's counters                                    (HBM)
//
// Layout of one MFMA tile: 32 pairs = 32 consecutive points x one corner.  Exact fp32 (k_pointnet_scatter,
// 32x32x2 MFMA): lane l = (j = l & 31: pair, h = l >> 5); D register r of a 32-feature block holds feature
// (r&3) + 8*(r>>2) + 4*h of pair j, so the K-step that consumes D[r] as its B operand contracts features
// {f0(r), f0(r)+4}.  Split modes (k_pointnet_scatter_x, 16x16x32 MFMA): lane l = (n = l & 15, g = l >> 4), two
// column blocks of 16 pairs, eight row blocks of 16 features (layout at the kernel).  The packed A operands
// (weights) are pre-permuted on the host to match (bnv_fusion_amd/weights.py: pack_pointnet).
#include <stddef.h>

#include <atomic>
#include <utility>
#include <vector>

#include "frontend.hpp"

namespace bnv {

int g_num_cus = 0;
int g_last_hip_error = 0;
// Process-wide words, all relaxed atomics read once per launch.  The A/B switches choose between implementations with
// identical results; the MLP mode here is only the DEFAULT of calls whose grid does not name one (bnv_grid_t.mlp_mode).
std::atomic<int> g_reserve_cus{0};  // bnv_set_option("reserve_cus"): CUs the persistent MLP kernels leave to other streams
std::atomic<int> g_finalize_blocks{0};      // bnv_set_option("finalize_blocks"): workgroups of k_finalize (0: 2 per CU, which is also the most it may use); tests force the striding with a small value
std::atomic<int> g_tcnn_shared_table{1};    // bnv_set_option("tcnn_shared_table"): 1 = one LDS table per workgroup and 16 x 16 patch, 0 = per wave and block
std::atomic<int> g_tcnn_block_encoder{1};  // bnv_set_option("tcnn_block_encoder"): 1 = k_pointnet_scatter_tb for whole frames
std::atomic<int> g_mlp_mode{1};  // default arithmetic: 0 exact fp32 MFMA; 1 fp32 operands split into f16 hi+lo; 2 tcnn fp16 networks; 3 f16 operands

// ---- HIP-event timing of the dominant kernels, recorded on the stream they are launched on ----
bool g_prof_on = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events[PROF_KINDS];
static size_t g_prof_used[PROF_KINDS] = {0, 0, 0, 0};

void prof_mark(int kind, bool begin, hipStream_t stream) {
  auto& ring = g_prof_events[kind];
  if (begin) {
    if (g_prof_used[kind] == ring.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      ring.emplace_back(a, b);
    }
    (void)hipEventRecord(ring[g_prof_used[kind]].first, stream);
  } else if (g_prof_used[kind] < ring.size()) {
    (void)hipEventRecord(ring[g_prof_used[kind]].second, stream);
    ++g_prof_used[kind];
  }
}

// ------------------------------------------------------------------------------------------
// packed point-encoder weights (floats)
// ------------------------------------------------------------------------------------------
constexpr int PN_W1 = 0;                        // [3 kstep][4 mb][64 lane]
constexpr int PN_W2 = PN_W1 + 3 * 4 * 64;       // [4 mb][4 nb][4 rq][64 lane][4]
constexpr int PN_W3 = PN_W2 + 128 * 128;        // same
constexpr int PN_W4 = PN_W3 + 128 * 128;        // [4 nb][4 rq][2 h][8 n][4]
constexpr int PN_B1 = PN_W4 + 4 * 4 * 2 * 8 * 4;  // [128]
constexpr int PN_B2 = PN_B1 + 128;
constexpr int PN_B3 = PN_B2 + 128;
constexpr int PN_B4 = PN_B3 + 128;              // [8]
constexpr int PN_TOTAL = PN_B4 + 8;             // 34,952 floats = 139,808 B of LDS

// split-operand pack of the f16 modes (appended to the same pack, units: 16-bit halves from float offset PN_TOTAL),
// operand order of v_mfma_f32_16x16x32_f16 (k_pointnet_scatter_x)
constexpr int PX_W1 = 0;                          // [8 rb][hi/lo][64 lane][8]
constexpr int PX_W2 = PX_W1 + 8 * 2 * 64 * 8;     // [4 s][8 rb][hi/lo][64 lane][8]
constexpr int PX_W3 = PX_W2 + 4 * 8 * 2 * 64 * 8;
constexpr int PX_W4 = PX_W3 + 4 * 8 * 2 * 64 * 8; // [4 s][hi/lo][64 lane][8], rows >= 8 zero
constexpr int PX_TOTAL = PX_W4 + 4 * 2 * 64 * 8;  // 77,824 halves = 155,648 B
constexpr int PX_OFF = PN_TOTAL;                  // float offset of the PX pack in the packed weights
constexpr int PN_CERT = PN_TOTAL + PX_TOTAL / 2;  // [4]: certified bound on |normal component| of the split modes
constexpr int PN_PACK_FLOATS = PN_CERT + 4;
constexpr int PX_LDS_BYTES = PX_TOTAL * 2 + (128 * 3 + 8) * 4 + 32;  // halves + fp32 biases + tile counter (+ pad: lanes g = 3 read 16 B past b4) = 157,248 B

constexpr float kFixedScale = 4294967296.0f;    // 2^32: per-voxel sums are exact integers

// ------------------------------------------------------------------------------------------
// workspace layout
// ------------------------------------------------------------------------------------------
// Control block: the first 512 bytes of the workspace.  All-zero between frames (k_finalize's closing workgroup
// leaves it so), so no kernel of a frame needs a memset in front of it.
struct EncCtl {
  int32_t n_pairs;             // sharded encode: (point, corner) pairs whose voxel this rank owns (mark kernel)
  int32_t n_unique;            // U: touched voxels (k_rank)
  int32_t error;               // != 0: a capacity was exceeded
  int32_t n_orphans;           // first-touch ownership: points with a corner voxel in a block that has no owner yet
  int32_t n_deferred;          // first-touch ownership: touched voxels whose boundary test waits for k_shard_assign
  int32_t pad[11];
  int32_t shard_boundary[64];  // sharded encode: touched BOUNDARY voxels owned by each rank (k_rank) -- an upper
                               // bound of the boundary records that rank will exchange for this frame, known on
                               // every rank (the voxelisation is replicated) before the encoder MLP starts
};
static_assert(sizeof(EncCtl) <= 512, "control block");

struct EncodeWs {
  EncCtl* ctl;
  uint64_t* tile_state;   // [n_tiles] look-back state of k_rank / k_finalize (epoch-tagged, never cleared)
  int32_t* valid_blocks;  // [ceil(max_points / 256)] points that passed the bounds mask, per workgroup of the mark kernel
  int32_t* pair_list;     // [8 * max_points] sharded encode: (point << 3 | corner) of the pairs this rank owns
  int32_t* orphan_list;   // [max_points] first-touch ownership: points the mark kernel could not decide (k_shard_own)
  int32_t* defer_list;    // [max_unique] first-touch ownership: slots whose boundary test k_rank could not decide
  uint8_t* bytemap;       // [n_words * 32] one byte per voxel: set by the mark kernel (plain stores), consumed and
                          // cleared by k_rank
  uint8_t* chunk_flag;    // [n_chunks] one byte per 64 voxels (2 bitmap words): any byte of the chunk set
  uint32_t* bitmap;       // [n_words] one bit per touched voxel: written by k_rank, read by the encoder, cleared by
                          // k_finalize
  uint32_t* word_prefix;  // [n_words]
  int32_t* ids;           // [max_unique] flat voxel id of slot s (ascending)
  int32_t* counts;        // [max_unique]
  long long* acc;         // [max_unique][8] fixed-point feature sums
  int64_t n_words;
  int64_t n_chunks;       // n_words / 2
  int64_t max_unique;
  int64_t n_tiles;
};

constexpr int kScanThreads = 256;
#ifndef BNV_FIN_THREADS
#define BNV_FIN_THREADS 1024
#endif
constexpr int kFinTile = BNV_FIN_THREADS;              // slots per workgroup (k_finalize: one per thread)
constexpr int kRankItems = 4;
constexpr int kRankTile = kScanThreads * kRankItems;  // 1024 chunks (of 64 voxels = 2 bitmap words) per workgroup (k_rank)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static size_t encode_ws_layout(int64_t max_points, const int32_t n_xyz[3], char* base, EncodeWs* ws) {
  const int64_t nvox = (int64_t)n_xyz[0] * n_xyz[1] * n_xyz[2];
  const int64_t n_words = align_up((size_t)((nvox + 31) / 32), 8);   // whole chunks, whole u32x4 of chunk flags
  const int64_t n_chunks = n_words / 2;
  int64_t max_unique = 8 * max_points;
  if (max_unique > nvox) max_unique = nvox;
  if (max_unique < 1) max_unique = 1;
  const int64_t nb_words = (n_chunks + kRankTile - 1) / kRankTile;
  const int64_t nb_unique = (max_unique + kFinTile - 1) / kFinTile;
  const int64_t n_tiles = nb_words > nb_unique ? nb_words : nb_unique;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off = align_up(off + bytes, 256);
    return p;
  };
  char* p_ctl = take(512);   // control block first: its offset does not depend on the sizes
  char* p_state = take(n_tiles * 8);
  char* p_valid = take(((max_points + 255) / 256 + 1) * 4);
  char* p_pairs = take((size_t)(max_points > 0 ? max_points : 1) * 8 * 4);
  char* p_orph = take((size_t)(max_points > 0 ? max_points : 1) * 4);
  char* p_bytes = take(n_words * 32);
  char* p_chunks = take(n_chunks);
  char* p_bitmap = take(n_words * 4);
  char* p_prefix = take(n_words * 4);
  char* p_ids = take(max_unique * 4);
  char* p_counts = take(max_unique * 4);
  char* p_acc = take(max_unique * 8 * 8);
  char* p_defer = take(max_unique * 4);
  if (ws) {
    ws->ctl = (EncCtl*)p_ctl;
    ws->tile_state = (uint64_t*)p_state;
    ws->valid_blocks = (int32_t*)p_valid;
    ws->pair_list = (int32_t*)p_pairs;
    ws->orphan_list = (int32_t*)p_orph;
    ws->bytemap = (uint8_t*)p_bytes;
    ws->chunk_flag = (uint8_t*)p_chunks;
    ws->n_chunks = n_chunks;
    ws->bitmap = (uint32_t*)p_bitmap;
    ws->word_prefix = (uint32_t*)p_prefix;
    ws->ids = (int32_t*)p_ids;
    ws->counts = (int32_t*)p_counts;
    ws->acc = (long long*)p_acc;
    ws->defer_list = (int32_t*)p_defer;
    ws->n_words = n_words;
    ws->max_unique = max_unique;
    ws->n_tiles = n_tiles;
  }
  return off;
}

// every launch of a look-back kernel takes a fresh epoch (bnv_common.hpp: lookback_exclusive)
// (atomic: host threads driving different streams / volumes each get their own; the 30-bit tag never takes the value
// 0, which is what a zero-initialised workspace word carries)
static std::atomic<uint32_t> g_epoch{0};
uint32_t next_epoch() {
  uint32_t e;
  do e = g_epoch.fetch_add(1, std::memory_order_relaxed) + 1;
  while ((e & 0x3fffffffu) == 0);
  return e;
}

// ------------------------------------------------------------------------------------------
// mark: one thread per point; flags its 8 corner voxels in the grid byte map
// ------------------------------------------------------------------------------------------
// One BYTE per voxel, written with plain stores: setting a flag is idempotent, so no atomic is needed, nothing is
// read back and nothing waits -- where the former bit map cost one visibility load + one device-scope atomicOr per
// (point, column): 32 us of a 42 us kernel (tools/probe_mark.hip: 4 us with byte stores).  A second byte per 64-voxel
// chunk lets k_rank find the touched chunks without reading the whole map (134 MB at 512^3).  ~20 pairs fall into each
// voxel and neighbouring pixels (= neighbouring lanes) mostly share it: a lane skips a column its predecessor writes.
// The number of valid points goes to valid_blocks[blockIdx.x] as a plain store: one atomicAdd per wave on a single
// counter serialises in the memory-side atomic unit at ~11 ns each -- 4,800 of them were 52 us per frame.
// Every thread of the (256-thread) workgroup must call this.
//
// Spatial sharding (g.shard_world > 1, pair_list set): the (point, corner) pairs whose voxel THIS rank owns are also
// listed -- (point << 3 | corner), corners of a workgroup's points in (corner, point) order so that neighbouring
// pixels stay neighbours and the encoder's wave-level run reduction keeps working -- and the encoder then forms
// its tiles from the list: 1 / world of the pairs instead of every tile that holds at least one owned pair
// (with 8^3-voxel blocks that was ~60 % of the tiles at world 8).  One atomicAdd per WORKGROUP on the list counter.
// The (point, corner) pairs of a 256-thread workgroup's points whose voxel THIS rank owns, appended to pair_list as
// (point << 3 | corner) in (corner, point) order.  Every thread of the workgroup must call this (two barriers).
// First-touch ownership, mark kernel (orphan_list set): a point with a corner voxel in a block that has no owner YET
// (the frame's k_shard_assign has not run) lists nothing here and goes to orphan_list; k_shard_own lists its pairs
// once the owners are known.  A frame that touches no new block has no orphan.
__device__ __forceinline__ void list_owned_pairs(bool valid, int fx, int cx, int fy, int cy, int fz, int cz,
                                                 const bnv_grid_t& g, int point_index,
                                                 int32_t* __restrict__ pair_list, int32_t* __restrict__ n_pairs,
                                                 int32_t* __restrict__ orphan_list = nullptr,
                                                 int32_t* __restrict__ n_orphans = nullptr) {
  const int lane = threadIdx.x & 63;
  __shared__ int s_cnt[32];   // [corner][wave] owned pairs
  unsigned long long own[8];
  int owner8[8];
  bool orphan = false;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int gx = (k & 1) ? cx : fx, gy = (k & 2) ? cy : fy, gz = (k & 4) ? cz : fz;
    owner8[k] = valid ? voxel_owner(gx, gy, gz, g) : -2;
    orphan |= owner8[k] == -1;
  }
  if (orphan_list) {   // (workgroup-uniform)
    const unsigned long long ob = __ballot(orphan);
    if (ob) {
      int base = 0;
      if (lane == 0) base = atomicAdd(n_orphans, (int)__popcll(ob));
      base = __shfl(base, 0, 64);
      if (orphan) orphan_list[base + (int)__popcll(ob & ((1ull << lane) - 1ull))] = point_index;
    }
    if (orphan) valid = false;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    own[k] = __ballot(valid && owner8[k] == g.shard_rank);
    if (lane == 0) s_cnt[k * 4 + (threadIdx.x >> 6)] = (int)__popcll(own[k]);
  }
  __syncthreads();
  if (threadIdx.x < 64) {   // exclusive prefix of the 32 counts (first wave), then the workgroup's place in the list
    const int c = threadIdx.x < 32 ? s_cnt[threadIdx.x] : 0;
    int incl = c;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if ((int)threadIdx.x >= d) incl += o;
    }
    const int total = __shfl(incl, 31, 64);
    int base = 0;
    if (threadIdx.x == 0 && total) base = atomicAdd(n_pairs, total);
    base = __shfl(base, 0, 64);
    if (threadIdx.x < 32) s_cnt[threadIdx.x] = base + incl - c;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if ((own[k] >> lane) & 1ull)
      pair_list[s_cnt[k * 4 + (threadIdx.x >> 6)] + (int)__popcll(own[k] & ((1ull << lane) - 1ull))] =
          (point_index << 3) | k;
}

__device__ __forceinline__ void mark_point(bool valid, float x, float y, float z, const bnv_grid_t& g,
                                           uint8_t* __restrict__ bytemap, uint8_t* __restrict__ chunk_flag,
                                           int32_t* __restrict__ valid_blocks, int point_index = 0,
                                           int32_t* __restrict__ pair_list = nullptr,
                                           int32_t* __restrict__ n_pairs = nullptr,
                                           int32_t* __restrict__ orphan_list = nullptr,
                                           int32_t* __restrict__ n_orphans = nullptr) {
  int fx = 0, cx = 0, fy = 0, cy = 0, fz = 0, cz = 0;
  if (valid) {
    const float xn = voxel_coord(x, g.bound_min[0], g.voxel_size);
    const float yn = voxel_coord(y, g.bound_min[1], g.voxel_size);
    const float zn = voxel_coord(z, g.bound_min[2], g.voxel_size);
    fx = (int)floorf(xn), cx = (int)ceilf(xn);
    fy = (int)floorf(yn), cy = (int)ceilf(yn);
    fz = (int)floorf(zn), cz = (int)ceilf(zn);
  }
  const int nyz = g.n_xyz[1] * g.n_xyz[2];
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int gx = (k & 1) ? cx : fx, gy = (k & 2) ? cy : fy;
    int a = valid ? (gx * nyz + gy * g.n_xyz[2] + fz) : -1;   // (floor == ceil duplicates just set the flag again)
    int b = valid ? a + (cz - fz) : -1;
    const int p0 = __shfl_up(a, 1), p1 = __shfl_up(b, 1);
    if (lane > 0 && p0 == a && p1 == b) continue;   // the previous lane flags the very same voxels
    if (a < 0) continue;
    bytemap[a] = 1;
    chunk_flag[a >> 6] = 1;
    if (b != a) {
      bytemap[b] = 1;
      if ((b >> 6) != (a >> 6)) chunk_flag[b >> 6] = 1;
    }
  }
  __shared__ int s_valid[4];
  const unsigned long long b = __ballot(valid);
  if (lane == 0) s_valid[threadIdx.x >> 6] = (int)__popcll(b);
  if (pair_list) {   // (workgroup-uniform)
    list_owned_pairs(valid, fx, cx, fy, cy, fz, cz, g, point_index, pair_list, n_pairs, orphan_list, n_orphans);
  } else {
    __syncthreads();
  }
  if (threadIdx.x == 0) valid_blocks[blockIdx.x] = s_valid[0] + s_valid[1] + s_valid[2] + s_valid[3];
}

__global__ __launch_bounds__(256) void k_mark(const float* __restrict__ pts, int n_points, bnv_grid_t g,
                                              uint8_t* __restrict__ bytemap, uint8_t* __restrict__ chunk_flag,
                                              int32_t* __restrict__ valid_blocks,
                                              int32_t* __restrict__ pair_list, int32_t* __restrict__ n_pairs,
                                              int32_t* __restrict__ orphan_list, int32_t* __restrict__ n_orphans) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool valid = false;
  float x = 0.f, y = 0.f, z = 0.f;
  if (i < n_points) {
    x = pts[(size_t)i * 6 + 0];
    y = pts[(size_t)i * 6 + 1];
    z = pts[(size_t)i * 6 + 2];
    valid = in_bounds(x, y, z, g);
  }
  mark_point(valid, x, y, z, g, bytemap, chunk_flag, valid_blocks, i, pair_list, n_pairs, orphan_list, n_orphans);
}

// The same, fused behind the depth front end (frontend.hpp): one thread per PIXEL computes the pixel's world point
// and normal in float64 as the reference's loader does, writes the float32 row of input_pts (NaN for an invalid
// pixel: rows stay in pixel order, nothing is compacted -- the encoder's bounds mask drops NaN rows wherever they
// are) and marks the point's voxels from the registers: the 7.4 MB of points are not read back, one launch less.
__global__ __launch_bounds__(256) void k_front_mark(FrontArgs a, float* __restrict__ out_pts, bnv_grid_t g,
                                                    uint8_t* __restrict__ bytemap, uint8_t* __restrict__ chunk_flag,
                                                    int32_t* __restrict__ valid_blocks,
                                                    int32_t* __restrict__ pair_list, int32_t* __restrict__ n_pairs,
                                                    int32_t* __restrict__ orphan_list,
                                                    int32_t* __restrict__ n_orphans) {
  const int64_t n = (int64_t)a.H * a.W;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float p[6];
  bool have = false;
  if (i < n) have = front_point(a, (int)(i / a.W), (int)(i % a.W), p);
  if (i < n) {
    float* o = out_pts + (size_t)i * 6;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      f32x2 v;
      v[0] = have ? p[2 * r] : __builtin_nanf("");
      v[1] = have ? p[2 * r + 1] : __builtin_nanf("");
      *(f32x2*)(o + 2 * r) = v;
    }
  }
  const bool valid = have && in_bounds(p[0], p[1], p[2], g);
  mark_point(valid, p[0], p[1], p[2], g, bytemap, chunk_flag, valid_blocks, (int)i, pair_list, n_pairs, orphan_list,
             n_orphans);
}

// ------------------------------------------------------------------------------------------
// rank: sorted-unique without a sort.  One pass over the chunk flags (decoupled look-back over the workgroups); a
// flagged chunk's 64 voxel bytes become two bitmap words (and are cleared, with the flag, for the next frame):
// bitmap = one bit per touched voxel, word_prefix[w] = set bits before word w, ids[] = the set bits in ascending
// order (= torch.unique's output), ctl->n_unique = their number.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bytes_to_bits16(const uint32_t (&v)[4]) {
  // 16 flag bytes (0 / 1) -> 16 bits: (b0 | b1 << 8 | b2 << 16 | b3 << 24) * 0x01020408 has b0..b3 in bits 24..27
  uint32_t r = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) r |= (((v[q] * 0x01020408u) >> 24) & 0xFu) << (4 * q);
  return r;
}

__global__ __launch_bounds__(kScanThreads) void k_rank(uint8_t* __restrict__ bytemap, uint8_t* __restrict__ chunk_flag,
                                                       int64_t n_chunks, uint64_t* __restrict__ tile_state,
                                                       uint32_t epoch, uint32_t* __restrict__ bitmap,
                                                       uint32_t* __restrict__ word_prefix,
                                                       int32_t* __restrict__ ids, int64_t max_unique,
                                                       EncCtl* __restrict__ ctl, bnv_grid_t g,
                                                       int32_t* __restrict__ defer_list) {
  __shared__ uint32_t wave_tot[kScanThreads / 64];
  __shared__ uint32_t s_excl;
  __shared__ int s_hist[64];
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const int64_t base = (int64_t)blockIdx.x * kRankTile + (int64_t)threadIdx.x * kRankItems;   // first chunk
  uint32_t flags = 0;
  if (base < n_chunks) flags = *(const uint32_t*)&chunk_flag[base];   // 4 chunk flags (n_chunks is a multiple of 4)
  uint32_t w[2 * kRankItems];
  uint32_t s = 0;
#pragma unroll
  for (int e = 0; e < kRankItems; ++e) {
    w[2 * e] = w[2 * e + 1] = 0u;
    if ((flags >> (8 * e)) & 0xffu) {
      u32x4* src = (u32x4*)&bytemap[(base + e) * 64];
      const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int hw = 0; hw < 2; ++hw) {
        const u32x4 lo = src[2 * hw], hi = src[2 * hw + 1];
        const uint32_t l4[4] = {lo[0], lo[1], lo[2], lo[3]}, h4[4] = {hi[0], hi[1], hi[2], hi[3]};
        w[2 * e + hw] = bytes_to_bits16(l4) | (bytes_to_bits16(h4) << 16);
        src[2 * hw] = z;       // consumed: clean for the next frame
        src[2 * hw + 1] = z;
      }
      s += __popc(w[2 * e]) + __popc(w[2 * e + 1]);
    }
  }
  if (flags) *(uint32_t*)&chunk_flag[base] = 0u;
  __shared__ int s_cur[64];
  if (g.shard_world > 1 && threadIdx.x < 64) {
    s_hist[threadIdx.x] = 0;
    s_cur[threadIdx.x] = 0;
  }
  uint32_t total;
  uint32_t run = block_exclusive_scan<kScanThreads>(s, wave_tot, &total);
  if (threadIdx.x < 64) {
    const uint32_t excl = lookback_exclusive(tile_state, (int)blockIdx.x, total, epoch);
    if (threadIdx.x == 0) {
      s_excl = excl;
      if (blockIdx.x == gridDim.x - 1) ctl->n_unique = (int32_t)(excl + total);
    }
  }
  __syncthreads();
  if (total == 0) return;  // nothing set in this tile: prefixes are never read for clear words
  run += s_excl;
  const int nyz = g.n_xyz[1] * g.n_xyz[2];
#pragma unroll
  for (int e = 0; e < 2 * kRankItems; ++e) {
    uint32_t bits = w[e];
    if (!bits) continue;
    const int64_t word = base * 2 + e;
    bitmap[word] = bits;
    word_prefix[word] = run;
    while (bits) {
      const int b = __ffs(bits) - 1;
      bits &= bits - 1;
      const int id = (int)(word * 32 + b);
      if (run < max_unique) ids[run] = id;
      else ctl->error = 1;
      ++run;
    }
  }
  if (g.shard_world > 1 && g.shard_state) {
    // first-touch ownership: the touched voxels of every block no frame has touched before are counted -- the
    // block's weight when k_shard_assign gives it an owner; the exchange bound follows in k_shard_own
    __syncthreads();
    ShardState S;
    shard_state_layout(g.n_xyz, g.shard_block_log2, (char*)g.shard_state, &S);
    const int sh = g.shard_block_log2, mb = (1 << sh) - 1;
    const int nby = (g.n_xyz[1] + mb) >> sh, nbz = (g.n_xyz[2] + mb) >> sh;
    const int64_t lo = s_excl, hi = (int64_t)s_excl + total < max_unique ? (int64_t)s_excl + total : max_unique;
    for (int64_t j = lo + threadIdx.x; j < hi; j += kScanThreads) {
      const int id = __hip_atomic_load(&ids[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int x = id / nyz, r = id - x * nyz, y = r / g.n_xyz[2], z = r - y * g.n_xyz[2];
      const int b = ((x >> sh) * nby + (y >> sh)) * nbz + (z >> sh);
      const uint8_t tb = S.table[b];
      // the frame's load per rank with the owners as they stand (region rule: who may take new territory)
      if (tb & kOwnAssigned) atomicAdd(&s_cur[tb & kOwnRank], 1);
      if (!(tb & kOwnTouched) && atomicAdd(&S.blk_w[b], 1u) == 0u) {
        // the block's first voxel: list the block (k_shard_assign sorts the list; past its capacity it scans the
        // weight table instead, so the count alone is what matters then)
        const uint32_t pos = (uint32_t)atomicAdd(&S.hdr->any_new, 1);
        if (pos < kNewListCap) S.new_list[pos] = (uint32_t)b;
      }
      // the exchange bound with the owners as they stand; a voxel next to a block that has no owner yet is left
      // to k_shard_own (behind this frame's k_shard_assign).  A frame without a new block defers nothing.
      const int st = shard_boundary_state(x, y, z, g);
      if (st == 1) atomicAdd(&s_hist[voxel_owner(x, y, z, g) & 63], 1);
      else if (st == 2) defer_list[atomicAdd(&ctl->n_deferred, 1)] = (int32_t)j;
    }
    __syncthreads();
    if (threadIdx.x < 64 && threadIdx.x < g.shard_world) {
      if (s_hist[threadIdx.x]) atomicAdd(&ctl->shard_boundary[threadIdx.x], s_hist[threadIdx.x]);
      if (s_cur[threadIdx.x]) atomicAdd(&S.hdr->cur[threadIdx.x], (uint32_t)s_cur[threadIdx.x]);
    }
  } else if (g.shard_world > 1) {
    // the exchange bound: touched BOUNDARY voxels per owner.  The set bits sit in a few threads (a thread holds 256
    // consecutive voxels), so the ~30 ownership hashes of a boundary test are spread over the workgroup: it walks
    // the ids it has just written (its own stretch of the sorted list), one voxel per thread and step
    __syncthreads();
    const int64_t lo = s_excl, hi = (int64_t)s_excl + total < max_unique ? (int64_t)s_excl + total : max_unique;
    for (int64_t j = lo + threadIdx.x; j < hi; j += kScanThreads) {
      const int id = __hip_atomic_load(&ids[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (written by other lanes of this workgroup)
      const int x = id / nyz, r = id - x * nyz, y = r / g.n_xyz[2], z = r - y * g.n_xyz[2];
      if (shard_is_boundary(x, y, z, g)) atomicAdd(&s_hist[voxel_owner(x, y, z, g) & 63], 1);
    }
    __syncthreads();
    if (threadIdx.x < 64 && threadIdx.x < g.shard_world && s_hist[threadIdx.x])
      atomicAdd(&ctl->shard_boundary[threadIdx.x], s_hist[threadIdx.x]);
  }
}

// ------------------------------------------------------------------------------------------
// first-touch ownership (bnv_grid_t.shard_state): owners for the blocks this frame touches for the first time
// ------------------------------------------------------------------------------------------
// ONE small workgroup (256 threads, no LDS to speak of: it must find room beside the persistent MLP kernels of the
// other streams, which leave a CU one wave slot per SIMD and little else -- the first version, 1,024 threads, sat in the
// front stream for the whole of a table kernel in every frame); nothing to do (one load) in a frame without a new
// block.  (1) the new blocks in ascending block order: k_rank has listed them (the first voxel that touches a block
// appends it), a bitonic sort of that list in place -- or, when a frame brings more than the list holds (the first
// frame of a scene), an ordered compaction of the dense weight table; (2) one wave walks them: a block that has no
// owner yet goes to the rank with the least load so far (lowest rank on ties), a block that was pinned earlier as
// somebody's neighbour keeps its owner, and either way its weight joins that rank's load; (3) the neighbour blocks of
// the new blocks that still have no owner are pinned to the lattice rule; (4) the weights are cleared.  Every rank
// runs this on the same replicated voxelisation, so every rank's table is the same -- no communication.
// ---- region rule (BNV_SHARD_RULE_REGION; include/bnv_fusion.h: bnv_grid_t.shard_state; host restatement:
// distributed.OwnershipModel) ----------------------------------------------------------------------------------------
// Table bytes are read and written with relaxed agent-scope atomics and a fence behind every store: the walk reads
// entries it has written a few iterations earlier.
__device__ __forceinline__ uint8_t own_load(const uint8_t* t, int64_t i) {
  return __hip_atomic_load(&t[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void own_store(uint8_t* t, int64_t i, uint8_t v) {
  __hip_atomic_store(&t[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// least-loaded rank (lowest rank on ties); cur: lane r holds rank r's load
__device__ __forceinline__ int least_rank(uint32_t cur, int world) {
  const int lane = threadIdx.x & 63;
  unsigned long long key = lane < world ? (((unsigned long long)cur << 6) | (unsigned long long)lane) : ~0ull;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = __shfl_xor(key, d, 64);
    key = o < key ? o : key;
  }
  return (int)(key & 63ull);
}
// the first wave of k_shard_assign walks the frame's n_new blocks (new_list holds their WALK KEYS, ascending)
__device__ void shard_assign_region(const bnv_grid_t& g, const ShardState& S, uint32_t n_new, uint32_t n_touched) {
  const int lane = threadIdx.x & 63;
  const int world = g.shard_world, axis = S.hdr->axis;
  int nb3[3];
  shard_block_dims(g.n_xyz, g.shard_block_log2, nb3);
  uint32_t cur = lane < world ? S.hdr->cur[lane] : 0u;
  unsigned long long load = lane < world ? S.hdr->load[lane] : 0ull;
  const unsigned long long nt = n_touched;
  // neighbour this lane looks at (lanes 0..26; 13 = the block itself)
  const int ddx = lane / 9 - 1, ddy = (lane / 3) % 3 - 1, ddz = lane % 3 - 1;
  int recv = S.hdr->recv_p1 - 1;
  {
    const uint32_t cr = __shfl(cur, recv < 0 ? 0 : recv, 64);
    if (recv < 0 || (unsigned long long)cr * (unsigned)world >= nt) recv = least_rank(cur, world);
  }
  // (1) owners for the new blocks, in walk order
  for (uint32_t i = 0; i < n_new; ++i) {
    const uint32_t b = shard_walk_block(S.new_list[i], nb3, axis);
    const uint32_t w = S.blk_w[b];
    const uint8_t t = own_load(S.table, b);
    int r;
    if (t & kOwnAssigned) {
      r = (int)(t & kOwnRank);   // pinned earlier: cur counts its voxels already (k_rank)
    } else {
      const int bz = (int)(b % (uint32_t)nb3[2]), by = (int)((b / (uint32_t)nb3[2]) % (uint32_t)nb3[1]),
                bx = (int)(b / ((uint32_t)nb3[2] * (uint32_t)nb3[1]));
      const int x = bx + ddx, y = by + ddy, z = bz + ddz;
      uint8_t tv = 0;
      if (lane < 27 && (unsigned)x < (unsigned)nb3[0] && (unsigned)y < (unsigned)nb3[1] && (unsigned)z < (unsigned)nb3[2])
        tv = own_load(S.table, ((int64_t)x * nb3[1] + y) * nb3[2] + z);
      const int c = (int)(tv & kOwnRank);
      const uint32_t cc = __shfl(cur, c, 64);
      const bool cand = (tv & kOwnAssigned) && (unsigned long long)cc * (unsigned)world < nt;   // assigned and not full
      unsigned long long key = cand ? (((unsigned long long)cc << 6) | (unsigned long long)c) : ~0ull;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(key, d, 64);
        key = o < key ? o : key;
      }
      if (key != ~0ull) {
        r = (int)(key & 63ull);
      } else {
        const uint32_t cr = __shfl(cur, recv, 64);
        if ((unsigned long long)cr * (unsigned)world >= nt) recv = least_rank(cur, world);
        r = recv;
      }
      if (lane == r) cur += w;
    }
    if (lane == r) load += w;
    if (lane == 0) own_store(S.table, b, (uint8_t)(r | kOwnAssigned | kOwnTouched));
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  }
  // (2) the untouched neighbours of the new blocks are pinned: regions grow outwards
  {
    const uint32_t cr = __shfl(cur, recv, 64);
    if ((unsigned long long)cr * (unsigned)world * 8ull > 9ull * nt) recv = least_rank(cur, world);
  }
  for (uint32_t i = 0; i < n_new; ++i) {
    const uint32_t b = shard_walk_block(S.new_list[i], nb3, axis);
    int r = (int)(own_load(S.table, b) & kOwnRank);
    const uint32_t cr = __shfl(cur, r, 64);
    if ((unsigned long long)cr * (unsigned)world * 8ull > 9ull * nt) r = recv;   // overloaded: no pins for it
    const int bz = (int)(b % (uint32_t)nb3[2]), by = (int)((b / (uint32_t)nb3[2]) % (uint32_t)nb3[1]),
              bx = (int)(b / ((uint32_t)nb3[2] * (uint32_t)nb3[1]));
    const int x = bx + ddx, y = by + ddy, z = bz + ddz;
    if (lane == 13) S.blk_w[b] = 0u;
    else if (lane < 27 && (unsigned)x < (unsigned)nb3[0] && (unsigned)y < (unsigned)nb3[1] && (unsigned)z < (unsigned)nb3[2]) {
      const int64_t e = ((int64_t)x * nb3[1] + y) * nb3[2] + z;
      if (!(own_load(S.table, e) & kOwnAssigned)) own_store(S.table, e, (uint8_t)(r | kOwnAssigned));
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  }
  if (lane < world) S.hdr->load[lane] = load;
  if (lane == 0) S.hdr->recv_p1 = recv + 1;
}

__global__ __launch_bounds__(256) void k_shard_assign(bnv_grid_t g, const EncCtl* __restrict__ ctl) {
  ShardState S;
  shard_state_layout(g.n_xyz, g.shard_block_log2, (char*)g.shard_state, &S);
  const uint32_t n_listed = (uint32_t)S.hdr->any_new;     // (k_rank counts the new blocks in it)
  // region rule: contiguous regions keep a rank's load level only while the view stays put; a frame whose most loaded
  // rank carries more than 1.3 x its share of the touched voxels means the camera sweeps -- from then on new territory
  // is handed out by the greedy rule (fine interleave: every rank holds an even sample of any view).  Sticky.
  __shared__ int s_inter;
  if (threadIdx.x < 64) {
    const int lane_ = threadIdx.x;
    uint32_t c = lane_ < g.shard_world ? S.hdr->cur[lane_] : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint32_t o = __shfl_xor(c, d, 64);
      c = o > c ? o : c;
    }
    if (lane_ == 0) {
      int inter = S.hdr->interleave;
      if (!inter && S.hdr->rule == BNV_SHARD_RULE_REGION &&
          (unsigned long long)c * (unsigned)g.shard_world * 10ull > 13ull * (unsigned long long)(uint32_t)ctl->n_unique) {
        inter = 1;
        S.hdr->interleave = 1;
      }
      s_inter = inter;
    }
  }
  __syncthreads();
  if (n_listed == 0) {
    if (threadIdx.x < 64) S.hdr->cur[threadIdx.x] = 0u;   // (k_rank of the NEXT frame adds to it)
    return;
  }
  const bool region = S.hdr->rule == BNV_SHARD_RULE_REGION && !s_inter;
  int nbw[3];
  shard_block_dims(g.n_xyz, g.shard_block_log2, nbw);
  const int axis = region ? S.hdr->axis : 0;
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t s_n;
  const int lane = threadIdx.x & 63;
  uint32_t n_new;
  if (n_listed <= kNewListCap) {
    // bitonic sort of new_list[0, n_listed) padded with ~0 to the next power of two, in global memory (L2)
    uint32_t np2 = 1;
    while (np2 < n_listed) np2 <<= 1;
    if (axis != 0)   // the region rule walks in key order: sort the keys
      for (uint32_t i = threadIdx.x; i < n_listed; i += 256) S.new_list[i] = shard_walk_key(S.new_list[i], nbw, axis);
    for (uint32_t i = n_listed + threadIdx.x; i < np2; i += 256) S.new_list[i] = 0xffffffffu;
    __syncthreads();
    for (uint32_t k = 2; k <= np2; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t i = threadIdx.x; i < np2; i += 256) {
          const uint32_t l = i ^ j;
          if (l > i) {
            const uint32_t a = S.new_list[i], b = S.new_list[l];
            const bool up = (i & k) == 0;
            if ((a > b) == up) {
              S.new_list[i] = b;
              S.new_list[l] = a;
            }
          }
        }
        __syncthreads();
      }
    n_new = n_listed;
  } else {
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < S.n_blocks; b0 += 256) {
      const int64_t b = b0 + threadIdx.x;   // (a walk key: the list comes out in walk order)
      const uint32_t f = (b < S.n_blocks && S.blk_w[shard_walk_block((uint32_t)b, nbw, axis)] > 0) ? 1u : 0u;
      uint32_t tot;
      const uint32_t pos = block_exclusive_scan<256>(f, wave_tot, &tot);
      const uint32_t base = s_n;
      if (f) S.new_list[base + pos] = (uint32_t)b;
      __syncthreads();
      if (threadIdx.x == 0) s_n = base + tot;
      __syncthreads();
    }
    n_new = s_n;
  }
  const int world = g.shard_world;
  if (region) {
    if (threadIdx.x < 64) {
      shard_assign_region(g, S, n_new, (uint32_t)ctl->n_unique);
      S.hdr->cur[lane] = 0u;
      if (lane == 0) S.hdr->any_new = 0;
    }
    return;
  }
  if (threadIdx.x < 64) {
    unsigned long long load = lane < world ? S.hdr->load[lane] : 0ull;
    for (uint32_t i0 = 0; i0 < n_new; i0 += 64) {
      // 64 entries at a time in registers: the walk itself then touches no memory
      const uint32_t i = i0 + lane;
      uint32_t mb = 0, mw = 0, mt = 0;
      if (i < n_new) {
        mb = S.new_list[i];
        mw = S.blk_w[mb];
        mt = S.table[mb];
      }
      int mine = -1;
      const int cnt = (int)(n_new - i0 < 64u ? n_new - i0 : 64u);
      for (int k = 0; k < cnt; ++k) {
        const uint32_t w = __shfl(mw, k, 64), t = __shfl(mt, k, 64);
        int r;
        if (t & kOwnAssigned) {
          r = (int)(t & kOwnRank);
        } else {
          unsigned long long key = lane < world ? ((load << 6) | (unsigned long long)lane) : ~0ull;
#pragma unroll
          for (int d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_xor(key, d, 64);
            key = o < key ? o : key;
          }
          r = (int)(key & 63ull);
        }
        if (lane == r) load += w;
        if (lane == k) mine = r;
      }
      if (i < n_new) S.table[mb] = (uint8_t)(mine | kOwnAssigned | kOwnTouched);
    }
    if (lane < world) S.hdr->load[lane] = load;
  }
  __syncthreads();
  // every new block has its owner now; blocks around them that nobody has touched yet are pinned to the lattice rule
  int nb3[3];
  shard_block_dims(g.n_xyz, g.shard_block_log2, nb3);
  for (uint64_t q = threadIdx.x; q < (uint64_t)n_new * 27u; q += 256) {
    const uint32_t b = S.new_list[q / 27u];
    const int d = (int)(q % 27u);
    if (d == 13) {
      S.blk_w[b] = 0u;   // (4)
      continue;
    }
    const int bz = (int)(b % (uint32_t)nb3[2]), by = (int)((b / (uint32_t)nb3[2]) % (uint32_t)nb3[1]),
              bx = (int)(b / ((uint32_t)nb3[2] * (uint32_t)nb3[1]));
    const int x = bx + d / 9 - 1, y = by + (d / 3) % 3 - 1, z = bz + d % 3 - 1;
    if ((unsigned)x >= (unsigned)nb3[0] || (unsigned)y >= (unsigned)nb3[1] || (unsigned)z >= (unsigned)nb3[2]) continue;
    uint8_t* e = &S.table[((int64_t)x * nb3[1] + y) * nb3[2] + z];
    if (!(*e & kOwnAssigned)) *e = (uint8_t)(shard_lattice_owner(x, y, z, world) | kOwnAssigned);   // (same value from every writer)
  }
  if (threadIdx.x == 0) S.hdr->any_new = 0;
  if (threadIdx.x < 64) S.hdr->cur[threadIdx.x] = 0u;
}

// With the owners of the frame's blocks known: the owned-pair list of the encoder (what the mark kernel does itself
// under the hash rule) and the exchange bound (what k_rank does itself under the hash rule).
__global__ __launch_bounds__(256) void k_shard_own(const float* __restrict__ pts, int n_points, bnv_grid_t g,
                                                   int32_t* __restrict__ pair_list, int32_t* __restrict__ n_pairs,
                                                   const int32_t* __restrict__ orphan_list,
                                                   const int32_t* __restrict__ ids,
                                                   const int32_t* __restrict__ defer_list,
                                                   EncCtl* __restrict__ ctl) {
  if (pair_list) {
    // the points the mark kernel left undecided (a corner voxel in a block without an owner at that time)
    const int n_orph = ctl->n_orphans;
    for (int pb = blockIdx.x; pb * 256 < n_orph; pb += gridDim.x) {   // (workgroup-uniform trip count)
      const int o = pb * 256 + threadIdx.x;
      const int i = o < n_orph ? orphan_list[o] : n_points;
      bool valid = false;
      int fx = 0, cx = 0, fy = 0, cy = 0, fz = 0, cz = 0;
      if (i < n_points) {
        const float x = pts[(size_t)i * 6 + 0], y = pts[(size_t)i * 6 + 1], z = pts[(size_t)i * 6 + 2];
        valid = in_bounds(x, y, z, g);
        if (valid) {
          const float xn = voxel_coord(x, g.bound_min[0], g.voxel_size);
          const float yn = voxel_coord(y, g.bound_min[1], g.voxel_size);
          const float zn = voxel_coord(z, g.bound_min[2], g.voxel_size);
          fx = (int)floorf(xn), cx = (int)ceilf(xn);
          fy = (int)floorf(yn), cy = (int)ceilf(yn);
          fz = (int)floorf(zn), cz = (int)ceilf(zn);
        }
      }
      list_owned_pairs(valid, fx, cx, fy, cy, fz, cz, g, i, pair_list, n_pairs);
      __syncthreads();   // s_cnt is reused by the next block of points
    }
  }
  __shared__ int s_hist[64];
  if (threadIdx.x < 64) s_hist[threadIdx.x] = 0;
  __syncthreads();
  // the touched voxels whose boundary test k_rank had to leave open
  const int64_t n = ctl->n_deferred;
  const int nyz = g.n_xyz[1] * g.n_xyz[2];
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256) {
    const int id = ids[defer_list[q]];
    const int x = id / nyz, r = id - x * nyz, y = r / g.n_xyz[2], z = r - y * g.n_xyz[2];
    if (shard_is_boundary(x, y, z, g)) atomicAdd(&s_hist[voxel_owner(x, y, z, g) & 63], 1);
  }
  __syncthreads();
  if (threadIdx.x < 64 && threadIdx.x < g.shard_world && s_hist[threadIdx.x])
    atomicAdd(&ctl->shard_boundary[threadIdx.x], s_hist[threadIdx.x]);
}

struct ValidFlags {  // 1 where the voxel in slot s is emitted
  const int32_t* counts;
  const int32_t* ids;
  bnv_grid_t g;
  int emit_all;
  __device__ uint32_t operator()(int64_t s) const {
    if (!emit_all && counts[s] < g.min_pts_in_grid) return 0;
    if (g.shard_world > 1) {
      const int id = ids[s];
      const int nyz = g.n_xyz[1] * g.n_xyz[2];
      const int x = id / nyz, r = id - x * nyz, y = r / g.n_xyz[2], z = r - y * g.n_xyz[2];
      if (voxel_owner(x, y, z, g) != g.shard_rank) return 0;
    }
    return 1;
  }
};

// Tiles of the point encoder: 32 (point, corner) pairs.  Unsharded: tile t = corner t / n_pblocks of the 32 consecutive
// points of block t % n_pblocks.  Sharded: 32 consecutive entries of the owned-pair list the mark kernel built.
struct PairTiles {
  const int32_t* list;   // null: unsharded
  int n_pairs, n_points, n_pblocks, n_tiles;
};
__device__ __forceinline__ PairTiles pair_tiles(int n_points, const int32_t* __restrict__ pair_list,
                                                const int32_t* __restrict__ n_pairs) {
  PairTiles T;
  T.list = pair_list;
  T.n_points = n_points;
  T.n_pblocks = (n_points + 31) >> 5;
  T.n_pairs = pair_list ? *n_pairs : 0;
  T.n_tiles = pair_list ? (T.n_pairs + 31) >> 5 : T.n_pblocks * 8;
  return T;
}
// pair j of tile t -> point index and corner; false past the end
__device__ __forceinline__ bool tile_pair(const PairTiles& T, int t, int j, int* i, int* k) {
  if (T.list) {
    const int e = t * 32 + j;
    if (e >= T.n_pairs) return false;
    const int p = T.list[e];
    *i = p >> 3;
    *k = p & 7;
    return true;
  }
  *k = t / T.n_pblocks;
  *i = (t - *k * T.n_pblocks) * 32 + j;
  return *i < T.n_points;
}

// ------------------------------------------------------------------------------------------
// k_pointnet_scatter
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x16 relu16(f32x16 v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = relu_bits(v[r]);
  return v;
}

__device__ __forceinline__ f32x16 bias_init(const float* __restrict__ b, int mb, int h) {
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 t = *(const f32x4*)&b[mb * 32 + 8 * q + 4 * h];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[4 * q + i] = t[i];
  }
  return v;
}

// 128 -> 128 layer: out[mb] += W[mb][nb] * in[nb]
__device__ __forceinline__ void layer128(const float* __restrict__ wp, const float* __restrict__ bias,
                                         const f32x16 (&in)[4], f32x16 (&out)[4], int lane, int h) {
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) out[mb] = bias_init(bias, mb, h);
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      f32x4 a[4];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
        a[mb] = *(const f32x4*)&wp[(((mb * 4 + nb) * 4 + rq) * 64 + lane) * 4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
          out[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb][i], in[nb][4 * rq + i], out[mb], 0, 0, 0);
      }
    }
  }
}

// Scatter of one tile: lane (j, h) holds output features 4h .. 4h+3 of pair j.  Consecutive pairs are
// neighbouring pixels and mostly fall into the same voxel, so the tile's values are summed per RUN of equal slots
// and only the last lane of a run issues the atomics (bit-identical to per-pair atomics: the sums are integers).
// The kernel is bound by instruction issue, not by the MFMA pipe (tools/phase_prof.py), so
// this is written for instruction count:
//  * 2^32 fixed point in 8 VALU ops per value: r = rndne(f * 2^32) is an integer-valued float, hi = floor(r /
//    2^32), lo = r - hi * 2^32 (both exact) -- the same integer llrintf gives, without the generic f32 -> i64
//    conversion sequence;
//  * ONE unsegmented inclusive prefix sum P over the 32 lanes of a half (5 DPP steps: row_shr 1, 2, 4, 8 and
//    row_bcast:15; an add / add-with-carry pair per 64-bit value and step, no LDS crossbar traffic), then
//    run [s, e] = P[e] - P[s - 1] in modular arithmetic: lanes of other runs -- invalid ones included, whatever
//    they hold -- cancel exactly, so nothing is masked; one ds_bpermute per register fetches P[s - 1];
//  * the run's pair count is its length.
// (Round 1's segmented Hillis-Steele scan over ds_bpermute took ~300 instructions per tile; this takes ~110.)
// Used by the exact-fp32 encoder (lane = (pair j, feature half h)); the split modes use scatter_tile_x.
// -> for the LAST lane of every run of equal slots: v[q] = the run's sum of output 4 h + q (2^32 fixed point), len = its
// pair count; is_end tells whether this lane is such a lane (lanes with slot < 0 form runs too: the caller skips them)
__device__ __forceinline__ void tile_run_sums(const f32x16& o, int slot, int j, int h, unsigned long long (&v)[4],
                                              bool& is_end, int& len) {
  uint32_t lo[4], hi[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float r = __builtin_rintf(o[q] * kFixedScale);          // integer-valued; |r| < 2^63 for |feature| < 2^31
    const float hf = __builtin_floorf(r * (1.0f / kFixedScale));  // exact: a power-of-two scaling, then floor
    hi[q] = (uint32_t)(int)hf;
    lo[q] = (uint32_t)__builtin_fmaf(hf, -kFixedScale, r);        // exact, in [0, 2^32)
  }
  // inclusive prefix over the 32 lanes of each half.  DPP reads need two wait states behind the VALU write of
  // their source: every register is re-read eight instructions after it was written; s_nop 1 covers the entry.
#define BNV_SCAN_STEP(ctrl)                                                      \
  "v_add_co_u32_dpp %0, vcc, %0, %0 " ctrl "\n"                                  \
  "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " ctrl "\n"                            \
  "v_add_co_u32_dpp %2, vcc, %2, %2 " ctrl "\n"                                  \
  "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc " ctrl "\n"                            \
  "v_add_co_u32_dpp %4, vcc, %4, %4 " ctrl "\n"                                  \
  "v_addc_co_u32_dpp %5, vcc, %5, %5, vcc " ctrl "\n"                            \
  "v_add_co_u32_dpp %6, vcc, %6, %6 " ctrl "\n"                                  \
  "v_addc_co_u32_dpp %7, vcc, %7, %7, vcc " ctrl "\n"
  asm volatile("s_nop 1\n"
               BNV_SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
               BNV_SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
               BNV_SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
               BNV_SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
               BNV_SCAN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
               : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3])
               :
               : "vcc");
#undef BNV_SCAN_STEP
  // run geometry from the heads mask of the half: head = first lane of a run
  const int prev = __builtin_amdgcn_update_dpp(0, slot, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  const unsigned long long heads64 = __ballot(j == 0 || prev != slot);
  const uint32_t heads = h ? (uint32_t)(heads64 >> 32) : (uint32_t)heads64;
  const int s = 31 - __clz((int)(heads & (0xffffffffu >> (31 - j))));   // head of this lane's run (bit 0 is set)
  is_end = j == 31 || ((heads >> (j + 1)) & 1u);
  len = j - s + 1;
  const int src = (h * 32 + (s > 0 ? s - 1 : 0)) * 4;                      // lane holding P[s - 1]
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t plo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)lo[q]);
    const uint32_t phi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)hi[q]);
    v[q] = ((unsigned long long)hi[q] << 32) | lo[q];
    if (s > 0) v[q] -= ((unsigned long long)phi << 32) | plo;
  }
}

__device__ __forceinline__ void scatter_tile(const f32x16& o, int slot, int j, int h, int32_t* __restrict__ counts,
                                             long long* __restrict__ acc) {
  unsigned long long v[4];
  bool is_end;
  int len;
  tile_run_sums(o, slot, j, h, v, is_end, len);
#ifdef BNV_PROBE_NO_SCATTER   // development probe (tools/): what do the scatter atomics cost?  keeps 1 of 64 tiles' atomics
  if ((blockIdx.x & 63) != 0) return;
#endif
  if (slot >= 0 && is_end) {
    unsigned long long* dst = (unsigned long long*)acc + ((uint32_t)slot * 8u + 4u * (uint32_t)h);   // 32-bit index: no loop-invariant 64-bit VGPR pair
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(dst + q, v[q]);
    if (h == 0) atomicAdd(&counts[slot], len);
  }
}

__global__ __launch_bounds__(512, 2) void k_pointnet_scatter(
    const float* __restrict__ pts, int n_points, bnv_grid_t g, const float* __restrict__ wpack,
    const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ word_prefix,
    int32_t* __restrict__ counts, long long* __restrict__ acc, const int32_t* __restrict__ pair_list,
    const int32_t* __restrict__ n_pairs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // stage all packed weights into LDS once per workgroup (persistent grid)
  stage_to_lds<512>(wpack, lds, PN_TOTAL * 4);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  const PairTiles T = pair_tiles(n_points, pair_list, n_pairs);
  const int n_tiles = T.n_tiles;
  const int nyz = g.n_xyz[1] * g.n_xyz[2];

  for (int t = blockIdx.x * 8 + wave; t < n_tiles; t += gridDim.x * 8) {
    int i = 0, k = 0;
    const bool have = tile_pair(T, t, j, &i, &k);
    float in0 = 0.f, in1 = 0.f, in2 = 0.f;  // this lane's half of the 6 inputs: features 2s + h
    int slot = -1;
    bool valid = false;
    if (have) {
      const float* p = pts + (size_t)i * 6;
      const float x = p[0], y = p[1], z = p[2];
      valid = in_bounds(x, y, z, g);
      if (valid) {
        const float xn = voxel_coord(x, g.bound_min[0], g.voxel_size);
        const float yn = voxel_coord(y, g.bound_min[1], g.voxel_size);
        const float zn = voxel_coord(z, g.bound_min[2], g.voxel_size);
        const int gx = (k & 1) ? (int)ceilf(xn) : (int)floorf(xn);
        const int gy = (k & 2) ? (int)ceilf(yn) : (int)floorf(yn);
        const int gz = (k & 4) ? (int)ceilf(zn) : (int)floorf(zn);
        if (voxel_owner(gx, gy, gz, g) == g.shard_rank) {
          const uint32_t id = (uint32_t)(gx * nyz + gy * g.n_xyz[2] + gz);
          const uint32_t word = bitmap[id >> 5];
          slot = (int)(word_prefix[id >> 5] + __popc(word & ((1u << (id & 31)) - 1u)));
        }
        const float rx = relative_coord(xn, gx, g.voxel_size);
        const float ry = relative_coord(yn, gy, g.voxel_size);
        const float rz = relative_coord(zn, gz, g.voxel_size);
        // inputs [rx, ry, rz, nx, ny, nz]; K-step s contracts inputs (2s, 2s+1) = (h=0, h=1)
        in0 = h ? ry : rx;
        in1 = h ? p[3] : rz;
        in2 = h ? p[5] : p[4];
      }
    }
    // the tile is skipped when no lane contributes (wave-uniform branch)
    if (__ballot(slot >= 0) == 0ULL) continue;

    // ---- layer 1: 6 -> 128 --------------------------------------------------------------
    f32x16 ha[4], hb[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) ha[mb] = bias_init(lds + PN_B1, mb, h);
    {
      const float bin[3] = {in0, in1, in2};
#pragma unroll
      for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
          ha[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(lds[PN_W1 + (s * 4 + mb) * 64 + lane], bin[s],
                                                        ha[mb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) ha[mb] = relu16(ha[mb]);
    // ---- layers 2, 3: 128 -> 128 ---------------------------------------------------------
    layer128(lds + PN_W2, lds + PN_B2, ha, hb, lane, h);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) hb[mb] = relu16(hb[mb]);
    layer128(lds + PN_W3, lds + PN_B3, hb, ha, lane, h);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) ha[mb] = relu16(ha[mb]);
    // ---- layer 4: 128 -> 8 (rows 8..31 of the MFMA tile are zero padding) ------------------
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    {
      const f32x4 b4 = *(const f32x4*)&lds[PN_B4 + 4 * h];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = b4[r];
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (j < 8) a = *(const f32x4*)&lds[PN_W4 + ((((nb * 4 + rq) * 2 + h) * 8) + j) * 4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          o = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], ha[nb][4 * rq + i], o, 0, 0, 0);
      }
    }
    scatter_tile(o, slot, j, h, counts, acc);
  }
}


// ------------------------------------------------------------------------------------------
// Split-operand encoder (MLP modes 1 and 3): every fp32 operand is split into f16 hi + lo (x = hi + lo to ~22 bits;
// f16 subnormals are kept by the MFMA) and a.b ~ ah.bh + ah.bl + al.bh on the f16 MFMA with fp32 accumulation:
// fp32-class results at 16/3 x the fp32 MFMA rate (mode 3: ah.bh only).
// ------------------------------------------------------------------------------------------
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// LDS reads of the split-operand encoder go through a handful of OPAQUE 32-bit base addresses plus compile-time
// byte offsets that fit the 16-bit immediate of ds_read_b128.  Written as plain pointer arithmetic on the 150 KB
// weight image the compiler kept ~40 VGPRs of pre-added addresses alive across the tile loop (and spilled the
// staged point of the next tile for them); with three weight bases (lane * 16 + 0 / 60 KB / 120 KB) and one for
// the biases it keeps four.
typedef __attribute__((address_space(3))) const half8 lds_half8_t;
typedef __attribute__((address_space(3))) const f32x4 lds_f32x4_t;
constexpr int kLdsWin = 61440;   // span of one weight base (< 64 KB immediate range, multiple of 1024)

#ifdef BNV_PHASE_PROF
__device__ unsigned long long g_enc_phase[8 * 16];
#define BNV_EPH(i)                                                                          \
  do {                                                                                      \
    if ((threadIdx.x & 63) == 0) {                                                          \
      unsigned long long* _p = (unsigned long long*)((char*)lds + PX_LDS_BYTES) + (threadIdx.x >> 6) * 16; \
      const unsigned long long _t = clock64();                                              \
      _p[i] += _t - _p[15];                                                                 \
      _p[15] = _t;                                                                          \
    }                                                                                       \
  } while (0)
constexpr int kEncProfLds = 8 * 16 * 8;
#else
#define BNV_EPH(i)
constexpr int kEncProfLds = 0;
#endif

// ------------------------------------------------------------------------------------------
// k_pointnet_scatter_x: the split-operand encoder on v_mfma_f32_16x16x32_f16.
// The kernel runs at the package power limit (tools/power_probe.py) and under that limit the 16x16x32 form
// delivers ~14 % more FLOP/s than the 32x32x16 form (tools/probe_shapes.hip).  Same
// arithmetic (three products, fp32 accumulation), same bytes from LDS, another shape of a wave's tile:
//  * lane (n = l & 15, g = l >> 4); a tile is still 32 pairs = 2 COLUMN blocks of 16 (pair p = 16 cb + n) and a
//    128-wide layer is 8 ROW blocks of 16 features: 16 accumulators of 4 registers, register i of acc[rb][cb] =
//    feature 16 rb + 4 g + i of pair 16 cb + n;
//  * chaining: a K-step is 32 deep, operand slot jj of K-group g is K index 8 g + jj.  The eight registers
//    {acc[2 s][cb][0..3], acc[2 s + 1][cb][0..3]} of a lane are exactly its operand of K-step s of the next layer
//    for column block cb (slot jj <-> feature 32 s + 16 (jj >> 2) + 4 g + (jj & 3)): no cross-lane traffic
//    between the layers, as before.  The weights are packed to that order on the host (weights.py:
//    _pack_pointnet_split16, PX_* below);
//  * a pair is STAGED by the two lanes (n, 2 c) and (n, 2 c + 1) of its column block c (both need its slot for the
//    scatter: they scatter output features 0..3 and 4..7); the first layer's inputs live in K-group 0, so lanes
//    g = 0 take the six inputs of pair 16 + n from lane l + 32;
//  * the last layer is ONE row block (8 of 16 rows used) instead of one 32-row tile (8 of 32): half the MFMA work
//    of that layer; its outputs for column block 1 go back to the lanes that staged those pairs (lane l + 32);
//  * the scatter's prefix sums are row-local (a DPP row = a column block of a feature half: 4 steps) and joined
//    across the two column blocks through lane 15.
// ------------------------------------------------------------------------------------------
struct EncLdsX {
  uint32_t w[3];   // lane * 16 + kLdsWin * {0, 1, 2}
  uint32_t b;      // biases: 4 * g floats into the bias block
};
__device__ __forceinline__ half8 ldsx_wfrag(const EncLdsX& L, int byte_off) {
  const int b = byte_off / kLdsWin;
  return *(lds_half8_t*)((b == 0 ? L.w[0] : (b == 1 ? L.w[1] : L.w[2])) + (uint32_t)(byte_off - b * kLdsWin));
}
__device__ __forceinline__ f32x4 ldsx_bias(const EncLdsX& L, int layer, int rb) {
  return *(lds_f32x4_t*)(L.b + (uint32_t)((layer * 128 + rb * 16) * 4));
}

// 128 -> 128 layer: 32 steps q = (K-step s = q >> 3, row block rb = q & 7), six MFMAs per step (two column blocks
// x three products; the two chains of a step alternate), the weight fragments of step q + 1 fetched before them
template <int NPROD>
__device__ __forceinline__ void layer128_x(const EncLdsX& L, int w_off, int layer, const half8 (&inh)[4][2],
                                           const half8 (&inl)[4][2], f32x4 (&out)[8][2]) {
#pragma unroll
  for (int rb = 0; rb < 8; ++rb) out[rb][0] = out[rb][1] = ldsx_bias(L, layer, rb);
  half8 ah[2], al[2];
#define BNV_LOAD_WX(q)                                                        \
  {                                                                           \
    const int wb = (w_off + (q) * 2 * 64 * 8) * 2;                            \
    ah[(q) & 1] = ldsx_wfrag(L, wb);                                          \
    if (NPROD == 3) al[(q) & 1] = ldsx_wfrag(L, wb + 1024);                   \
  }
  BNV_LOAD_WX(0);
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    if (q + 1 < 32) BNV_LOAD_WX(q + 1);
    __builtin_amdgcn_sched_barrier(0);
    const int s = q >> 3, rb = q & 7;
    if constexpr (NPROD == 3) {
      out[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[q & 1], inh[s][0], out[rb][0], 0, 0, 0);
      out[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[q & 1], inh[s][1], out[rb][1], 0, 0, 0);
      out[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q & 1], inl[s][0], out[rb][0], 0, 0, 0);
      out[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q & 1], inl[s][1], out[rb][1], 0, 0, 0);
    }
    out[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q & 1], inh[s][0], out[rb][0], 0, 0, 0);
    out[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q & 1], inh[s][1], out[rb][1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef BNV_LOAD_WX
}

// ReLU + hi/lo split of a layer's accumulators into the next layer's operands
template <int NPROD>
__device__ __forceinline__ void split_x(const f32x4 (&acc)[8][2], half8 (&oh)[4][2], half8 (&ol)[4][2]) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = relu_bits(acc[2 * s + (e >> 2)][cb][e & 3]);
      if (NPROD == 3) {
        split8_f16(x, oh[s][cb], ol[s][cb]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) oh[s][cb][e] = (_Float16)x[e];
      }
    }
}

// Scatter of one tile: this lane holds output features 4 fh .. 4 fh + 3 (fh = g & 1) of pair p = 16 (g >> 1) + n: the
// 32 pairs of a feature half are DPP rows fh and fh + 2.  Same scheme as scatter_tile (run sums = differences of ONE
// inclusive prefix sum over the 32 pairs, exact in modular arithmetic; the run's pair count is its length): row-local
// prefix sums (4 DPP steps), then rows 2 and 3 add the totals of rows 0 and 1 (lane 15 of those rows).
__device__ __forceinline__ void scatter_tile_x(const f32x4& o, int slot, int n, int g, int32_t* __restrict__ counts,
                                               long long* __restrict__ acc) {
  uint32_t lo[4], hi[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float r = __builtin_rintf(o[q] * kFixedScale);
    const float hf = __builtin_floorf(r * (1.0f / kFixedScale));
    hi[q] = (uint32_t)(int)hf;
    lo[q] = (uint32_t)__builtin_fmaf(hf, -kFixedScale, r);
  }
#define BNV_SCAN_STEP(ctrl)                                                      \
  "v_add_co_u32_dpp %0, vcc, %0, %0 " ctrl "\n"                                  \
  "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc " ctrl "\n"                            \
  "v_add_co_u32_dpp %2, vcc, %2, %2 " ctrl "\n"                                  \
  "v_addc_co_u32_dpp %3, vcc, %3, %3, vcc " ctrl "\n"                            \
  "v_add_co_u32_dpp %4, vcc, %4, %4 " ctrl "\n"                                  \
  "v_addc_co_u32_dpp %5, vcc, %5, %5, vcc " ctrl "\n"                            \
  "v_add_co_u32_dpp %6, vcc, %6, %6 " ctrl "\n"                                  \
  "v_addc_co_u32_dpp %7, vcc, %7, %7, vcc " ctrl "\n"
  asm volatile("s_nop 1\n"
               BNV_SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
               BNV_SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
               BNV_SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
               BNV_SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
               : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3])
               :
               : "vcc");
#undef BNV_SCAN_STEP
  {   // rows 2, 3 (pairs 16..31): + the total of pairs 0..15 of the same feature half (lane 15 of row g - 2)
    const int tsrc = ((g & 1) * 16 + 15) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t tl = (uint32_t)__builtin_amdgcn_ds_bpermute(tsrc, (int)lo[q]);
      const uint32_t th = (uint32_t)__builtin_amdgcn_ds_bpermute(tsrc, (int)hi[q]);
      if (g >= 2) {
        const unsigned long long v = (((unsigned long long)hi[q] << 32) | lo[q]) + (((unsigned long long)th << 32) | tl);
        lo[q] = (uint32_t)v;
        hi[q] = (uint32_t)(v >> 32);
      }
    }
  }
  const int p = (g >> 1) * 16 + n;
  const int slot15 = __builtin_amdgcn_readlane(slot, 15);                     // pair 15 (lanes 15 and 31 stage it)
  const int prev_row = __builtin_amdgcn_update_dpp(0, slot, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
  const int prev = n == 0 ? slot15 : prev_row;
  const unsigned long long heads64 = __ballot(p == 0 || prev != slot);
  const uint32_t heads = ((uint32_t)heads64 & 0xffffu) | (((uint32_t)(heads64 >> 32) & 0xffffu) << 16);   // rows 0 and 2
  const int s = 31 - __clz((int)(heads & (0xffffffffu >> (31 - p))));         // head of this lane's run (bit 0 is set)
  const bool is_end = p == 31 || ((heads >> (p + 1)) & 1u);
  const int sp = s > 0 ? s - 1 : 0;                                            // pair holding P[s - 1]
  const int src = (((sp >> 4) * 2 + (g & 1)) * 16 + (sp & 15)) * 4;
  uint32_t plo[4], phi[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    plo[q] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)lo[q]);
    phi[q] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)hi[q]);
  }
#ifdef BNV_PROBE_NO_SCATTER   // development probe (tools/enc_time.py): keeps 1 of 64 workgroups' atomics
  if ((blockIdx.x & 63) != 0) return;
#endif
  if (slot >= 0 && is_end) {
    unsigned long long* dst = (unsigned long long*)acc + ((uint32_t)slot * 8u + 4u * (uint32_t)(g & 1));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned long long v = ((unsigned long long)hi[q] << 32) | lo[q];
      if (s > 0) v -= ((unsigned long long)phi[q] << 32) | plo[q];
      atomicAdd(dst + q, v);
    }
    if ((g & 1) == 0) atomicAdd(&counts[slot], p - s + 1);
  }
}

template <int NPROD>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(120))) void k_pointnet_scatter_x(
    const float* __restrict__ pts, int n_points, bnv_grid_t g, const float* __restrict__ wpack,
    const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ word_prefix,
    int32_t* __restrict__ counts, long long* __restrict__ acc, int32_t* __restrict__ error,
    const int32_t* __restrict__ pair_list, const int32_t* __restrict__ n_pairs) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const float n_cert = wpack[PN_CERT];
  float* lb = lds + PX_TOTAL / 2;                      // b1 b2 b3 b4
  for (int i = threadIdx.x; i < 128 * 3 + 8; i += 512) lb[i] = wpack[PN_B1 + i];
  stage_to_lds<512>(wpack + PX_OFF, lds, PX_TOTAL * 2);
  int* tile_ctr = (int*)((char*)lds + PX_TOTAL * 2 + (128 * 3 + 8) * 4);
  if (threadIdx.x == 0) *tile_ctr = 0;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int n = lane & 15, gk = lane >> 4;          // K-group / accumulator row group
  const int pair = (gk >> 1) * 16 + n;              // the pair this lane stages and scatters
  const PairTiles T = pair_tiles(n_points, pair_list, n_pairs);
  const int n_tiles = T.n_tiles;
  const int nyz = g.n_xyz[1] * g.n_xyz[2];
  EncLdsX L;
  {
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    L.w[0] = lds0 + lane * 16;
    L.w[1] = L.w[0] + kLdsWin;
    L.w[2] = L.w[0] + 2 * kLdsWin;
    L.b = lds0 + PX_TOTAL * 2 + gk * 16;
    asm volatile("" : "+v"(L.w[0]), "+v"(L.w[1]), "+v"(L.w[2]), "+v"(L.b));
  }

  // Software pipeline over this wave's tiles: while tile t runs its MLP, the point of tile t+2 and the
  // bitmap / prefix words of tile t+1 are in flight (three dependent memory latencies per tile).
  // The workgroup's tiles {8 b + k + i * 8 * gridDim} are handed to its 8 waves DYNAMICALLY (LDS counter):
  // of the two waves on a SIMD the older one wins issue arbitration and runs ~1.4x faster, so with a
  // static split the younger waves were still working when the older ones had finished
  // (tools/phase_prof.py).  The scatter is order-independent, so results do not depend on who takes what.
  const int tstep = gridDim.x * 8;
  auto grab = [&]() -> int {
    int c = 0;
    if (lane == 0) c = atomicAdd(tile_ctr, 1);
    c = __builtin_amdgcn_readfirstlane(c);
    return blockIdx.x * 8 + (c & 7) + (c >> 3) * tstep;
  };
  float raw[6];                 // stage 1 (tile t+2): the raw point and its corner
  int raw_k = 0;
  bool raw_ok = false;
  float nin[6];                 // stage 2 (tile t+1): network inputs (lanes of even g), voxel id, bitmap / prefix words
  uint32_t n_id = 0, n_word = 0, n_pref = 0;
  bool n_own = false;
  auto stage1 = [&](int t) {
    raw_ok = false;
    if (t < n_tiles) {
      int i = 0;
      if (tile_pair(T, t, pair, &i, &raw_k)) {
        const float* p = pts + (size_t)i * 6;
#pragma unroll
        for (int c = 0; c < 6; ++c) raw[c] = p[c];
        raw_ok = true;
      }
    }
  };
  auto stage2 = [&](int t) {
    n_own = false;
#pragma unroll
    for (int c = 0; c < 6; ++c) nin[c] = 0.f;
    if (raw_ok && in_bounds(raw[0], raw[1], raw[2], g)) {
      const int k = raw_k;
      const float xn = voxel_coord(raw[0], g.bound_min[0], g.voxel_size);
      const float yn = voxel_coord(raw[1], g.bound_min[1], g.voxel_size);
      const float zn = voxel_coord(raw[2], g.bound_min[2], g.voxel_size);
      const int gx = (k & 1) ? (int)ceilf(xn) : (int)floorf(xn);
      const int gy = (k & 2) ? (int)ceilf(yn) : (int)floorf(yn);
      const int gz = (k & 4) ? (int)ceilf(zn) : (int)floorf(zn);
      if (voxel_owner(gx, gy, gz, g) == g.shard_rank) {
        n_own = true;
        n_id = (uint32_t)(gx * nyz + gy * g.n_xyz[2] + gz);
        n_word = bitmap[n_id >> 5];
        n_pref = word_prefix[n_id >> 5];
      }
      if ((gk & 1) == 0) {
        nin[0] = relative_coord(xn, gx, g.voxel_size);
        nin[1] = relative_coord(yn, gy, g.voxel_size);
        nin[2] = relative_coord(zn, gz, g.voxel_size);
        nin[3] = raw[3];
        nin[4] = raw[4];
        nin[5] = raw[5];
      }
      if (!(fmaxf(fmaxf(fabsf(raw[3]), fabsf(raw[4])), fabsf(raw[5])) <= n_cert)) *error = 3;
    }
  };
  int t = grab();
  stage1(t);
  stage2(t);
  int t_next = grab(), t_next2 = 0;
  stage1(t_next);
#ifdef BNV_PHASE_PROF
  if ((threadIdx.x & 63) < 16)
    ((unsigned long long*)((char*)lds + PX_LDS_BYTES))[(threadIdx.x >> 6) * 16 + (threadIdx.x & 63)] = 0;
  if ((threadIdx.x & 63) == 0) ((unsigned long long*)((char*)lds + PX_LDS_BYTES))[(threadIdx.x >> 6) * 16 + 15] = clock64();
#endif

  for (; t < n_tiles; t = t_next, t_next = t_next2) {
    float in[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) in[c] = nin[c];
    const int slot = n_own ? (int)(n_pref + __popc(n_word & ((1u << (n_id & 31)) - 1u))) : -1;
    stage2(t_next);
    t_next2 = grab();
    stage1(t_next2);
    __builtin_amdgcn_sched_barrier(0);
    BNV_EPH(0);
    if (__ballot(slot >= 0) == 0ULL) continue;

    // ---- layer 1: 6 -> 128, one K-step of 32 (inputs in K-group 0: slots 0..5 of lanes g = 0) --------------
    f32x4 ha[8][2], hb[8][2];
    {
      half8 bh[2], bl[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.f;
#pragma unroll
        for (int e = 0; e < 6; ++e) {
          const float other = __shfl(in[e], (lane + 32) & 63, 64);   // pair 16 + n is staged by lane l + 32
          x[e] = gk == 0 ? (cb == 0 ? in[e] : other) : 0.f;
        }
        if (NPROD == 3) {
          split8_f16(x, bh[cb], bl[cb]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) bh[cb][e] = (_Float16)x[e];
        }
      }
#pragma unroll
      for (int rb = 0; rb < 8; ++rb) {
        const half8 ahi = ldsx_wfrag(L, (PX_W1 + rb * 2 * 64 * 8) * 2), alo = ldsx_wfrag(L, (PX_W1 + rb * 2 * 64 * 8) * 2 + 1024);
        const f32x4 b = ldsx_bias(L, 0, rb);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          f32x4 c = b;
          if constexpr (NPROD == 3) {
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bh[cb], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bl[cb], c, 0, 0, 0);
          }
          ha[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bh[cb], c, 0, 0, 0);
        }
      }
    }
    BNV_EPH(1);
    half8 sh[4][2], sl[4][2];
    split_x<NPROD>(ha, sh, sl);
    BNV_EPH(2);
    layer128_x<NPROD>(L, PX_W2, 1, sh, sl, hb);
    BNV_EPH(3);
    split_x<NPROD>(hb, sh, sl);
    BNV_EPH(4);
    layer128_x<NPROD>(L, PX_W3, 2, sh, sl, ha);
    BNV_EPH(5);
    split_x<NPROD>(ha, sh, sl);
    BNV_EPH(6);
    // ---- layer 4: 128 -> 8, one row block (rows >= 8 are zero weights); rows 4 g + i of lanes g >= 2 are unused
    f32x4 o[2];
    o[0] = o[1] = *(lds_f32x4_t*)(L.b + 384 * 4);
    {
      half8 w4h[4], w4l[4];
      __builtin_amdgcn_sched_barrier(0);  // keep these loads below layer 3 (register peak)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        w4h[s] = ldsx_wfrag(L, (PX_W4 + s * 2 * 64 * 8) * 2);
        if (NPROD == 3) w4l[s] = ldsx_wfrag(L, (PX_W4 + s * 2 * 64 * 8) * 2 + 1024);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          if constexpr (NPROD == 3) {
            o[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w4l[s], sh[s][cb], o[cb], 0, 0, 0);
            o[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w4h[s], sl[s][cb], o[cb], 0, 0, 0);
          }
          o[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w4h[s], sh[s][cb], o[cb], 0, 0, 0);
        }
      }
    }
    // outputs of column block 1 back to the lanes that staged those pairs: lane (n, g) with g >= 2 takes features
    // 4 (g & 1) .. + 3 of pair 16 + n from lane l - 32
    f32x4 mine;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float other = __shfl(o[1][i], (lane + 32) & 63, 64);
      mine[i] = gk < 2 ? o[0][i] : other;
    }
    BNV_EPH(7);
    scatter_tile_x(mine, slot, n, gk, counts, acc);
    BNV_EPH(8);
  }
#ifdef BNV_PHASE_PROF
  if ((threadIdx.x & 63) < 15)
    atomicAdd(&g_enc_phase[(threadIdx.x >> 6) * 16 + (threadIdx.x & 63)],
              ((unsigned long long*)((char*)lds + PX_LDS_BYTES))[(threadIdx.x >> 6) * 16 + (threadIdx.x & 63)]);
#endif
}

// ------------------------------------------------------------------------------------------
// k_pointnet_scatter_t: the tiny-cuda-nn point encoder of the reference's default checkpoint
// (pointnet_tcnn.ckpt; tcnnPointNetEncoder, pointnet_utils.py:269-294; FullyFusedMLP per
// src/models/tcnn_config.json): 6 inputs padded to 16 with 1.0 -> 64 -> 64 -> 64 -> 16 (first 8 used),
// ReLU, no bias, fp16 weights and activations.  Here: f16 MFMA with fp32 accumulation, activations
// rounded to f16 between layers and at the output, as the CUDA kernel stores them.
// Pack (halves): W1 [2 mb][64 lane][8] | W2, W3 [2 mb][4 g][64][8] | W4 [4 g][64][8] (rows >= 16 zero).
// ------------------------------------------------------------------------------------------
constexpr int PT_W1 = 0;
constexpr int PT_W2 = PT_W1 + 2 * 64 * 8;
constexpr int PT_W3 = PT_W2 + 2 * 4 * 64 * 8;
constexpr int PT_W4 = PT_W3 + 2 * 4 * 64 * 8;
constexpr int PT_TOTAL = PT_W4 + 4 * 64 * 8;  // 11,264 halves = 22,528 B

__device__ __forceinline__ half8 to_half8_relu(const f32x16& v, int base) {
  half8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    r[e] = (_Float16)relu_bits(v[base + e]);
  }
  return r;
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = 0.f;
  return v;
}

__global__ __launch_bounds__(256) void k_pointnet_scatter_t(
    const float* __restrict__ pts, int n_points, bnv_grid_t g, const float* __restrict__ wpack,
    const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ word_prefix,
    int32_t* __restrict__ counts, long long* __restrict__ acc, const int32_t* __restrict__ pair_list,
    const int32_t* __restrict__ n_pairs) {
  __shared__ __attribute__((aligned(16))) _Float16 wh[PT_TOTAL];
  stage_to_lds<256>(wpack, wh, PT_TOTAL * 2);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h = lane >> 5;
  const PairTiles T = pair_tiles(n_points, pair_list, n_pairs);
  const int n_tiles = T.n_tiles;
  const int nyz = g.n_xyz[1] * g.n_xyz[2];
  for (int t = blockIdx.x * 4 + wave; t < n_tiles; t += gridDim.x * 4) {
    int i = 0, k = 0;
    const bool have = tile_pair(T, t, j, &i, &k);
    // operand slots of this lane half: features 8 (jj >> 2) + 4 h + (jj & 3); inputs 0..5, the rest 1.0
    half8 b;
#pragma unroll
    for (int e = 0; e < 8; ++e) b[e] = (_Float16)1.0f;
    int slot = -1;
    bool valid = false;
    if (have) {
      const float* p = pts + (size_t)i * 6;
      const float x = p[0], y = p[1], z = p[2];
      if (in_bounds(x, y, z, g)) {
        valid = true;
        const float xn = voxel_coord(x, g.bound_min[0], g.voxel_size);
        const float yn = voxel_coord(y, g.bound_min[1], g.voxel_size);
        const float zn = voxel_coord(z, g.bound_min[2], g.voxel_size);
        const int gx = (k & 1) ? (int)ceilf(xn) : (int)floorf(xn);
        const int gy = (k & 2) ? (int)ceilf(yn) : (int)floorf(yn);
        const int gz = (k & 4) ? (int)ceilf(zn) : (int)floorf(zn);
        if (voxel_owner(gx, gy, gz, g) == g.shard_rank) {
          const uint32_t id = (uint32_t)(gx * nyz + gy * g.n_xyz[2] + gz);
          const uint32_t word = bitmap[id >> 5];
          slot = (int)(word_prefix[id >> 5] + __popc(word & ((1u << (id & 31)) - 1u)));
        }
        if (h == 0) {
          b[0] = (_Float16)relative_coord(xn, gx, g.voxel_size);
          b[1] = (_Float16)relative_coord(yn, gy, g.voxel_size);
          b[2] = (_Float16)relative_coord(zn, gz, g.voxel_size);
          b[3] = (_Float16)p[3];
        } else {
          b[0] = (_Float16)p[4];
          b[1] = (_Float16)p[5];
        }
      }
    }
    if (__ballot(slot >= 0) == 0ULL) continue;
    (void)valid;
    f32x16 ha[2], hb[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
      ha[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&wh[PT_W1 + (mb * 64 + lane) * 8], b, zero16(),
                                                      0, 0, 0);
    half8 s[4];
    auto layer64 = [&](int woff, const f32x16 (&in)[2], f32x16 (&out)[2]) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        s[nb * 2] = to_half8_relu(in[nb], 0);
        s[nb * 2 + 1] = to_half8_relu(in[nb], 8);
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        out[mb] = zero16();
#pragma unroll
        for (int gk = 0; gk < 4; ++gk)
          out[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&wh[woff + ((mb * 4 + gk) * 64 + lane) * 8],
                                                          s[gk], out[mb], 0, 0, 0);
      }
    };
    layer64(PT_W2, ha, hb);
    layer64(PT_W3, hb, ha);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      s[nb * 2] = to_half8_relu(ha[nb], 0);
      s[nb * 2 + 1] = to_half8_relu(ha[nb], 8);
    }
    f32x16 o = zero16();
#pragma unroll
    for (int gk = 0; gk < 4; ++gk)
      o = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&wh[PT_W4 + (gk * 64 + lane) * 8], s[gk], o, 0, 0, 0);
    // the network returns fp16; lane (j, h) holds outputs 4h .. 4h+3 of pair j
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = (float)(_Float16)o[q];
    scatter_tile(o, slot, j, h, counts, acc);
  }
}

// ------------------------------------------------------------------------------------------
// k_pointnet_scatter_tb: the tiny-cuda-nn encoder for WHOLE frames (unsharded encode), built around the scatter.
// With this small network the kernel's floor was its global atomics: 5.7 M device-scope 64-bit atomics per frame
// (366 MB of write traffic tallied at 64 B each) = 0.31 ms whatever the weights, against 0.22 ms without them
// (profiles/r02_power_probe.txt, r02_bench_line_tcnn.json).  Here a wave's unit of work is a BLOCK of 32 points --
// 8 x 4 pixels of the depth image when the frame's width is known, else 32 consecutive points -- with all EIGHT
// corner tiles of those points, and the per-voxel sums are formed in an LDS table before they go to the global
// accumulators (sums are integers: bit-identical in any order):
//  * the point is loaded and voxelised ONCE for its eight corners (3 + 6 IEEE divisions per point instead of 48:
//    the relative coordinate of an axis has two values, floor and ceil) and the 16 bitmap / prefix words of the
//    eight corners are requested together;
//  * the table (slot, count, 8 x i64; open addressing on a multiplicative hash of the slot, LDS compare-and-swap,
//    kAccProbes probes, then the run goes to the global accumulators itself) takes the run sums of the eight tiles
//    with LDS atomics.  How much that saves depends on the patch it covers -- measured on the bench frame (134 k
//    touched voxels): an 8 x 4 patch with its corners touches 40 voxels (383 k table entries per frame, each
//    flushed with 9 atomics), 8 x 8: 61 (295 k), 16 x 16: 173 (208 k), 32 x 16: 310 (186 k);
//  * SHARED = false (r03 first version): one 64-entry table per wave, flushed per block, no barrier;
//    SHARED = true: the workgroup's 8 waves take the 8 blocks of a 16 x 16 patch and share ONE 512-entry table,
//    flushed by all threads behind a barrier: 46 % fewer flushed entries for two barriers per patch.
//  512 threads share one copy of the weights (22.5 KB) + 36.9 KB of tables: two workgroups per CU.
// Sharded encodes (owned-pair list) keep k_pointnet_scatter_t.
// ------------------------------------------------------------------------------------------
constexpr int kAccProbes = 6;     // probes before a run goes to the global accumulators instead
constexpr int kTbWaves = 8;       // waves per workgroup = blocks per 16 x 16 patch
constexpr int kAccCap = 64 * kTbWaves;
struct WgAcc {
  int key[kAccCap];
  int cnt[kAccCap];
  unsigned long long sum[kAccCap][8];
};

template <bool SHARED>
__global__ __launch_bounds__(64 * kTbWaves) void k_pointnet_scatter_tb(
    const float* __restrict__ pts, int n_points, int frame_w, bnv_grid_t g, const float* __restrict__ wpack,
    const uint32_t* __restrict__ bitmap, const uint32_t* __restrict__ word_prefix, int32_t* __restrict__ counts,
    long long* __restrict__ acc) {
  __shared__ __attribute__((aligned(16))) _Float16 wh[PT_TOTAL];
  __shared__ WgAcc T;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the table region this wave inserts into: all of it, or its own 64 entries
  constexpr int kSpan = SHARED ? kAccCap : 64;
  constexpr int kHashShift = SHARED ? 32 - 9 : 32 - 6;
  static_assert(kAccCap == 512, "hash width");
  const int t_base = SHARED ? 0 : wave * 64;
  T.key[threadIdx.x] = -1;
  T.cnt[threadIdx.x] = 0;
#pragma unroll
  for (int f = 0; f < 8; ++f) T.sum[threadIdx.x][f] = 0ull;
  stage_to_lds<64 * kTbWaves>(wpack, wh, PT_TOTAL * 2);
  __syncthreads();
  const int j = lane & 31, h = lane >> 5;
  const int nyz = g.n_xyz[1] * g.n_xyz[2];
  const bool sharded = g.shard_world > 1;
  // blocks: 8 x 4 pixel patches of a frame_w-wide image, or runs of 32 points; a workgroup's 8 waves take the 2 x 4
  // blocks of a 16 x 16 patch (SHARED) or 8 consecutive blocks
  const bool image = frame_w > 0 && n_points % frame_w == 0;
  const int frame_h = image ? n_points / frame_w : 1;
  const int bw = image ? (frame_w + 7) >> 3 : 0, bh = image ? (frame_h + 3) >> 2 : 0;
  const int n_blocks = image ? bw * bh : (n_points + 31) >> 5;
  const int uw = (bw + 1) >> 1;
  const int n_units = (SHARED && image) ? uw * ((bh + 3) >> 2) : (n_blocks + kTbWaves - 1) / kTbWaves;
  // entry e of the table goes to the global accumulators and is empty again
  auto flush_entry = [&](int e) {
    const int key = T.key[e];
    if (key >= 0) {
      unsigned long long* dst = (unsigned long long*)acc + (uint32_t)key * 8u;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        atomicAdd(dst + f, T.sum[e][f]);
        T.sum[e][f] = 0ull;
      }
      atomicAdd(&counts[key], T.cnt[e]);
      T.key[e] = -1;
      T.cnt[e] = 0;
    }
  };
  for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
    int i = -1;
    if (SHARED && image) {
      const int uy = u / uw, ux = u - uy * uw;
      const int by = uy * 4 + (wave >> 1), bx = ux * 2 + (wave & 1);
      const int x = bx * 8 + (j & 7), y = by * 4 + (j >> 3);
      if (bx < bw && x < frame_w && y < frame_h) i = y * frame_w + x;
    } else if (image) {
      const int b = u * kTbWaves + wave;
      const int by = b / bw, bx = b - by * bw;
      const int x = bx * 8 + (j & 7), y = by * 4 + (j >> 3);
      if (b < n_blocks && x < frame_w && y < frame_h) i = y * frame_w + x;
    } else if ((u * kTbWaves + wave) * 32 + j < n_points) {
      i = (u * kTbWaves + wave) * 32 + j;
    }
    bool valid = false;
    float px = 0.f, py = 0.f, pz = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
    if (i >= 0) {
      const float* p = pts + (size_t)i * 6;
      px = p[0], py = p[1], pz = p[2], n0 = p[3], n1 = p[4], n2 = p[5];
      valid = in_bounds(px, py, pz, g);
    }
    if (__ballot(valid) != 0ULL) {
    // voxelisation of the point, once for its eight corners
    int lo3[3] = {0, 0, 0}, hi3[3] = {0, 0, 0};
    _Float16 rl[3], rh[3];     // relative coordinate of an axis towards its floor / ceil voxel, as the network takes it
    {
      const float c3[3] = {px, py, pz};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float xn = valid ? voxel_coord(c3[a], g.bound_min[a], g.voxel_size) : 0.f;
        lo3[a] = (int)floorf(xn);
        hi3[a] = (int)ceilf(xn);
        rl[a] = (_Float16)relative_coord(xn, lo3[a], g.voxel_size);
        rh[a] = (_Float16)relative_coord(xn, hi3[a], g.voxel_size);
      }
    }
    // bitmap word + prefix of the eight corner voxels, all requested before the first is used
    uint32_t bw8[8], pf8[8], id8[8];
    bool own8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int gx = (k & 1) ? hi3[0] : lo3[0], gy = (k & 2) ? hi3[1] : lo3[1], gz = (k & 4) ? hi3[2] : lo3[2];
      id8[k] = (uint32_t)(gx * nyz + gy * g.n_xyz[2] + gz);
      bw8[k] = 0u;
      pf8[k] = 0u;
      // sharded volume: only the pairs whose voxel this rank owns (ownership goes by 8^3-voxel blocks, a patch of
      // the image lies in one or two of them: most corner tiles are all or nothing and the others are skipped)
      own8[k] = valid && (!sharded || voxel_owner(gx, gy, gz, g) == g.shard_rank);
      if (own8[k]) {
        bw8[k] = bitmap[id8[k] >> 5];
        pf8[k] = word_prefix[id8[k] >> 5];
      }
    }
    const _Float16 hn0 = (_Float16)n0, hn1 = (_Float16)n1, hn2 = (_Float16)n2;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int slot = own8[k] ? (int)(pf8[k] + __popc(bw8[k] & ((1u << (id8[k] & 31)) - 1u))) : -1;
      if (sharded && __ballot(slot >= 0) == 0ULL) continue;     // nothing of this corner tile is ours
      // operand slots of this lane half: features 8 (jj >> 2) + 4 h + (jj & 3); inputs 0..5, the rest 1.0
      half8 bop;
#pragma unroll
      for (int e = 0; e < 8; ++e) bop[e] = (_Float16)1.0f;
      if (valid) {
        if (h == 0) {
          bop[0] = (k & 1) ? rh[0] : rl[0];
          bop[1] = (k & 2) ? rh[1] : rl[1];
          bop[2] = (k & 4) ? rh[2] : rl[2];
          bop[3] = hn0;
        } else {
          bop[0] = hn1;
          bop[1] = hn2;
        }
      }
      f32x16 ha[2], hb[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
        ha[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&wh[PT_W1 + (mb * 64 + lane) * 8], bop, zero16(),
                                                        0, 0, 0);
      half8 s4[4];
      auto layer64 = [&](int woff, const f32x16 (&in)[2], f32x16 (&out)[2]) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          s4[nb * 2] = to_half8_relu(in[nb], 0);
          s4[nb * 2 + 1] = to_half8_relu(in[nb], 8);
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          out[mb] = zero16();
#pragma unroll
          for (int gk = 0; gk < 4; ++gk)
            out[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&wh[woff + ((mb * 4 + gk) * 64 + lane) * 8],
                                                            s4[gk], out[mb], 0, 0, 0);
        }
      };
      layer64(PT_W2, ha, hb);
      layer64(PT_W3, hb, ha);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        s4[nb * 2] = to_half8_relu(ha[nb], 0);
        s4[nb * 2 + 1] = to_half8_relu(ha[nb], 8);
      }
      f32x16 o = zero16();
#pragma unroll
      for (int gk = 0; gk < 4; ++gk)
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8*)&wh[PT_W4 + (gk * 64 + lane) * 8], s4[gk], o, 0, 0, 0);
      // the network returns fp16; lane (j, h) holds outputs 4h .. 4h+3 of pair j
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = (float)(_Float16)o[q];
      unsigned long long v[4];
      bool is_end;
      int len;
      tile_run_sums(o, slot, j, h, v, is_end, len);
      if (slot >= 0 && is_end) {
        // the run's sums into the table (both halves of a pair probe the same way and meet in the same entry).
        // Slots are ranks of ascending voxel ids -- a z-run of voxels is a run of slots: the multiplicative hash
        // keeps the runs of different rows from piling into one probe chain
        int p = (int)(((uint32_t)slot * 0x9E3779B1u) >> kHashShift), found = -1;
        for (int probe = 0; probe < kAccProbes; ++probe) {
          const int old = atomicCAS(&T.key[t_base + p], -1, slot);
          if (old == -1 || old == slot) {
            found = t_base + p;
            break;
          }
          p = (p + 1) & (kSpan - 1);
        }
        if (found >= 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) atomicAdd(&T.sum[found][4 * h + q], v[q]);
          if (h == 0) atomicAdd(&T.cnt[found], len);
        } else {   // no room within kAccProbes probes: straight to the global accumulators
          unsigned long long* dst = (unsigned long long*)acc + ((uint32_t)slot * 8u + 4u * (uint32_t)h);
#pragma unroll
          for (int q = 0; q < 4; ++q) atomicAdd(dst + q, v[q]);
          if (h == 0) atomicAdd(&counts[slot], len);
        }
      }
    }
    }
    if constexpr (SHARED) {
      __syncthreads();            // every wave's runs are in the table
      flush_entry(threadIdx.x);
      __syncthreads();            // the table is empty before the next patch inserts
    } else {
      flush_entry(t_base + lane);
    }
  }
}

// ------------------------------------------------------------------------------------------
// finalize: mean, min-points filter, ORDERED compaction of the emitted voxels (one pass: decoupled look-back over the
// workgroups), unflatten, cleanup of the per-frame scratch; the workgroup of the last tile completes the frame's
// counters and clears the control block.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kFinTile) void k_finalize(
    bnv_grid_t g, int emit_all, uint32_t* __restrict__ bitmap, int32_t* __restrict__ ids,
    int32_t* __restrict__ counts, long long* __restrict__ acc, uint64_t* __restrict__ tile_state, uint32_t epoch,
    EncCtl* __restrict__ ctl, const int32_t* __restrict__ valid_blocks, int n_mark_blocks,
    float* __restrict__ out_feats, int64_t* __restrict__ out_pcounts,
    int64_t* __restrict__ out_flat, int64_t* __restrict__ out_grid, int64_t out_capacity,
    bnv_encode_counters_t* __restrict__ counters) {
  __shared__ uint32_t wave_tot[kFinTile / 64];
  __shared__ uint32_t s_excl;
  const int64_t n = ctl->n_unique;
  if (n == 0) {
    // no voxel touched (no point passed the bounds mask): workgroup 0 reports the empty frame
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      counters->n_valid_points = 0;
      counters->n_unique = 0;
      counters->n_out = 0;
      counters->n_avg_pts = 0.f;
      counters->error = ctl->error;
      counters->reserved[0] = counters->reserved[1] = counters->reserved[2] = 0;
      ctl->error = 0;
      ctl->n_pairs = 0;
      ctl->n_orphans = 0;
      ctl->n_deferred = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) ctl->shard_boundary[threadIdx.x] = 0;
    return;
  }
  // ONE slot per thread: consecutive threads read consecutive 64-byte accumulator rows (with 8 slots per thread
  // every thread walked its own 512-byte stretch and the kernel was latency-bound at 34 us for 25 MB)
  // The number of slots is only known on the device: the launch is sized for a few thousand tiles at most and the
  // workgroups stride over the tiles in increasing order (tile t waits for tile t - 1 only: a workgroup that is
  // behind never waits for one that is ahead).  It used to cover max_unique -- 9,600 workgroups at 640x480, of which
  // ~500 had a tile and the others read n_unique and left.
  for (int64_t tile = blockIdx.x; tile * kFinTile < n; tile += gridDim.x) {
  if (tile != (int64_t)blockIdx.x) __syncthreads();      // wave_tot / s_excl of the previous tile are no longer read
  const int64_t sl = tile * kFinTile + threadIdx.x;
  ValidFlags flags{counts, ids, g, emit_all};
  const uint32_t fl = sl < n ? flags(sl) : 0u;
  int id = 0, c = 0;
  long long a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // sharded encode: 1 / world of the touched voxels are this rank's; the accumulators of the others were never
  // written (count 0), so their 64-byte rows are neither read nor cleaned
  bool dirty = false;
  if (sl < n) {
    id = ids[sl];
    c = counts[sl];
    dirty = g.shard_world <= 1 || c != 0;
    if (dirty) {
      typedef long long i64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const i64x2 v = *(const i64x2*)&acc[sl * 8 + 2 * q];
        a8[2 * q] = v[0];
        a8[2 * q + 1] = v[1];
      }
    }
  }
  uint32_t total;
  uint32_t run = block_exclusive_scan<kFinTile>(fl, wave_tot, &total);
  const bool last_tile = (tile + 1) * kFinTile >= n;
  if (threadIdx.x < 64) {
    const uint32_t excl = lookback_exclusive(tile_state, (int)tile, total, epoch);
    if (threadIdx.x == 0) s_excl = excl;
    if (last_tile) {
      // every other tile has published (so it has read ctl->n_unique): complete the counters, leave the
      // control block clean for the next frame
      int nv = 0;
      for (int b = threadIdx.x; b < n_mark_blocks; b += 64) nv += valid_blocks[b];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) nv += __shfl_xor(nv, d, 64);
      if (threadIdx.x == 0) {
        const int32_t n_out = (int32_t)(excl + total);
        counters->n_valid_points = nv;
        counters->n_unique = (int32_t)n;
        counters->n_out = n_out;
        // n_avg_pts = mean over ALL U voxels of the pair count (local_point_fusion.py:143); every valid point
        // contributes exactly 8 pairs, so the fp32 sum torch.mean forms is exactly 8 * n_valid
        counters->n_avg_pts = __fdiv_rn((float)(8 * nv), (float)n);
        counters->error = ctl->error ? ctl->error : ((int64_t)n_out > out_capacity ? 2 : 0);
        counters->reserved[0] = ctl->n_pairs;   // sharded encode: the (point, corner) pairs this rank encoded
        counters->reserved[1] = counters->reserved[2] = 0;
        ctl->n_unique = 0;
        ctl->error = 0;
        ctl->n_pairs = 0;
        ctl->n_orphans = 0;
        ctl->n_deferred = 0;
      }
      ctl->shard_boundary[threadIdx.x] = 0;
    }
  }
  __syncthreads();
  if (sl >= n) continue;
  run += s_excl;
  if (fl && (int64_t)run < out_capacity) {
    const bool keep = c >= g.min_pts_in_grid;  // emit_all: features zeroed below min_pts (:126)
    const float inv_scale = 1.0f / kFixedScale;
    f32x4 o[2];
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      // mean = sum / max(count, 1) (torch_scatter.scatter_mean); the fixed-point sum is exact
      const double sum = (double)a8[f] * (double)inv_scale;
      o[f >> 2][f & 3] = keep ? (float)(sum / (double)(c > 1 ? c : 1)) : 0.f;
    }
    *(f32x4*)&out_feats[(size_t)run * 8] = o[0];
    *(f32x4*)&out_feats[(size_t)run * 8 + 4] = o[1];
    out_pcounts[run] = c;
    out_flat[run] = id;
    const int nyz = g.n_xyz[1] * g.n_xyz[2];
    const int x = id / nyz, r = id - x * nyz, y = r / g.n_xyz[2], z = r - y * g.n_xyz[2];
    out_grid[(size_t)run * 3 + 0] = x;
    out_grid[(size_t)run * 3 + 1] = y;
    out_grid[(size_t)run * 3 + 2] = z;
  }
  // leave the scratch clean for the next frame
  if (dirty) {
    counts[sl] = 0;
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    const i64x2 z2 = {0, 0};
#pragma unroll
    for (int q = 0; q < 4; ++q) *(i64x2*)&acc[sl * 8 + 2 * q] = z2;
  }
  bitmap[id >> 5] = 0u;
  }
}

// ------------------------------------------------------------------------------------------
// k_voxelize_pairs (dense path + tests)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_voxelize_pairs(const float* __restrict__ pts, int n_points,
                                                       bnv_grid_t g, int32_t* __restrict__ grid_ids,
                                                       int64_t* __restrict__ flat_ids,
                                                       float* __restrict__ rel_xyz,
                                                       uint8_t* __restrict__ bound_mask) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_points) return;
  const float x = pts[(size_t)i * 6 + 0], y = pts[(size_t)i * 6 + 1], z = pts[(size_t)i * 6 + 2];
  if (bound_mask) bound_mask[i] = in_bounds(x, y, z, g) ? 1 : 0;
  const float xn = voxel_coord(x, g.bound_min[0], g.voxel_size);
  const float yn = voxel_coord(y, g.bound_min[1], g.voxel_size);
  const float zn = voxel_coord(z, g.bound_min[2], g.voxel_size);
  const int nyz = g.n_xyz[1] * g.n_xyz[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int cb = kCornerCeilBits[k];  // pair order follows the reference's corner order
    const int gx = (cb & 1) ? (int)ceilf(xn) : (int)floorf(xn);
    const int gy = (cb & 2) ? (int)ceilf(yn) : (int)floorf(yn);
    const int gz = (cb & 4) ? (int)ceilf(zn) : (int)floorf(zn);
    const size_t p = (size_t)k * n_points + i;
    if (grid_ids) {
      grid_ids[p * 3 + 0] = gx;
      grid_ids[p * 3 + 1] = gy;
      grid_ids[p * 3 + 2] = gz;
    }
    if (flat_ids) flat_ids[p] = (int64_t)(gx * nyz + gy * g.n_xyz[2] + gz);  // int32 arithmetic as the reference
    if (rel_xyz) {
      // relative_xyz = (xyz_normalized - grid_id) * voxel_size (local_point_fusion.py:163-164)
      rel_xyz[p * 3 + 0] = __fmul_rn(__fsub_rn(xn, (float)gx), g.voxel_size);
      rel_xyz[p * 3 + 1] = __fmul_rn(__fsub_rn(yn, (float)gy), g.voxel_size);
      rel_xyz[p * 3 + 2] = __fmul_rn(__fsub_rn(zn, (float)gz), g.voxel_size);
    }
  }
}

}  // namespace bnv

using namespace bnv;

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

#ifdef BNV_PHASE_PROF
int bnv_dev_enc_phase_read(unsigned long long* out128) {
  BNV_HIP_CHECK(hipDeviceSynchronize());
  BNV_HIP_CHECK(hipMemcpyFromSymbol(out128, HIP_SYMBOL(bnv::g_enc_phase), 128 * sizeof(unsigned long long)));
  unsigned long long z[128] = {};
  BNV_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(bnv::g_enc_phase), z, sizeof(z)));
  return BNV_OK;
}
#endif


int bnv_init(int device) {
  BNV_HIP_CHECK(hipSetDevice(device));
  int cus = 0;
  BNV_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
  g_num_cus = cus;
  BNV_HIP_CHECK(hipFuncSetAttribute((const void*)k_pointnet_scatter,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, PN_TOTAL * 4));
  BNV_HIP_CHECK(hipFuncSetAttribute((const void*)k_pointnet_scatter_x<3>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, PX_LDS_BYTES + kEncProfLds));
  BNV_HIP_CHECK(hipFuncSetAttribute((const void*)k_pointnet_scatter_x<1>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, PX_LDS_BYTES + kEncProfLds));
  extern int bnv_decode_init();
  return bnv_decode_init();
}

int bnv_num_compute_units(void) { return g_num_cus; }
int bnv_last_hip_error(void) { return g_last_hip_error; }

const char* bnv_status_string(int s) {
  switch (s) {
    case BNV_OK: return "ok";
    case BNV_ERR_INVALID_ARGUMENT: return "invalid argument";
    case BNV_ERR_WORKSPACE_TOO_SMALL: return "workspace too small";
    case BNV_ERR_HIP: return "HIP runtime error";
    case BNV_ERR_NOT_INITIALISED: return "bnv_init not called";
    case BNV_ERR_CAPACITY: return "capacity exceeded";
    default: return "unknown";
  }
}

size_t bnv_pointnet_pack_floats(void) { return PN_PACK_FLOATS; }

int bnv_set_mlp_mode(int mode) {
  if (mode < 0 || mode > 3) return BNV_ERR_INVALID_ARGUMENT;
  g_mlp_mode.store(mode, std::memory_order_relaxed);
  return BNV_OK;
}
int bnv_get_mlp_mode(void) { return g_mlp_mode.load(std::memory_order_relaxed); }

int bnv_profile_enable(int on) {
  for (int k = 0; k < PROF_KINDS; ++k) g_prof_used[k] = 0;
  g_prof_on = on != 0;
  return BNV_OK;
}

int bnv_profile_read(double* total_ms, int64_t* launches) {
  if (!total_ms || !launches) return BNV_ERR_INVALID_ARGUMENT;
  for (int k = 0; k < PROF_KINDS; ++k) {
    double ms = 0.0;
    for (size_t i = 0; i < g_prof_used[k]; ++i) {
      float t = 0.f;
      BNV_HIP_CHECK(hipEventSynchronize(g_prof_events[k][i].second));
      BNV_HIP_CHECK(hipEventElapsedTime(&t, g_prof_events[k][i].first, g_prof_events[k][i].second));
      ms += t;
    }
    total_ms[k] = ms;
    launches[k] = (int64_t)g_prof_used[k];
  }
  return BNV_OK;
}

size_t bnv_encode_workspace_bytes(int64_t max_points, const int32_t n_xyz[3]) {
  return encode_ws_layout(max_points, n_xyz, nullptr, nullptr);
}

int bnv_encode_workspace_reset(void* ws, size_t ws_bytes, bnv_stream_t stream) {
  if (!ws) return BNV_ERR_INVALID_ARGUMENT;
  BNV_HIP_CHECK(hipMemsetAsync(ws, 0, ws_bytes, (hipStream_t)stream));
  return BNV_OK;
}

// ---- encode in two halves.  begin = voxelise (bounds mask, 8 corner voxels, bitmap) + sorted-unique (rank);
// finish = PointNet + scatter-mean + min-points filter + ordered compaction.  Everything between the two lives in
// the workspace; bnv_encode_pointcloud is begin + finish.
static bool tcnn_blocks(const bnv_grid_t& g);

// rank (sorted-unique); under first-touch ownership then: owners for the frame's new blocks, the owned-pair list and
// the exchange bound (`pts`: the frame's input_pts rows)
static int encode_rank(const EncodeWs& ws, const bnv_grid_t& g, const float* pts, int n_points, hipStream_t stream) {
  const int nb_chunks = (int)((ws.n_chunks + kRankTile - 1) / kRankTile);
  hipLaunchKernelGGL(k_rank, dim3(nb_chunks), dim3(kScanThreads), 0, stream, ws.bytemap, ws.chunk_flag, ws.n_chunks,
                     ws.tile_state, next_epoch(), ws.bitmap, ws.word_prefix, ws.ids, ws.max_unique, ws.ctl, g,
                     ws.defer_list);
  BNV_LAUNCH_CHECK();
  if (g.shard_world > 1 && g.shard_state) {
    hipLaunchKernelGGL(k_shard_assign, dim3(1), dim3(256), 0, stream, g, (const EncCtl*)ws.ctl);
    BNV_LAUNCH_CHECK();
    // (a frame without a new block leaves it nothing to do: a small grid that strides, not a workgroup per 256 points)
    const int nb = (n_points + 255) / 256;
    const int cap = 64;
    hipLaunchKernelGGL(k_shard_own, dim3(nb < cap ? (nb > 0 ? nb : 1) : cap), dim3(256), 0, stream, pts, n_points, g,
                       tcnn_blocks(g) ? (int32_t*)nullptr : ws.pair_list, &ws.ctl->n_pairs, ws.orphan_list, ws.ids,
                       ws.defer_list, ws.ctl);
    BNV_LAUNCH_CHECK();
  }
  return BNV_OK;
}

// The tiny-cuda-nn block encoder finds a shard's pairs itself: `begin` then makes no pair list.  (begin and finish of
// one frame are called with the same grid, hence the same MLP mode: the mode follows the weight pack.)
static bool tcnn_blocks(const bnv_grid_t& g) {
  return mlp_mode_of(g.mlp_mode) == 2 && g_tcnn_block_encoder.load(std::memory_order_relaxed);
}

static bool grid_ok(const bnv_grid_t& g) {
  return (int64_t)g.n_xyz[0] * g.n_xyz[1] * g.n_xyz[2] < (1LL << 31) && g.n_xyz[0] > 0 && g.n_xyz[1] > 0 &&
         g.n_xyz[2] > 0 && g.shard_world >= 1 && g.shard_world <= 64 && g.shard_rank >= 0 &&
         g.shard_rank < g.shard_world && mlp_mode_field_ok(g.mlp_mode);
}

size_t bnv_encode_shard_counts_offset(void) { return offsetof(EncCtl, shard_boundary); }

size_t bnv_shard_state_bytes(const int32_t n_xyz[3], int32_t block_log2) {
  if (!n_xyz || block_log2 < 0 || block_log2 > 8) return 0;
  return shard_state_layout(n_xyz, block_log2, nullptr, nullptr);
}
int bnv_shard_state_configure(void* shard_state, int32_t rule, int32_t axis, bnv_stream_t stream) {
  if (!shard_state || (rule != BNV_SHARD_RULE_GREEDY && rule != BNV_SHARD_RULE_REGION) || axis < 0 || axis > 2)
    return BNV_ERR_INVALID_ARGUMENT;
  const int32_t words[2] = {rule, axis};
  BNV_HIP_CHECK(hipMemcpyAsync((char*)shard_state + offsetof(ShardHdr, rule), words, sizeof(words), hipMemcpyHostToDevice,
                               (hipStream_t)stream));
  BNV_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));   // (a set-up call: `words` lives on this stack frame)
  return BNV_OK;
}
size_t bnv_shard_state_loads_offset(void) { return offsetof(ShardHdr, load); }
size_t bnv_shard_state_table_offset(void) { return kShardHdrBytes; }

int bnv_encode_begin(const float* input_pts, int64_t n_points, const bnv_grid_t* grid_host, void* ws_ptr,
                     size_t ws_bytes, int64_t ws_max_points, bnv_stream_t stream_) {
  if (g_num_cus <= 0) return BNV_ERR_NOT_INITIALISED;
  if (!input_pts || !grid_host || !ws_ptr || n_points < 0 || n_points > (1 << 27) || ws_max_points < n_points)
    return BNV_ERR_INVALID_ARGUMENT;
  const bnv_grid_t g = *grid_host;
  if (!grid_ok(g)) return BNV_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  EncodeWs ws;
  // the layout is a function of the workspace's capacity, not of this frame's point count, so
  // the scratch the previous frame left clean stays where this frame expects it
  if (encode_ws_layout(ws_max_points, g.n_xyz, (char*)ws_ptr, &ws) > ws_bytes) return BNV_ERR_WORKSPACE_TOO_SMALL;
  if (n_points == 0) return BNV_OK;
  const int n = (int)n_points;
  hipLaunchKernelGGL(k_mark, dim3((n + 255) / 256), dim3(256), 0, stream, input_pts, n, g, ws.bytemap, ws.chunk_flag,
                     ws.valid_blocks,
                     (g.shard_world > 1 && !tcnn_blocks(g)) ? ws.pair_list : (int32_t*)nullptr, &ws.ctl->n_pairs,
                     g.shard_state ? ws.orphan_list : (int32_t*)nullptr, &ws.ctl->n_orphans);
  BNV_LAUNCH_CHECK();
  return encode_rank(ws, g, input_pts, n, stream);
}

int bnv_encode_begin_depth(const void* depth, int depth_dtype, int H, int W, const double* intr_host,
                           const double* T_wc_host, double max_depth, const bnv_grid_t* grid_host, void* ws_ptr,
                           size_t ws_bytes, int64_t ws_max_points, float* out_pts, bnv_stream_t stream_) {
  if (g_num_cus <= 0) return BNV_ERR_NOT_INITIALISED;
  if (!depth || !intr_host || !T_wc_host || !grid_host || !ws_ptr || !out_pts || H <= 0 || W <= 0 || depth_dtype < 0 ||
      depth_dtype > 2 || (int64_t)H * W > (1 << 27) || ws_max_points < (int64_t)H * W)
    return BNV_ERR_INVALID_ARGUMENT;
  const bnv_grid_t g = *grid_host;
  if (!grid_ok(g)) return BNV_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  EncodeWs ws;
  if (encode_ws_layout(ws_max_points, g.n_xyz, (char*)ws_ptr, &ws) > ws_bytes) return BNV_ERR_WORKSPACE_TOO_SMALL;
  FrontArgs a;
  front_args_fill(a, depth, depth_dtype, H, W, intr_host, T_wc_host, max_depth);
  const int64_t n = (int64_t)H * W;
  hipLaunchKernelGGL(k_front_mark, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, out_pts, g, ws.bytemap,
                     ws.chunk_flag, ws.valid_blocks,
                     (g.shard_world > 1 && !tcnn_blocks(g)) ? ws.pair_list : (int32_t*)nullptr, &ws.ctl->n_pairs,
                     g.shard_state ? ws.orphan_list : (int32_t*)nullptr, &ws.ctl->n_orphans);
  BNV_LAUNCH_CHECK();
  return encode_rank(ws, g, out_pts, (int)n, stream);
}

int bnv_encode_finish(const float* input_pts, int64_t n_points, const bnv_grid_t* grid_host,
                      const float* pointnet_pack, void* ws_ptr, size_t ws_bytes, int64_t ws_max_points,
                      float* out_feats, int64_t* out_pcounts, int64_t* out_flat_ids, int64_t* out_grid_ids,
                      int64_t out_capacity, int emit_all, bnv_encode_counters_t* counters, bnv_stream_t stream) {
  return bnv_encode_finish_image(input_pts, n_points, 0, grid_host, pointnet_pack, ws_ptr, ws_bytes, ws_max_points,
                                 out_feats, out_pcounts, out_flat_ids, out_grid_ids, out_capacity, emit_all, counters,
                                 stream);
}

int bnv_encode_finish_image(const float* input_pts, int64_t n_points, int image_width, const bnv_grid_t* grid_host,
                            const float* pointnet_pack, void* ws_ptr, size_t ws_bytes, int64_t ws_max_points,
                            float* out_feats, int64_t* out_pcounts, int64_t* out_flat_ids, int64_t* out_grid_ids,
                            int64_t out_capacity, int emit_all, bnv_encode_counters_t* counters, bnv_stream_t stream_) {
  return bnv_encode_finish_image_wg(input_pts, n_points, image_width, grid_host, pointnet_pack, ws_ptr, ws_bytes,
                                    ws_max_points, out_feats, out_pcounts, out_flat_ids, out_grid_ids, out_capacity,
                                    emit_all, counters, 0, stream_);
}

int bnv_encode_finish_image_wg(const float* input_pts, int64_t n_points, int image_width, const bnv_grid_t* grid_host,
                               const float* pointnet_pack, void* ws_ptr, size_t ws_bytes, int64_t ws_max_points,
                               float* out_feats, int64_t* out_pcounts, int64_t* out_flat_ids, int64_t* out_grid_ids,
                               int64_t out_capacity, int emit_all, bnv_encode_counters_t* counters,
                               int max_workgroups, bnv_stream_t stream_) {
  return bnv_encode_finish_image_parts(input_pts, n_points, image_width, grid_host, pointnet_pack, ws_ptr, ws_bytes,
                                       ws_max_points, out_feats, out_pcounts, out_flat_ids, out_grid_ids, out_capacity,
                                       emit_all, counters, max_workgroups, 3, stream_);
}

int bnv_encode_finish_image_parts(const float* input_pts, int64_t n_points, int image_width,
                                  const bnv_grid_t* grid_host, const float* pointnet_pack, void* ws_ptr,
                                  size_t ws_bytes, int64_t ws_max_points, float* out_feats, int64_t* out_pcounts,
                                  int64_t* out_flat_ids, int64_t* out_grid_ids, int64_t out_capacity, int emit_all,
                                  bnv_encode_counters_t* counters, int max_workgroups, int parts,
                                  bnv_stream_t stream_) {
  if (g_num_cus <= 0) return BNV_ERR_NOT_INITIALISED;
  if (image_width < 0 || max_workgroups < 0 || parts < 1 || parts > 3) return BNV_ERR_INVALID_ARGUMENT;
  if (!input_pts || !grid_host || !pointnet_pack || !ws_ptr || !counters || n_points < 0 ||
      n_points > (1 << 27) || ws_max_points < n_points)
    return BNV_ERR_INVALID_ARGUMENT;
  const bnv_grid_t g = *grid_host;
  if (!grid_ok(g)) return BNV_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  EncodeWs ws;
  if (encode_ws_layout(ws_max_points, g.n_xyz, (char*)ws_ptr, &ws) > ws_bytes) return BNV_ERR_WORKSPACE_TOO_SMALL;
  if (n_points == 0) {
    if (parts & 2) BNV_HIP_CHECK(hipMemsetAsync(counters, 0, sizeof(bnv_encode_counters_t), stream));
    return BNV_OK;
  }
  const int n = (int)n_points;
  // point encoder + scatter
  const int n_tiles = ((n + 31) / 32) * 8;
  const int reserve = g_reserve_cus.load(std::memory_order_relaxed);
  int grid_pn = g_num_cus - reserve > 0 ? g_num_cus - reserve : 1;
  if (max_workgroups > 0 && grid_pn > max_workgroups) grid_pn = max_workgroups;
  if (grid_pn > (n_tiles + 7) / 8) grid_pn = (n_tiles + 7) / 8;
  const int mlp = mlp_mode_of(g.mlp_mode);
  // sharded: owned pairs only -- from the list `begin` made, or (block encoder) by an ownership test in the kernel
  const int32_t* plist = (g.shard_world > 1 && !tcnn_blocks(g)) ? ws.pair_list : (const int32_t*)nullptr;
  if (parts & 1) {
    ProfScope prof(PROF_POINTNET, stream);
    if (tcnn_blocks(g)) {
      const int n_blocks = (n + 31) / 32 + 64;   // (an upper bound of the 8 x 4 patches as well, up to ragged edges)
      const int n_units = (n_blocks + kTbWaves - 1) / kTbWaves + 64;   // (16 x 16 patches: up to ragged edges)
      const int cus_tb = max_workgroups > 0 && max_workgroups < g_num_cus ? max_workgroups : g_num_cus;
      const int grid_tb = cus_tb * 2 < n_units ? cus_tb * 2 : n_units;
      auto kern = g_tcnn_shared_table.load(std::memory_order_relaxed) ? k_pointnet_scatter_tb<true> : k_pointnet_scatter_tb<false>;
      hipLaunchKernelGGL(kern, dim3(grid_tb), dim3(64 * kTbWaves), 0, stream, input_pts, n, image_width, g,
                         pointnet_pack, ws.bitmap, ws.word_prefix, ws.counts, ws.acc);
    } else if (mlp == 2)
      hipLaunchKernelGGL(k_pointnet_scatter_t, dim3(g_num_cus * 4 < (n_tiles + 3) / 4 ? g_num_cus * 4 : (n_tiles + 3) / 4),
                         dim3(256), 0, stream, input_pts, n, g, pointnet_pack, ws.bitmap, ws.word_prefix, ws.counts,
                         ws.acc, plist, &ws.ctl->n_pairs);
    else if (mlp == 1) {
      hipLaunchKernelGGL((k_pointnet_scatter_x<3>), dim3(grid_pn), dim3(512), PX_LDS_BYTES + kEncProfLds, stream,
                         input_pts, n, g, pointnet_pack, ws.bitmap, ws.word_prefix, ws.counts, ws.acc, &ws.ctl->error,
                         plist, &ws.ctl->n_pairs);
    } else if (mlp == 3) {
      hipLaunchKernelGGL((k_pointnet_scatter_x<1>), dim3(grid_pn), dim3(512), PX_LDS_BYTES + kEncProfLds, stream,
                         input_pts, n, g, pointnet_pack, ws.bitmap, ws.word_prefix, ws.counts, ws.acc, &ws.ctl->error,
                         plist, &ws.ctl->n_pairs);
    } else
      hipLaunchKernelGGL(k_pointnet_scatter, dim3(grid_pn), dim3(512), PN_TOTAL * 4, stream, input_pts, n, g,
                         pointnet_pack, ws.bitmap, ws.word_prefix, ws.counts, ws.acc, plist, &ws.ctl->n_pairs);
  }
  BNV_LAUNCH_CHECK();
  if (!(parts & 2)) return BNV_OK;
  // ordered compaction of the emitted voxels; the number of slots is only known on the device: a capped grid strides
  // over the tiles
  const int nb_max = (int)((ws.max_unique + kFinTile - 1) / kFinTile);
  // forward progress of the look-back needs every launched workgroup resident at once (a tile waits for its
  // predecessor's word): 1,024-thread workgroups, 2 per CU at most -- the option can lower the count, never raise it
  const int nb_res = 2 * g_num_cus;
  const int nb_opt = g_finalize_blocks.load(std::memory_order_relaxed);
  const int nb_cap = nb_opt > 0 && nb_opt < nb_res ? nb_opt : nb_res;
  const int nb_u = nb_max < nb_cap ? nb_max : nb_cap;
  hipLaunchKernelGGL(k_finalize, dim3(nb_u), dim3(kFinTile), 0, stream, g, emit_all, ws.bitmap, ws.ids,
                     ws.counts, ws.acc, ws.tile_state, next_epoch(), ws.ctl, ws.valid_blocks, (n + 255) / 256, out_feats,
                     out_pcounts, out_flat_ids, out_grid_ids, out_capacity, counters);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_encode_pointcloud(const float* input_pts, int64_t n_points, const bnv_grid_t* grid_host,
                          const float* pointnet_pack, void* ws_ptr, size_t ws_bytes, int64_t ws_max_points,
                          float* out_feats,
                          int64_t* out_pcounts, int64_t* out_flat_ids, int64_t* out_grid_ids,
                          int64_t out_capacity, int emit_all, bnv_encode_counters_t* counters,
                          bnv_stream_t stream) {
  if (!pointnet_pack || !counters) return BNV_ERR_INVALID_ARGUMENT;
  const int rc = bnv_encode_begin(input_pts, n_points, grid_host, ws_ptr, ws_bytes, ws_max_points, stream);
  if (rc != BNV_OK) return rc;
  return bnv_encode_finish(input_pts, n_points, grid_host, pointnet_pack, ws_ptr, ws_bytes, ws_max_points, out_feats,
                           out_pcounts, out_flat_ids, out_grid_ids, out_capacity, emit_all, counters, stream);
}

int bnv_voxelize_pairs(const float* input_pts, int64_t n_points, const bnv_grid_t* grid_host,
                       int32_t* grid_ids, int64_t* flat_ids, float* rel_xyz, uint8_t* bound_mask,
                       bnv_stream_t stream) {
  if (!input_pts || !grid_host || n_points < 0 || n_points > (1 << 27)) return BNV_ERR_INVALID_ARGUMENT;
  if (n_points == 0) return BNV_OK;
  hipLaunchKernelGGL(k_voxelize_pairs, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, input_pts, (int)n_points, *grid_host, grid_ids, flat_ids, rel_xyz,
                     bound_mask);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

}  // extern "C"
