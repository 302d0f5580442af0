// shard.hip -- the exchange step of the spatially sharded volume (BASELINE.json north_star:
// "the active-voxel set shards by spatial hash across the GPUs ... with an RCCL all-gather over xGMI of
// boundary-voxel features before decode").  New design: the reference is single-GPU.
//
// A voxel is owned by hash(block coordinate) % world (bnv_common.hpp: voxel_owner).  Fusion needs no exchange: a
// (point, corner) pair belongs to exactly one voxel, every rank voxelises the whole frame and encodes / upserts only
// the voxels it owns.  Decode reads the 3x3x3 neighbourhood of a voxel, so a rank also needs the rows of foreign
// voxels that touch its own: GHOST rows.  A row changes only when its voxel is emitted by a frame's encode, so per
// frame every rank sends the rows it has just updated that are BOUNDARY voxels (some voxel of their 3x3x3
// neighbourhood belongs to another rank -- a function of the coordinates alone), one all-gather moves them, and
// every rank installs the records that touch voxels it owns.  After that the local volume (own rows + ghost rows)
// decodes exactly like the single-GPU volume.
//
//   k_shard_pack     this frame's emitted voxels that are boundary voxels -> 48-byte records (key, weight, feature)
//                    with their LIVE values (after the upsert), appended behind a header record that carries the count
//   k_shard_install  the other ranks' records that are adjacent to this rank: upsert as ghost rows (overwrite)
//
// HBM-bound and small: ~58 % of a frame's emitted voxels at 8^3-voxel blocks, 48 B each.
#include "bnv_common.hpp"

namespace bnv {

__global__ __launch_bounds__(256) void k_shard_pack(bnv_volume_t v, bnv_grid_t g, const int64_t* __restrict__ coords,
                                                    int64_t n, const int32_t* __restrict__ n_dev,
                                                    ShardRec* __restrict__ block, int64_t capacity) {
  if (n_dev) n = (int64_t)*n_dev < n ? (int64_t)*n_dev : n;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool send = false;
  int x = 0, y = 0, z = 0, row = -1;
  if (i < n) {
    x = (int)coords[i * 3 + 0];
    y = (int)coords[i * 3 + 1];
    z = (int)coords[i * 3 + 2];
    if (shard_is_boundary(x, y, z, g)) {
      row = volume_row(v, x, y, z);
      send = row >= 0;
    }
  }
  // append, wave-aggregated: one atomic per wave on the block's counter
  const unsigned long long m = __ballot(send);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  const int leader = (int)__ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(&block[0].x, (int)__popcll(m));
  base = __shfl(base, leader, 64);
  if (!send) return;
  const int64_t slot = (int64_t)base + (int64_t)__popcll(m & ((1ull << lane) - 1ull));
  if (slot >= capacity) {
    block[0].z = 1;   // cannot happen when capacity >= the bound bnv_encode_begin reports; checked by the receiver
    return;
  }
  ShardRec r;
  r.x = x;
  r.y = y;
  r.z = z;
  r.w = v.weights[row];
#pragma unroll
  for (int f = 0; f < 8; ++f) r.f[f] = v.features[(size_t)row * 8 + f];
  block[1 + slot] = r;
}

__global__ void k_shard_pack_header(ShardRec* __restrict__ block, int rank) {
  block[0].x = 0;
  block[0].y = rank;
  block[0].z = 0;
}

// Ghost-row look-up of k_shard_install, one thread per record: the row of the record's voxel in
// this rank's volume, created (hash slot claimed by CAS, row number from a wave-aggregated atomic on the row counter:
// ghost rows need no particular order) when it does not exist yet.  Keys are unique over all records of a frame (every
// voxel has one owner, and an owner sends a voxel once), so a slot / row is touched by one thread only.  Every lane of
// the wave must call it (ballot); `want` = this lane holds a record for this rank.  -> row, or -1 (nothing to do, or an
// error that has been written to *error); *created tells a fresh row (coordinates, brick entry and num_hits are set).
__device__ __forceinline__ int64_t ghost_row(const bnv_volume_t& v, const ShardRec& r, bool want, bool* created_out,
                                             int32_t* __restrict__ error) {
  uint64_t key;
  int32_t slot = -1, created = 0;
  if (want) {
    if (pack_key(r.x, r.y, r.z, &key)) {
      const uint32_t mask = (uint32_t)(v.n_slots - 1);
      uint32_t s = mix64(key) & mask;
      for (uint32_t probe = 0; probe <= mask; ++probe) {
        uint64_t k = v.slot_keys[s];
        if (k == kEmptyKey) {
          k = atomicCAS((unsigned long long*)&v.slot_keys[s], (unsigned long long)kEmptyKey, (unsigned long long)key);
          if (k == kEmptyKey) {
            slot = (int32_t)s;
            created = 1;
            break;
          }
        }
        if (k == key) {
          slot = (int32_t)s;
          break;
        }
        s = (s + 1) & mask;
      }
      if (slot < 0) *error = 1;
    } else {
      *error = 2;
    }
  }
  const unsigned long long m = __ballot(created);
  int64_t row = -1;
  if (m) {
    const int lane = threadIdx.x & 63;
    const int leader = (int)__ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(&v.n_rows[0], (int)__popcll(m));
    base = __shfl(base, leader, 64);
    if (created) row = (int64_t)base + (int64_t)__popcll(m & ((1ull << lane) - 1ull));
  }
  *created_out = created != 0;
  if (slot < 0) return -1;
  if (created) {
    if (row >= v.row_capacity) {
      *error = 3;
      return -1;
    }
    v.slot_rows[slot] = (int32_t)row;
    v.row_coords[row * 3 + 0] = r.x;
    v.row_coords[row * 3 + 1] = r.y;
    v.row_coords[row * 3 + 2] = r.z;
    brick_set(v, r.x, r.y, r.z, (int32_t)row);
    v.num_hits[row] = 0.f;
    return row;
  }
  row = v.slot_rows[slot];
  return (row < 0 || row >= v.row_capacity) ? -1 : row;
}

// one thread per (sender, record slot)
__global__ __launch_bounds__(256) void k_shard_install(bnv_volume_t v, bnv_grid_t g,
                                                       const ShardRec* __restrict__ blocks, int world,
                                                       int64_t capacity, int32_t* __restrict__ error,
                                                       ShardRec* __restrict__ own_block) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // the frame pipeline (pipeline.hip) appends a frame's records from inside the upsert kernel; the block that was
  // just exchanged starts its next frame empty (the collective that read it is complete: this kernel is behind it)
  if (t == 0 && own_block) own_block[0].x = 0;
  const int sender = (int)(t / capacity);
  const int64_t i = t - (int64_t)sender * capacity;
  bool want = false;
  ShardRec r = {};
  if (sender < world && sender != g.shard_rank) {
    const ShardRec* blk = blocks + (size_t)sender * (size_t)(capacity + 1);
    int cnt = blk[0].x;
    // the sender's block overflowed, or it holds more records than were exchanged: records are missing
    if (blk[0].z || cnt > capacity) *error = 4;
    if (cnt > capacity) cnt = (int)capacity;
    if (i < cnt) {
      r = blk[1 + i];
      want = shard_adjacent_to(r.x, r.y, r.z, g, g.shard_rank);
    }
  }
  bool created;
  const int64_t row = ghost_row(v, r, want, &created, error);
  if (row < 0) return;
#pragma unroll
  for (int f = 0; f < 8; ++f) v.features[row * 8 + f] = r.f[f];
  v.weights[row] = r.w;
  if (v.lattice_have) v.lattice_have[row] = 0u;   // a ghost row with new values: its table entries are stale
}

}  // namespace bnv

using namespace bnv;

static bool shard_vol_ok(const bnv_volume_t* v) {
  return v && v->slot_keys && v->slot_rows && v->row_coords && v->features && v->weights && v->num_hits && v->n_rows &&
         v->n_slots > 0 && (v->n_slots & (v->n_slots - 1)) == 0 && v->row_capacity > 0 && v->n_feats == 8;
}

extern "C" {

int bnv_shard_pack(const bnv_volume_t* vol, const bnv_grid_t* grid, const int64_t* coords, int64_t n,
                   const int32_t* n_dev, void* block, int64_t capacity, bnv_stream_t stream_) {
  if (!shard_vol_ok(vol) || !grid || !block || n < 0 || capacity < 0 || grid->shard_world < 1) return BNV_ERR_INVALID_ARGUMENT;
  if (n > 0 && !coords) return BNV_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(k_shard_pack_header, dim3(1), dim3(1), 0, stream, (ShardRec*)block, grid->shard_rank);
  BNV_LAUNCH_CHECK();
  if (n == 0) return BNV_OK;
  hipLaunchKernelGGL(k_shard_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, *vol, *grid, coords, n,
                     n_dev, (ShardRec*)block, capacity);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_shard_install_reset(const bnv_volume_t* vol, const bnv_grid_t* grid, const void* blocks, int world,
                            int64_t capacity, void* own_send_block, bnv_stream_t stream_) {
  if (!shard_vol_ok(vol) || !grid || !blocks || world < 1 || world != grid->shard_world || capacity < 0)
    return BNV_ERR_INVALID_ARGUMENT;
  if (capacity == 0 || world == 1) return BNV_OK;
  const int64_t total = (int64_t)world * capacity;
  hipLaunchKernelGGL(k_shard_install, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, *vol,
                     *grid, (const ShardRec*)blocks, world, capacity, vol->n_rows + 1, (ShardRec*)own_send_block);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_shard_install(const bnv_volume_t* vol, const bnv_grid_t* grid, const void* blocks, int world,
                      int64_t capacity, bnv_stream_t stream) {
  return bnv_shard_install_reset(vol, grid, blocks, world, capacity, nullptr, stream);
}

}  // extern "C"
