// volume.hip -- SparseVolume (reference sparse_volume.py:484-695) on gfx950.
//
// Index: open-addressing hash table, 64-bit packed (x,y,z) key -> row, linear probing, CAS insert.
// Payload: dense row arrays (coords, features[8], weight, num_hits) in insertion order, so
// to_tensor() is a slice.  New rows are numbered by an ORDERED prefix sum over the batch (not by
// an atomic counter), so row order -- and every later result -- is reproducible run to run.
//
// All kernels are HBM/L2-latency bound gather/scatter: one thread per key, 64-B payload rows.
#include "bnv_common.hpp"

namespace bnv {

constexpr int kVolThreads = 256;
constexpr int kVolItems = 8;
constexpr int kVolTile = kVolThreads * kVolItems;

struct VolWs {
  int32_t* slot_of;        // [n] slot index of each key (or -1: key out of range)
  int32_t* is_new;         // [n] 1 if this thread's CAS created the slot
  uint32_t* block_sums;    // [n_blocks + 1]
  uint32_t* block_new;     // [ceil(n / 256) + 1] created keys per 256-key block (integrate)
  uint64_t* tile_state;    // [ws_bytes / 2048 + 2] look-back state of k_vol_integrate (epoch-tagged, never cleared; fixed place)
  int32_t* total_new;      // [1] workspace word 0: created keys (insert) / first new row (integrate)
  int32_t* error;          // [1] = vol.n_rows + 1: sticky error code (1 table full, 2 key range, 3 row capacity)
};

// The look-back words sit at a FIXED place -- right behind the control block, sized from the workspace's own byte
// count, never from a call's n -- so that no per-call array (slot_of / is_new of a larger call) can ever alias them:
// the words are epoch-tagged and never cleared, and a stale slot index in their place could pass for a published
// prefix of the current epoch (the workspace is shared by insert / integrate / integrate_batch calls of any size).
// n keys take >= 8 bytes of workspace, i.e. ws_bytes / 2048 + 2 words cover every tile count a call can have.
static size_t vol_tile_bytes(size_t ws_bytes) { return ((ws_bytes / 2048 + 2) * 8 + 255) / 256 * 256; }

static size_t vol_ws_layout(int64_t n, char* base, size_t ws_bytes, VolWs* ws) {
  if (n < 1) n = 1;
  const int64_t nb = (n + kVolTile - 1) / kVolTile;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off = (off + bytes + 255) / 256 * 256;
    return p;
  };
  char* d = take(256);   // control words first: their offsets do not depend on n
  char* t = take(vol_tile_bytes(ws_bytes));
  char* a = take(n * 4);
  char* b = take(n * 4);
  char* c = take((nb + 1) * 4);
  char* e = take(((n + 255) / 256 + 1) * 4);
  if (ws) {
    ws->tile_state = (uint64_t*)t;
    ws->slot_of = (int32_t*)a;
    ws->is_new = (int32_t*)b;
    ws->block_sums = (uint32_t*)c;
    ws->block_new = (uint32_t*)e;
    ws->total_new = (int32_t*)d;
    ws->error = nullptr;   // set by the entry points: the word behind the volume's row counter
  }
  return off;
}

__global__ __launch_bounds__(256) void k_vol_clear(uint64_t* __restrict__ slot_keys,
                                                   int32_t* __restrict__ slot_rows, int64_t n_slots,
                                                   int32_t* __restrict__ n_rows) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n_slots) {
    slot_keys[i] = kEmptyKey;
    slot_rows[i] = -1;
  }
  if (i == 0 && n_rows) {
    n_rows[0] = 0;
    n_rows[1] = 0;  // sticky error word
  }
}

// probe / CAS-insert one key per thread; records the slot and whether this thread created it
// every kernel of the upsert takes its element count either from the host (n) or, when n_dev is set,
// from device memory (then n is only the capacity the grid was sized for)
__device__ __forceinline__ int64_t dev_count(int64_t n, const int32_t* n_dev) {
  if (!n_dev) return n;
  const int64_t m = *n_dev;
  return m < n ? m : n;
}

__global__ __launch_bounds__(256) void k_vol_probe_insert(bnv_volume_t v, const int64_t* __restrict__ coords,
                                                          int64_t n, const int32_t* __restrict__ n_dev,
                                                          int32_t* __restrict__ slot_of,
                                                          int32_t* __restrict__ is_new,
                                                          int32_t* __restrict__ error,
                                                          uint32_t* __restrict__ block_new = nullptr) {
  n = dev_count(n, n_dev);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t key;
  int32_t slot = -1, created = 0;
  if (i >= n) {
    // (fall through to the block count with created = 0)
  } else if (pack_key(coords[i * 3 + 0], coords[i * 3 + 1], coords[i * 3 + 2], &key)) {
    const uint32_t mask = (uint32_t)(v.n_slots - 1);
    uint32_t s = mix64(key) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
      uint64_t k = v.slot_keys[s];
      if (k == kEmptyKey) {
        k = atomicCAS((unsigned long long*)&v.slot_keys[s], (unsigned long long)kEmptyKey,
                      (unsigned long long)key);
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
    if (slot < 0) *error = 1;  // table full
  } else {
    *error = 2;  // coordinate outside the 21-bit key range
  }
  if (i < n) {
    slot_of[i] = slot;
    is_new[i] = created;
  }
  if (block_new) {  // created keys of this 256-key block (first level of the ordered numbering)
    __shared__ uint32_t wave_new[4];
    const unsigned long long b = __ballot(created);
    if ((threadIdx.x & 63) == 0) wave_new[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) block_new[blockIdx.x] = wave_new[0] + wave_new[1] + wave_new[2] + wave_new[3];
  }
}

__global__ __launch_bounds__(kVolThreads) void k_vol_scan_partial(const int32_t* __restrict__ is_new, int64_t n,
                                                                  const int32_t* __restrict__ n_dev,
                                                                  uint32_t* __restrict__ block_sums) {
  __shared__ uint32_t wave_tot[kVolThreads / 64];
  n = dev_count(n, n_dev);
  const int64_t base = (int64_t)blockIdx.x * kVolTile + (int64_t)threadIdx.x * kVolItems;
  uint32_t s = 0;
#pragma unroll
  for (int e = 0; e < kVolItems; ++e)
    if (base + e < n) s += (uint32_t)is_new[base + e];
  uint32_t total;
  block_exclusive_scan<kVolThreads>(s, wave_tot, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_vol_scan_top(uint32_t* __restrict__ block_sums, int n_blocks,
                                                       int32_t* __restrict__ total_out) {
  __shared__ uint32_t wave_tot[16];
  uint32_t carry = 0;
  for (int base = 0; base < n_blocks; base += 1024) {
    const int i = base + threadIdx.x;
    const uint32_t val = (i < n_blocks) ? block_sums[i] : 0;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<1024>(val, wave_tot, &total);
    if (i < n_blocks) block_sums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) *total_out = (int32_t)carry;
}

// assigns rows to the created slots in batch order: row = n_rows + (number of created keys before i)
__global__ __launch_bounds__(kVolThreads) void k_vol_assign_rows(bnv_volume_t v, const int64_t* __restrict__ coords,
                                                                 int64_t n, const int32_t* __restrict__ n_dev,
                                                                 const int32_t* __restrict__ slot_of,
                                                                 const int32_t* __restrict__ is_new,
                                                                 const uint32_t* __restrict__ block_sums,
                                                                 int32_t* __restrict__ error) {
  __shared__ uint32_t wave_tot[kVolThreads / 64];
  n = dev_count(n, n_dev);
  const int64_t base = (int64_t)blockIdx.x * kVolTile + (int64_t)threadIdx.x * kVolItems;
  uint32_t fl[kVolItems];
  uint32_t s = 0;
#pragma unroll
  for (int e = 0; e < kVolItems; ++e) {
    fl[e] = (base + e < n) ? (uint32_t)is_new[base + e] : 0u;
    s += fl[e];
  }
  uint32_t total;
  uint32_t run = block_exclusive_scan<kVolThreads>(s, wave_tot, &total) + block_sums[blockIdx.x];
  const int32_t first = *v.n_rows;
#pragma unroll
  for (int e = 0; e < kVolItems; ++e) {
    if (fl[e]) {
      const int64_t i = base + e;
      const int64_t row = (int64_t)first + run;
      if (row < v.row_capacity) {
        v.slot_rows[slot_of[i]] = (int32_t)row;
        v.row_coords[row * 3 + 0] = coords[i * 3 + 0];
        v.row_coords[row * 3 + 1] = coords[i * 3 + 1];
        v.row_coords[row * 3 + 2] = coords[i * 3 + 2];
        brick_set(v, coords[i * 3 + 0], coords[i * 3 + 1], coords[i * 3 + 2], (int32_t)row);
        // a fresh row reads as zeros (SparseVolume.query of an absent key, sparse_volume.py:677-679)
        for (int f = 0; f < v.n_feats; ++f) v.features[row * v.n_feats + f] = 0.f;
        v.weights[row] = 0.f;
        v.num_hits[row] = 0.f;
        if (v.lattice_have) v.lattice_have[row] = 0u;
      } else {
        *error = 3;
      }
      ++run;
    }
  }
}

__global__ void k_vol_commit(int32_t* __restrict__ n_rows, const int32_t* __restrict__ total_new) {
  *n_rows += *total_new;
}

// ---- _integrate in ONE launch.  Every workgroup (256 keys) probes / CAS-inserts its keys, counts the keys it
// created, obtains the first row of its new keys by decoupled look-back over the workgroups (rows are numbered in
// batch order: the reference's insertion order; tile 0 seeds the chain with the volume's row count, the last tile
// commits the new count), creates the rows and applies the running average.  Keys are unique within a batch, so a
// slot / row is touched by exactly one thread; the look-back state is epoch-tagged and needs no clearing.
//
// FRAME (bnv_volume_integrate_frame, the per-frame pipeline): the same launch also does what used to be two more --
//  * the rows it touches are the ORIGINS of the frame's lattice decode: origin_stamp[row] = stamp_epoch
//    (k_lattice_stamp's job; the decode then starts at k_lattice_neighbors);
//  * sharded volume: the boundary voxels among them leave as 48-byte records {x, y, z, weight, features} with the
//    values just written, appended behind the block's header (k_shard_pack's job, without re-finding the rows).
struct IntegrateExtras {
  ShardRec* block;          // null: no records
  int64_t block_capacity;
  bnv_grid_t grid;          // ownership / boundary predicates (with block)
  int32_t* origin_stamp;    // null: no stamps
  int32_t stamp_epoch;
  int32_t* lattice_ctl;     // with origin_stamp: control words of the decode's later stages, cleared here
};

#ifndef BNV_INT_THREADS
#define BNV_INT_THREADS 1024
#endif
constexpr int kIntThreads = BNV_INT_THREADS;   // keys per workgroup of the upsert (one per thread): the look-back chain has n / kIntThreads links
template <bool FRAME>
__global__ __launch_bounds__(kIntThreads) void k_vol_integrate(bnv_volume_t v, const int64_t* __restrict__ coords,
                                                       const float* __restrict__ feats,
                                                       const int64_t* __restrict__ pcounts, int64_t n,
                                                       const int32_t* __restrict__ n_dev,
                                                       uint64_t* __restrict__ tile_state, uint32_t epoch,
                                                       int32_t* __restrict__ error, IntegrateExtras X) {
  n = dev_count(n, n_dev);
  if constexpr (FRAME) {
    // entries listed / tile counter / spare of the lattice decode that follows (k_lattice_stamp's other job)
    if (blockIdx.x == 0 && threadIdx.x == 0 && X.lattice_ctl) X.lattice_ctl[1] = X.lattice_ctl[2] = X.lattice_ctl[3] = 0;
  }
  if ((int64_t)blockIdx.x * kIntThreads >= n) return;
  const int64_t i = (int64_t)blockIdx.x * kIntThreads + threadIdx.x;
  uint64_t key;
  int32_t slot = -1, created = 0;
  if (i < n) {
    if (pack_key(coords[i * 3 + 0], coords[i * 3 + 1], coords[i * 3 + 2], &key)) {
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
      if (slot < 0) *error = 1;  // table full
    } else {
      *error = 2;  // coordinate outside the 21-bit key range
    }
  }
  // the values of an existing row are requested before the look-back, which then waits on other workgroups
  float w_old = 0.f;
  float fo[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int64_t row = -1;
  if (slot >= 0 && !created) {
    row = v.slot_rows[slot];
    if (row >= 0 && row < v.row_capacity) {
      w_old = v.weights[row];
      const f32x4 a = *(const f32x4*)&v.features[row * 8], b = *(const f32x4*)&v.features[row * 8 + 4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        fo[f] = a[f];
        fo[4 + f] = b[f];
      }
    } else {
      row = -1;
    }
  }
  __shared__ uint32_t wave_new[kIntThreads / 64];
  __shared__ uint32_t s_first;
  const unsigned long long bal = __ballot(created);
  const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (ln == 0) wave_new[wv] = (uint32_t)__popcll(bal);
  __syncthreads();
  if (threadIdx.x < 64) {
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < kIntThreads / 64; ++k) total += wave_new[k];
    const uint32_t seed = blockIdx.x == 0 ? (uint32_t)v.n_rows[0] : 0u;
    const uint32_t first = lookback_exclusive(tile_state, (int)blockIdx.x, total, epoch, seed);
    if (threadIdx.x == 0) {
      s_first = first;
      if ((int64_t)(blockIdx.x + 1) * kIntThreads >= n) v.n_rows[0] = (int32_t)(first + total);   // last tile: commit
    }
  }
  __syncthreads();
  if (i >= n || slot < 0) return;
  if (created) {
    uint32_t before = s_first + (uint32_t)__popcll(bal & ((1ull << ln) - 1ull));
    for (int k = 0; k < wv; ++k) before += wave_new[k];
    row = (int64_t)before;
    if (row >= v.row_capacity) {
      *error = 3;
      return;
    }
    v.slot_rows[slot] = (int32_t)row;
    v.row_coords[row * 3 + 0] = coords[i * 3 + 0];
    v.row_coords[row * 3 + 1] = coords[i * 3 + 1];
    v.row_coords[row * 3 + 2] = coords[i * 3 + 2];
    brick_set(v, coords[i * 3 + 0], coords[i * 3 + 1], coords[i * 3 + 2], (int32_t)row);
    v.num_hits[row] = 0.f;   // a fresh row reads as zeros (SparseVolume.query of an absent key, :677-679)
  } else if (row < 0) {
    return;
  }
  // fine_weights = clip(pcounts / 32, max=1)   (:660; int64 / 32 -> float32 true division)
  const float w = fminf(__fdiv_rn((float)pcounts[i], 32.0f), 1.0f);
  const float w_new = __fadd_rn(w_old, w);  // updated_weights = old + new (:649)
  f32x4 o[2];
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    const float fn = feats[i * 8 + f];
    // (old_feats * old_weights + new_feats * new_weights) / updated_weights (:650)
    o[f >> 2][f & 3] = __fdiv_rn(__fadd_rn(__fmul_rn(fo[f], w_old), __fmul_rn(fn, w)), w_new);
  }
  *(f32x4*)&v.features[row * 8] = o[0];
  *(f32x4*)&v.features[row * 8 + 4] = o[1];
  v.weights[row] = w_new;
  if (v.lattice_have) v.lattice_have[row] = 0u;   // the row's table entries are those of its former features
  if constexpr (FRAME) {
    if (X.origin_stamp) X.origin_stamp[row] = X.stamp_epoch;
    if (!X.block) return;
    const int x = (int)coords[i * 3 + 0], y = (int)coords[i * 3 + 1], z = (int)coords[i * 3 + 2];
    const bool send = shard_is_boundary(x, y, z, X.grid);
    // append, wave-aggregated: one atomic per wave on the block's counter (record order is free)
    const unsigned long long m = __ballot(send);
    if (!send) return;
    const int leader = (int)__ffsll((long long)m) - 1;
    int base = 0;
    if (ln == leader) base = atomicAdd(&X.block[0].x, (int)__popcll(m));
    base = __shfl(base, leader, 64);
    const int64_t at = (int64_t)base + (int64_t)__popcll(m & ((1ull << ln) - 1ull));
    if (at >= X.block_capacity) {
      X.block[0].z = 1;   // cannot happen when the capacity covers the frame's emitted voxels; checked by the receiver
      return;
    }
    ShardRec r;
    r.x = x;
    r.y = y;
    r.z = z;
    r.w = w_new;
#pragma unroll
    for (int f = 0; f < 8; ++f) r.f[f] = o[f >> 2][f & 3];
    X.block[1 + at] = r;
  }
}

// ---- batched _integrate: up to kVolBatchMax consecutive frames in 2 launches, results identical to integrating
// them one after the other (the frame-parallel multi-GPU mode replays a whole batch of frames on every rank, and a
// launch costs ~10 us of stream time whatever it does).  Items are (frame s, index i); the grid is frame-major,
// every frame padded to whole 256-item blocks.  Two per-SLOT side tables that belong to the volume:
//   slot_mask [n_slots]                bit s set <=> frame s of the batch holds the slot's key (zero between calls)
//   slot_items[n_slots * kVolBatchMax] the index i of that key in frame s (valid where the bit is set)
// B1 (k_vol_batch_probe) probes / CAS-inserts every key and fills the two tables -- it must be complete for ALL items
// before a key's creator can be told, hence a launch of its own.  B2 (k_vol_batch_integrate) does the rest.
constexpr int kVolBatchMax = BNV_VOLUME_BATCH_MAX;

struct VolBatch {
  const int64_t* coords[kVolBatchMax];
  const float* feats[kVolBatchMax];
  const int64_t* pcounts[kVolBatchMax];
  const int32_t* n_dev[kVolBatchMax];
  int64_t n[kVolBatchMax];
  int32_t block_start[kVolBatchMax + 1];   // first block of each frame; [n_frames] = total blocks
  int32_t n_frames;
};

__device__ __forceinline__ int batch_frame_of(const VolBatch& b, int block) {
  int s = 0;
#pragma unroll
  for (int k = 1; k < kVolBatchMax; ++k) s += (k < b.n_frames && block >= b.block_start[k]) ? 1 : 0;
  return s;
}

__global__ __launch_bounds__(256) void k_vol_batch_probe(bnv_volume_t v, VolBatch b, int32_t* __restrict__ slot_of,
                                                         uint32_t* __restrict__ slot_mask,
                                                         int32_t* __restrict__ slot_items,
                                                         int32_t* __restrict__ error) {
  const int s = batch_frame_of(b, blockIdx.x);
  const int64_t n = dev_count(b.n[s], b.n_dev[s]);
  const int64_t i = (int64_t)(blockIdx.x - b.block_start[s]) * 256 + threadIdx.x;
  const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int32_t slot = -1;
  if (i < n) {
    const int64_t* c = b.coords[s] + i * 3;
    uint64_t key;
    if (pack_key(c[0], c[1], c[2], &key)) {
      const uint32_t mask = (uint32_t)(v.n_slots - 1);
      uint32_t p = mix64(key) & mask;
      for (uint32_t probe = 0; probe <= mask; ++probe) {
        uint64_t k = v.slot_keys[p];
        if (k == kEmptyKey)
          k = atomicCAS((unsigned long long*)&v.slot_keys[p], (unsigned long long)kEmptyKey, (unsigned long long)key);
        if (k == kEmptyKey || k == key) {
          slot = (int32_t)p;
          break;
        }
        p = (p + 1) & mask;
      }
      if (slot < 0) {
        *error = 1;  // table full
      } else {
        atomicOr(&slot_mask[slot], 1u << s);
        slot_items[(int64_t)slot * kVolBatchMax + s] = (int32_t)i;
      }
    } else {
      *error = 2;  // coordinate outside the 21-bit key range
    }
  }
  slot_of[item] = slot;
}

// B2 (one launch for what were three: creator count, offsets + commit, apply): the creator of a key -- the FIRST frame
// of a key that was not in the volume before the batch, exactly the frame whose sequential upsert would have created
// the row -- gets its row by decoupled look-back over the workgroups (rows numbered in (frame, index) order; tile 0 seeds
// the chain with the volume's row count, the last tile commits the new count); then the thread of a key's first
// occurrence creates the row if needed and applies the running average of every frame that holds the key, in frame
// order, in registers: one read and one write of the row.  Safe in one launch: only the first-occurrence thread of a
// key ever writes its slot / row / mask, and every other occurrence leaves at once whatever it reads there.
__global__ __launch_bounds__(256) void k_vol_batch_integrate(bnv_volume_t v, VolBatch b,
                                                             const int32_t* __restrict__ slot_of,
                                                             uint32_t* __restrict__ slot_mask,
                                                             const int32_t* __restrict__ slot_items,
                                                             uint64_t* __restrict__ tile_state, uint32_t epoch,
                                                             int32_t* __restrict__ error) {
  const int s = batch_frame_of(b, blockIdx.x);
  const int64_t i = (int64_t)(blockIdx.x - b.block_start[s]) * 256 + threadIdx.x;
  const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int32_t slot = slot_of[item];
  uint32_t frames = 0;
  bool first = false;
  int created = 0;
  if (slot >= 0) {
    frames = slot_mask[slot];
    first = (__ffs(frames) - 1) == s;   // a later occurrence: the first one does the work (mask 0: already done)
    created = first && v.slot_rows[slot] < 0;
  }
  __shared__ uint32_t wave_new[4];
  __shared__ uint32_t s_first;
  const unsigned long long bal = __ballot(created);
  const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (ln == 0) wave_new[wv] = (uint32_t)__popcll(bal);
  __syncthreads();
  if (threadIdx.x < 64) {
    const uint32_t total = wave_new[0] + wave_new[1] + wave_new[2] + wave_new[3];
    const uint32_t seed = blockIdx.x == 0 ? (uint32_t)v.n_rows[0] : 0u;
    const uint32_t fr = lookback_exclusive(tile_state, (int)blockIdx.x, total, epoch, seed);
    if (threadIdx.x == 0) {
      s_first = fr;
      if (blockIdx.x == gridDim.x - 1) v.n_rows[0] = (int32_t)(fr + total);   // last tile: commit
    }
  }
  __syncthreads();
  if (!first) return;
  int64_t row;
  float w_acc = 0.f;
  float fo[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (created) {
    uint32_t before = s_first + (uint32_t)__popcll(bal & ((1ull << ln) - 1ull));
    for (int k = 0; k < wv; ++k) before += wave_new[k];
    row = (int64_t)before;
    if (row >= v.row_capacity) {
      *error = 3;
      return;
    }
    const int64_t* c = b.coords[s] + i * 3;
    v.slot_rows[slot] = (int32_t)row;
    v.row_coords[row * 3 + 0] = c[0];
    v.row_coords[row * 3 + 1] = c[1];
    v.row_coords[row * 3 + 2] = c[2];
    brick_set(v, c[0], c[1], c[2], (int32_t)row);
    v.num_hits[row] = 0.f;
  } else {
    row = v.slot_rows[slot];
    if (row < 0 || row >= v.row_capacity) return;
    w_acc = v.weights[row];
#pragma unroll
    for (int f = 0; f < 8; ++f) fo[f] = v.features[row * 8 + f];
  }
  slot_mask[slot] = 0;   // the table is all zeros again when the launch ends
  while (frames) {
    const int t = __ffs(frames) - 1;
    frames &= frames - 1;
    const int64_t it = (t == s) ? i : (int64_t)slot_items[(int64_t)slot * kVolBatchMax + t];
    const float w = fminf(__fdiv_rn((float)b.pcounts[t][it], 32.0f), 1.0f);
    const float w_new = __fadd_rn(w_acc, w);
    const float* fn = b.feats[t] + it * 8;
#pragma unroll
    for (int f = 0; f < 8; ++f) fo[f] = __fdiv_rn(__fadd_rn(__fmul_rn(fo[f], w_acc), __fmul_rn(fn[f], w)), w_new);
    w_acc = w_new;
  }
#pragma unroll
  for (int f = 0; f < 8; ++f) v.features[row * 8 + f] = fo[f];
  v.weights[row] = w_acc;
  if (v.lattice_have) v.lattice_have[row] = 0u;
}

__global__ __launch_bounds__(256) void k_vol_insert_apply(bnv_volume_t v, const float* __restrict__ feats,
                                                          const float* __restrict__ weights,
                                                          const float* __restrict__ hits, int64_t n,
                                                          const int32_t* __restrict__ n_dev,
                                                          const int32_t* __restrict__ slot_of) {
  n = dev_count(n, n_dev);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t slot = slot_of[i];
  if (slot < 0) return;
  const int64_t row = v.slot_rows[slot];
  if (row < 0 || row >= v.row_capacity) return;
  for (int f = 0; f < v.n_feats; ++f) v.features[row * v.n_feats + f] = feats[i * v.n_feats + f];
  v.weights[row] = weights[i];
  v.num_hits[row] = hits[i];
  if (v.lattice_have) v.lattice_have[row] = 0u;
}

__global__ __launch_bounds__(256) void k_vol_query(bnv_volume_t v, const int64_t* __restrict__ coords, int64_t n,
                                                   const float* __restrict__ features,
                                                   const float* __restrict__ weights,
                                                   const float* __restrict__ hits, int64_t row_limit,
                                                   float* __restrict__ out_f, float* __restrict__ out_w,
                                                   float* __restrict__ out_h, int32_t* __restrict__ out_rows) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t key;
  int row = -1;
  if (pack_key(coords[i * 3 + 0], coords[i * 3 + 1], coords[i * 3 + 2], &key))
    row = volume_find(v.slot_keys, v.slot_rows, (uint32_t)(v.n_slots - 1), key);
  if (row >= row_limit) row = -1;
  if (out_rows) out_rows[i] = row;
  if (out_f)
    for (int f = 0; f < v.n_feats; ++f) out_f[i * v.n_feats + f] = row >= 0 ? features[(int64_t)row * v.n_feats + f] : 0.f;
  if (out_w) out_w[i] = row >= 0 ? weights[row] : 0.f;
  if (out_h) out_h[i] = row >= 0 ? hits[row] : 0.f;
}

// weights[rows] += 1 with index_put semantics: each distinct row once (sparse_volume.py:622)
__global__ __launch_bounds__(256) void k_vol_count_optim(bnv_volume_t v, const int64_t* __restrict__ coords,
                                                         int64_t n, float* __restrict__ weights,
                                                         int64_t row_limit, int32_t* __restrict__ stamp,
                                                         int32_t epoch) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t key;
  if (!pack_key(coords[i * 3 + 0], coords[i * 3 + 1], coords[i * 3 + 2], &key)) return;
  const int row = volume_find(v.slot_keys, v.slot_rows, (uint32_t)(v.n_slots - 1), key);
  if (row < 0 || row >= row_limit) return;
  if (atomicExch(&stamp[row], epoch) != epoch) weights[row] = __fadd_rn(weights[row], 1.0f);
}

__global__ __launch_bounds__(256) void k_vol_rehash(bnv_volume_t v, int32_t* __restrict__ error) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= *v.n_rows) return;
  uint64_t key;
  if (!pack_key(v.row_coords[row * 3 + 0], v.row_coords[row * 3 + 1], v.row_coords[row * 3 + 2], &key)) return;
  const uint32_t mask = (uint32_t)(v.n_slots - 1);
  uint32_t s = mix64(key) & mask;
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    if (atomicCAS((unsigned long long*)&v.slot_keys[s], (unsigned long long)kEmptyKey,
                  (unsigned long long)key) == kEmptyKey) {
      v.slot_rows[s] = (int32_t)row;
      return;
    }
    s = (s + 1) & mask;
  }
  if (error) *error = 1;
}

static bool vol_ok(const bnv_volume_t* v) {
  return v && v->slot_keys && v->slot_rows && v->row_coords && v->features && v->weights && v->num_hits &&
         v->n_rows && v->n_slots > 0 && (v->n_slots & (v->n_slots - 1)) == 0 && v->n_slots <= (1LL << 31) &&
         v->row_capacity > 0 && v->n_feats == 8 &&
         (!v->brick || (v->brick_dims[0] > 0 && v->brick_dims[1] > 0 && v->brick_dims[2] > 0));
}

static int vol_upsert_rows(const bnv_volume_t& v, const int64_t* coords, int64_t n, const int32_t* n_dev,
                           const VolWs& ws, hipStream_t stream) {
  const int nbt = (int)((n + kVolTile - 1) / kVolTile);
  const unsigned nb256 = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_vol_probe_insert, dim3(nb256), dim3(256), 0, stream, v, coords, n, n_dev, ws.slot_of,
                     ws.is_new, ws.error);
  BNV_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_vol_scan_partial, dim3(nbt), dim3(kVolThreads), 0, stream, ws.is_new, n, n_dev,
                     ws.block_sums);
  BNV_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_vol_scan_top, dim3(1), dim3(1024), 0, stream, ws.block_sums, nbt, ws.total_new);
  BNV_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_vol_assign_rows, dim3(nbt), dim3(kVolThreads), 0, stream, v, coords, n, n_dev,
                     ws.slot_of, ws.is_new, ws.block_sums, ws.error);
  BNV_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_vol_commit, dim3(1), dim3(1), 0, stream, v.n_rows, ws.total_new);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

}  // namespace bnv

using namespace bnv;

extern "C" {

size_t bnv_volume_workspace_bytes(int64_t max_keys) {
  // fixed point of size = arrays(max_keys) + look-back words(size); + 512: the words' share grows by one 256-byte
  // step per 64 KB of workspace, so a caller that rounds the size up still passes the layout check
  size_t s = vol_ws_layout(max_keys, nullptr, 0, nullptr);
  for (int it = 0; it < 8; ++it) s = vol_ws_layout(max_keys, nullptr, s, nullptr);
  return s + 512;
}

int bnv_volume_clear(const bnv_volume_t* vol, bnv_stream_t stream) {
  if (!vol_ok(vol)) return BNV_ERR_INVALID_ARGUMENT;
  if (vol->brick) {
    const size_t nvox = (size_t)vol->brick_dims[0] * vol->brick_dims[1] * vol->brick_dims[2];
    BNV_HIP_CHECK(hipMemsetAsync(vol->brick, 0xff, nvox * 4, (hipStream_t)stream));   // every word -1
  }
  hipLaunchKernelGGL(k_vol_clear, dim3((unsigned)((vol->n_slots + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     vol->slot_keys, vol->slot_rows, vol->n_slots, vol->n_rows);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_volume_rehash(const bnv_volume_t* vol, bnv_stream_t stream) {
  if (!vol_ok(vol)) return BNV_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_vol_clear, dim3((unsigned)((vol->n_slots + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     vol->slot_keys, vol->slot_rows, vol->n_slots, (int32_t*)nullptr);
  BNV_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_vol_rehash, dim3((unsigned)((vol->row_capacity + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, *vol, vol->n_rows + 1);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_volume_integrate(const bnv_volume_t* vol, const int64_t* coords, const float* feats,
                         const int64_t* pcounts, int64_t n, const int32_t* n_dev, void* ws_ptr, size_t ws_bytes,
                         bnv_stream_t stream_) {
  if (!vol_ok(vol) || n < 0) return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  if (!coords || !feats || !pcounts || !ws_ptr) return BNV_ERR_INVALID_ARGUMENT;
  VolWs ws;
  if (vol_ws_layout(n, (char*)ws_ptr, ws_bytes, &ws) > ws_bytes) return BNV_ERR_WORKSPACE_TOO_SMALL;
  ws.error = vol->n_rows + 1;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(k_vol_integrate<false>, dim3((unsigned)((n + kIntThreads - 1) / kIntThreads)), dim3(kIntThreads), 0, stream, *vol, coords, feats, pcounts, n, n_dev,
                     ws.tile_state, next_epoch(), ws.error, IntegrateExtras{});
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_volume_integrate_frame(const bnv_volume_t* vol, const int64_t* coords, const float* feats,
                               const int64_t* pcounts, int64_t n, const int32_t* n_dev, void* ws_ptr, size_t ws_bytes,
                               const bnv_integrate_extras_t* extras, bnv_stream_t stream_) {
  if (!vol_ok(vol) || n < 0 || !extras) return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  if (!coords || !feats || !pcounts || !ws_ptr) return BNV_ERR_INVALID_ARGUMENT;
  IntegrateExtras X = {};
  if (extras->shard_block) {
    if (!extras->grid_host || extras->shard_block_capacity < 0) return BNV_ERR_INVALID_ARGUMENT;
    X.block = (ShardRec*)extras->shard_block;
    X.block_capacity = extras->shard_block_capacity;
    X.grid = *extras->grid_host;
  }
  if (extras->lattice_ws) {
    if (extras->stamp_epoch == 0) return BNV_ERR_INVALID_ARGUMENT;
    lattice_ws_frame_words(extras->lattice_ws, vol->row_capacity, &X.origin_stamp, &X.lattice_ctl);
    X.stamp_epoch = extras->stamp_epoch;
  }
  VolWs ws;
  if (vol_ws_layout(n, (char*)ws_ptr, ws_bytes, &ws) > ws_bytes) return BNV_ERR_WORKSPACE_TOO_SMALL;
  ws.error = vol->n_rows + 1;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(k_vol_integrate<true>, dim3((unsigned)((n + kIntThreads - 1) / kIntThreads)), dim3(kIntThreads), 0, stream, *vol, coords, feats, pcounts, n, n_dev,
                     ws.tile_state, next_epoch(), ws.error, X);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_volume_integrate_batch(const bnv_volume_t* vol, int n_frames, const int64_t* const* coords,
                               const float* const* feats, const int64_t* const* pcounts, const int64_t* n,
                               const int32_t* const* n_dev, uint32_t* slot_mask, int32_t* slot_items, void* ws_ptr,
                               size_t ws_bytes, bnv_stream_t stream_) {
  if (!vol_ok(vol) || n_frames < 0 || n_frames > kVolBatchMax) return BNV_ERR_INVALID_ARGUMENT;
  if (n_frames == 0) return BNV_OK;
  if (!coords || !feats || !pcounts || !n || !slot_mask || !slot_items || !ws_ptr) return BNV_ERR_INVALID_ARGUMENT;
  VolBatch b = {};
  int64_t blocks = 0;
  for (int s = 0; s < n_frames; ++s) {
    if (n[s] < 0 || (n[s] > 0 && (!coords[s] || !feats[s] || !pcounts[s]))) return BNV_ERR_INVALID_ARGUMENT;
    b.coords[s] = coords[s];
    b.feats[s] = feats[s];
    b.pcounts[s] = pcounts[s];
    b.n_dev[s] = n_dev ? n_dev[s] : nullptr;
    b.n[s] = n[s];
    b.block_start[s] = (int32_t)blocks;
    blocks += (n[s] + 255) / 256;
  }
  for (int s = n_frames; s <= kVolBatchMax; ++s) b.block_start[s] = (int32_t)blocks;
  b.n_frames = n_frames;
  if (blocks == 0) return BNV_OK;
  if (blocks > 0x7fffffff / 256) return BNV_ERR_INVALID_ARGUMENT;
  VolWs ws;
  if (vol_ws_layout(blocks * 256, (char*)ws_ptr, ws_bytes, &ws) > ws_bytes) return BNV_ERR_WORKSPACE_TOO_SMALL;
  ws.error = vol->n_rows + 1;
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(k_vol_batch_probe, dim3((unsigned)blocks), dim3(256), 0, stream, *vol, b, ws.slot_of, slot_mask,
                     slot_items, ws.error);
  BNV_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_vol_batch_integrate, dim3((unsigned)blocks), dim3(256), 0, stream, *vol, b, ws.slot_of,
                     slot_mask, slot_items, ws.tile_state, next_epoch(), ws.error);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_volume_insert(const bnv_volume_t* vol, const int64_t* coords, const float* feats,
                      const float* weights, const float* num_hits, int64_t n, void* ws_ptr,
                      size_t ws_bytes, bnv_stream_t stream_) {
  if (!vol_ok(vol) || n < 0) return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  if (!coords || !feats || !weights || !num_hits || !ws_ptr) return BNV_ERR_INVALID_ARGUMENT;
  VolWs ws;
  if (vol_ws_layout(n, (char*)ws_ptr, ws_bytes, &ws) > ws_bytes) return BNV_ERR_WORKSPACE_TOO_SMALL;
  ws.error = vol->n_rows + 1;
  hipStream_t stream = (hipStream_t)stream_;
  const int rc = vol_upsert_rows(*vol, coords, n, nullptr, ws, stream);
  if (rc != BNV_OK) return rc;
  hipLaunchKernelGGL(k_vol_insert_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, *vol, feats,
                     weights, num_hits, n, (const int32_t*)nullptr, ws.slot_of);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_volume_query(const bnv_volume_t* vol, const int64_t* coords, int64_t n, const float* features,
                     const float* weights, const float* num_hits, int64_t row_limit, float* out_feats,
                     float* out_weights, float* out_hits, int32_t* out_rows, bnv_stream_t stream) {
  if (!vol_ok(vol) || n < 0) return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  if (!coords || !features || !weights || !num_hits) return BNV_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_vol_query, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *vol,
                     coords, n, features, weights, num_hits, row_limit, out_feats, out_weights, out_hits, out_rows);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

int bnv_volume_count_optim(const bnv_volume_t* vol, const int64_t* coords, int64_t n, float* weights,
                           int64_t row_limit, int32_t* stamp, int32_t epoch, bnv_stream_t stream) {
  if (!vol_ok(vol) || n < 0 || !stamp || !weights) return BNV_ERR_INVALID_ARGUMENT;
  if (n == 0) return BNV_OK;
  hipLaunchKernelGGL(k_vol_count_optim, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     *vol, coords, n, weights, row_limit, stamp, epoch);
  BNV_LAUNCH_CHECK();
  return BNV_OK;
}

}  // extern "C"
