// Shared device helpers for the gfx950 kernels (wave64, CDNA4).  Compiled with
// -ffp-contract=off: every float op below rounds exactly once, like the reference's ATen CPU ops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/bnv_fusion.h"

namespace bnv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;
extern int g_num_cus;
extern int g_last_hip_error;
// process-wide defaults / A-B switches: relaxed atomic words, read once per launch (bnv_set_mlp_mode, bnv_set_option)
extern std::atomic<int> g_mlp_mode;
extern std::atomic<int> g_reserve_cus;
extern std::atomic<int> g_tcnn_block_encoder;
extern std::atomic<int> g_tcnn_shared_table;
extern std::atomic<int> g_finalize_blocks;
// the arithmetic mode of a call: the grid's own (bnv_grid_t.mlp_mode = 1 + m) or the process default
static inline int mlp_mode_of(int32_t grid_mode) {
  return grid_mode >= 1 && grid_mode <= 4 ? grid_mode - 1 : g_mlp_mode.load(std::memory_order_relaxed);
}
static inline bool mlp_mode_field_ok(int32_t grid_mode) { return grid_mode >= 0 && grid_mode <= 4; }

// optional HIP-event timing of the dominant kernels (bnv_profile_enable / bnv_profile_read)
enum ProfKind { PROF_POINTNET = 0, PROF_DECODE_LATTICE = 1, PROF_DECODE_PTS = 2, PROF_DECODE_DENSE = 3, PROF_KINDS = 4 };
extern bool g_prof_on;
void prof_mark(int kind, bool begin, hipStream_t stream);
// ReLU as ONE instruction: v_max_i32 on the bit pattern (a float with the sign bit set is a negative
// int; -0.0 -> +0.0).  fmaxf() and fmed3 both cost an extra canonicalising v_max under IEEE mode, and an
// inline-asm v_max_f32 would hide the MFMA-result -> VALU hazard from the compiler.
__device__ __forceinline__ float relu_bits(float x) {
  const int b = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// hi/lo f16 split of fp32 values for the split-operand MFMA modes: hi = rn16(x), lo = rn16(x - hi).  Written with
// v_cvt_pk_f16_f32 + v_fma_mixlo/mixhi_f16 (x - hi is formed in fp32 straight from the packed f16 hi and rounded once
// into the destination half): 1.5 VALU ops per value where the plain C++ form compiles to ~2.9 (cvt, cvt back, sub,
// cvt, pack).  Bit-identical to that form, subnormals and overflow included (tools/probe_split_mix.hip).  Plain asm
// (not volatile): the compiler still schedules and dead-code-eliminates it; its inputs come out of a v_max_i32 or a
// plain move, so no MFMA-result hazard is hidden from it.
typedef _Float16 bnv_half8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split_pair_f16(float a, float b, unsigned& hi, unsigned& lo) {
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
  asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a));
  asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(b));
}
__device__ __forceinline__ void split8_f16(const float (&x)[8], bnv_half8& hi, bnv_half8& lo) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t h, l;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned hh, ll;
    split_pair_f16(x[2 * p], x[2 * p + 1], hh, ll);
    h[p] = hh;
    l[p] = ll;
  }
  hi = __builtin_bit_cast(bnv_half8, h);
  lo = __builtin_bit_cast(bnv_half8, l);
}

// Cooperative global -> LDS copy of `bytes` (a multiple of 16; src and dst 16-byte aligned) by the NT threads of a
// workgroup through the LDS-DMA path (global_load_lds_dwordx4: no VGPR round trip, every piece in flight at once).
// The persistent MLP kernels stage 20-150 KB of weights per workgroup before their first tile; written as a
// load / ds_write loop the compiler kept ONE 16-byte load in flight per thread (load, s_waitcnt vmcnt(0), ds_write,
// branch): 19 dependent memory round trips = ~15 us of every launch of the point encoder.  A wave's piece is 1 KB:
// wave-uniform LDS base + lane * 16.  Ends with vmcnt(0); the caller's __syncthreads() publishes the data.
template <int NT>
__device__ __forceinline__ void stage_to_lds(const void* __restrict__ src, void* __restrict__ lds_dst, int bytes) {
  typedef __attribute__((address_space(1))) const void gptr_t;
  typedef __attribute__((address_space(3))) void lptr_t;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* s = (const char*)src;
  char* d = (char*)lds_dst;
  for (int base = wave * 1024; base < bytes; base += NT * 16) {   // (wave-uniform loop)
    if (base + lane * 16 < bytes)
      __builtin_amdgcn_global_load_lds((gptr_t*)(s + base + lane * 16), (lptr_t*)(d + base), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

struct ProfScope {
  int kind;
  hipStream_t stream;
  ProfScope(int k, hipStream_t s) : kind(k), stream(s) { if (g_prof_on) prof_mark(kind, true, stream); }
  ~ProfScope() { if (g_prof_on) prof_mark(kind, false, stream); }
};

#define BNV_HIP_CHECK(expr)                      \
  do {                                           \
    hipError_t _e = (expr);                      \
    if (_e != hipSuccess) {                      \
      bnv::g_last_hip_error = (int)_e;           \
      return BNV_ERR_HIP;                        \
    }                                            \
  } while (0)

#define BNV_LAUNCH_CHECK() BNV_HIP_CHECK(hipGetLastError())

// Corner order of get_neighbors (modules.py:590-655): bit0 = x uses ceil, bit1 = y, bit2 = z.
__device__ __constant__ const unsigned char kCornerCeilBits[8] = {0, 1, 2, 4, 3, 5, 6, 7};

// Normalised voxel coordinate of one axis: (x - bound_min) / voxel, two IEEE fp32 roundings
// (local_point_fusion.py:159-160).
__device__ __forceinline__ float voxel_coord(float x, float bmin, float voxel) {
  return __fdiv_rn(__fsub_rn(x, bmin), voxel);
}

__device__ __forceinline__ bool in_bounds(float x, float y, float z, const bnv_grid_t& g) {
  // strict, one-voxel margin (local_point_fusion.py:94-100); NaN fails every comparison.
  return (x < g.bound_hi[0]) && (y < g.bound_hi[1]) && (z < g.bound_hi[2]) &&
         (x > g.bound_lo[0]) && (y > g.bound_lo[1]) && (z > g.bound_lo[2]);
}

// rel = ((xn - gid) * voxel) / voxel  (local_point_fusion.py:163-164 then :59)
__device__ __forceinline__ float relative_coord(float xn, int gid, float voxel) {
  float t = __fsub_rn(xn, (float)gid);
  return __fdiv_rn(__fmul_rn(t, voxel), voxel);
}

__device__ __forceinline__ uint32_t mix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return (uint32_t)k;
}

// ---- first-touch ownership (bnv_grid_t::shard_state; encode.hip: k_shard_assign) ------------------------------------
// [header 1024 B: uint64 load[64] | int32 any_new | ...][owner table: 1 byte per block][block weights u32][new list u32]
constexpr size_t kShardHdrBytes = 1024;
constexpr uint8_t kOwnRank = 0x3f, kOwnAssigned = 0x40, kOwnTouched = 0x80;
constexpr uint32_t kNewListCap = 4096;   // new blocks of a frame k_rank lists for k_shard_assign (a power of two: sorted in place)
struct ShardHdr {
  unsigned long long load[64];   // touched voxels (at first touch) of the blocks each rank owns
  int32_t any_new;               // k_rank: the blocks this frame touches for the first time (their number)
  int32_t rule;                  // BNV_SHARD_RULE_GREEDY (0) | BNV_SHARD_RULE_REGION (1): bnv_shard_state_configure
  int32_t axis;                  // region rule: the axis the first frame's bands are stacked along
  int32_t recv_p1;               // region rule: 1 + the receiver rank (0: none yet)
  int32_t interleave;            // region rule: 1 once a frame's load was out of balance (max > 1.3 x share): new territory
                                 // is handed out by the greedy rule from then on (a camera that sweeps)
  uint32_t cur[64];              // k_rank: the voxels THIS frame touches in blocks each rank owns (cleared by k_shard_assign)
};
static_assert(sizeof(ShardHdr) <= kShardHdrBytes, "shard state header");
struct ShardState {
  ShardHdr* hdr;
  uint8_t* table;
  uint32_t* blk_w;
  uint32_t* new_list;
  int64_t n_blocks;
};
__host__ __device__ inline void shard_block_dims(const int32_t n_xyz[3], int s, int nb[3]) {
  for (int a = 0; a < 3; ++a) nb[a] = (n_xyz[a] + (1 << s) - 1) >> s;
}
__host__ __device__ inline size_t shard_state_layout(const int32_t n_xyz[3], int s, char* base, ShardState* S) {
  int nb[3];
  shard_block_dims(n_xyz, s, nb);
  const int64_t n = (int64_t)nb[0] * nb[1] * nb[2];
  const size_t tab = ((size_t)n + 255) / 256 * 256;
  if (S) {
    S->hdr = (ShardHdr*)base;
    S->table = (uint8_t*)(base + kShardHdrBytes);
    S->blk_w = (uint32_t*)(base + kShardHdrBytes + tab);
    S->new_list = (uint32_t*)(base + kShardHdrBytes + tab + (size_t)n * 4);
    S->n_blocks = n;
  }
  // the new-block list is sorted in place padded to a power of two (k_shard_assign): room for that
  size_t list = 1;
  while (list < (size_t)n && list < kNewListCap) list <<= 1;
  if (list < (size_t)n) list = (size_t)n;
  return kShardHdrBytes + tab + (size_t)n * 4 + list * 4;
}
// Walk order of a frame's new blocks under the region rule: bands are stacked along `axis` -- that block coordinate is
// the most significant, the other two follow in x, y, z order.  key <-> block index (x-major) of an nb[3] block grid.
__host__ __device__ inline uint32_t shard_walk_key(uint32_t b, const int nb[3], int axis) {
  if (axis == 0) return b;
  const uint32_t bz = b % (uint32_t)nb[2], by = (b / (uint32_t)nb[2]) % (uint32_t)nb[1], bx = b / ((uint32_t)nb[2] * (uint32_t)nb[1]);
  return axis == 1 ? (by * (uint32_t)nb[0] + bx) * (uint32_t)nb[2] + bz : (bz * (uint32_t)nb[0] + bx) * (uint32_t)nb[1] + by;
}
__host__ __device__ inline uint32_t shard_walk_block(uint32_t key, const int nb[3], int axis) {
  if (axis == 0) return key;
  uint32_t bx, by, bz;
  if (axis == 1) {
    bz = key % (uint32_t)nb[2];
    bx = (key / (uint32_t)nb[2]) % (uint32_t)nb[0];
    by = key / ((uint32_t)nb[2] * (uint32_t)nb[0]);
  } else {
    by = key % (uint32_t)nb[1];
    bx = (key / (uint32_t)nb[1]) % (uint32_t)nb[0];
    bz = key / ((uint32_t)nb[1] * (uint32_t)nb[0]);
  }
  return (bx * (uint32_t)nb[1] + by) * (uint32_t)nb[2] + bz;
}
// the rule that pins blocks nobody has touched yet (neighbours of touched blocks): a lattice rule spreads any
// axis-aligned stretch of blocks evenly over the ranks
__host__ __device__ inline int shard_lattice_owner(int bx, int by, int bz, int world) {
  return (int)(((unsigned)bx + 5u * (unsigned)by + 7u * (unsigned)bz) % (unsigned)world);
}

// Shard owner of a voxel.  Hash rule: hash of its block coordinate.  First-touch rule: the
// table's entry, -1 for a voxel outside the grid or in a block without an owner (neither can hold a row that an
// emitted voxel's decode reads: see bnv_grid_t.shard_state).
__device__ __forceinline__ int voxel_owner(int x, int y, int z, const bnv_grid_t& g) {
  if (g.shard_world <= 1) return 0;
  const int s = g.shard_block_log2;
  if (g.shard_state) {
    if ((unsigned)x >= (unsigned)g.n_xyz[0] || (unsigned)y >= (unsigned)g.n_xyz[1] || (unsigned)z >= (unsigned)g.n_xyz[2])
      return -1;
    const int m = (1 << s) - 1;
    const int nby = (g.n_xyz[1] + m) >> s, nbz = (g.n_xyz[2] + m) >> s;
    const uint8_t t = ((const uint8_t*)g.shard_state + kShardHdrBytes)[((x >> s) * nby + (y >> s)) * nbz + (z >> s)];
    return (t & kOwnAssigned) ? (int)(t & kOwnRank) : -1;
  }
  uint64_t b = ((uint64_t)(uint32_t)(x >> s) << 42) | ((uint64_t)(uint32_t)(y >> s) << 21) |
               (uint64_t)(uint32_t)(z >> s);
  return (int)(mix64(b) % (uint32_t)g.shard_world);
}

// shard_is_boundary for the moment when some blocks may still be without an owner (first-touch ownership, between
// the frame's rank kernel and its k_shard_assign): 0 no, 1 yes, 2 cannot say yet -- the voxel's own block or a block
// of its neighbourhood has no owner.  Neighbours outside the grid hold no voxel and do not count.
__device__ __forceinline__ int shard_boundary_state(int x, int y, int z, const bnv_grid_t& g) {
  if (g.shard_world <= 1) return 0;
  const int m = (1 << g.shard_block_log2) - 1;
  const int bx = x & m, by = y & m, bz = z & m;
  const int me = voxel_owner(x, y, z, g);
  if (me < 0) return 2;
  if (bx != 0 && bx != m && by != 0 && by != m && bz != 0 && bz != m) return 0;  // interior of its block
  bool other = false, open = false;
  for (int dx = (bx == 0 ? -1 : 0); dx <= (bx == m ? 1 : 0); ++dx)
    for (int dy = (by == 0 ? -1 : 0); dy <= (by == m ? 1 : 0); ++dy)
      for (int dz = (bz == 0 ? -1 : 0); dz <= (bz == m ? 1 : 0); ++dz) {
        if ((dx | dy | dz) == 0) continue;
        const int X = x + dx, Y = y + dy, Z = z + dz;
        if ((unsigned)X >= (unsigned)g.n_xyz[0] || (unsigned)Y >= (unsigned)g.n_xyz[1] || (unsigned)Z >= (unsigned)g.n_xyz[2])
          continue;
        const int o = voxel_owner(X, Y, Z, g);
        open |= o < 0;
        other |= o >= 0 && o != me;
      }
  return other ? 1 : (open ? 2 : 0);
}

// Is the voxel a BOUNDARY voxel of the sharding: does any voxel of its 3x3x3 neighbourhood belong to another
// rank?  (Purely a function of the coordinates: only voxels on the faces of their block can qualify.)  The decode of
// a voxel reads the rows of that neighbourhood, so boundary voxels are the ones whose rows other ranks need.
__device__ __forceinline__ bool shard_is_boundary(int x, int y, int z, const bnv_grid_t& g) {
  if (g.shard_world <= 1) return false;
  if (g.shard_state) return shard_boundary_state(x, y, z, g) != 0;   // (2 cannot occur for an emitted voxel)
  const int m = (1 << g.shard_block_log2) - 1;
  const int bx = x & m, by = y & m, bz = z & m;
  if (bx != 0 && bx != m && by != 0 && by != m && bz != 0 && bz != m) return false;  // interior of its block
  const int me = voxel_owner(x, y, z, g);
  for (int dx = (bx == 0 ? -1 : 0); dx <= (bx == m ? 1 : 0); ++dx)
    for (int dy = (by == 0 ? -1 : 0); dy <= (by == m ? 1 : 0); ++dy)
      for (int dz = (bz == 0 ? -1 : 0); dz <= (bz == m ? 1 : 0); ++dz)
        if ((dx | dy | dz) != 0 && voxel_owner(x + dx, y + dy, z + dz, g) != me) return true;
  return false;
}

// Does rank `rank` own a voxel of the 3x3x3 neighbourhood of (x, y, z) (the voxel itself included)?  Then it
// reads this voxel's row when it decodes, and keeps a copy of it (a ghost row) if the voxel is somebody else's.
__device__ __forceinline__ bool shard_adjacent_to(int x, int y, int z, const bnv_grid_t& g, int rank) {
  if (g.shard_world <= 1) return rank == 0;
  const int m = (1 << g.shard_block_log2) - 1;
  const int bx = x & m, by = y & m, bz = z & m;
  for (int dx = (bx == 0 ? -1 : 0); dx <= (bx == m ? 1 : 0); ++dx)
    for (int dy = (by == 0 ? -1 : 0); dy <= (by == m ? 1 : 0); ++dy)
      for (int dz = (bz == 0 ? -1 : 0); dz <= (bz == m ? 1 : 0); ++dz)
        if (voxel_owner(x + dx, y + dy, z + dz, g) == rank) return true;
  return false;
}

// ---- boundary records of the sharded exchange (shard.hip; appended by the frame upsert in volume.hip) ----------------
struct ShardRec {   // 48 bytes, 16-byte aligned: three dwordx4
  int32_t x, y, z;
  float w;
  float f[8];
};
static_assert(sizeof(ShardRec) == BNV_SHARD_RECORD_BYTES, "record size");
// header record of a block: x = number of records, y = sender rank, z = 1 if the block overflowed its capacity

// workspace words of the lattice decode that kernels of other files touch (decode.hip: lattice_ws_layout)
void lattice_ws_frame_words(void* ws, int64_t row_capacity, int32_t** origin_stamp, int32_t** ctl);

// ---- packed volume keys -------------------------------------------------------------------
constexpr uint64_t kEmptyKey = ~0ULL;
constexpr int64_t kKeyOffset = 1 << 20;

__device__ __forceinline__ bool pack_key(int64_t x, int64_t y, int64_t z, uint64_t* key) {
  const int64_t a = x + kKeyOffset, b = y + kKeyOffset, c = z + kKeyOffset;
  if ((uint64_t)a >= (1u << 21) || (uint64_t)b >= (1u << 21) || (uint64_t)c >= (1u << 21)) return false;
  *key = ((uint64_t)a << 42) | ((uint64_t)b << 21) | (uint64_t)c;
  return true;
}

// Row of a key in the slot table, or -1.  Linear probing; rows < 0 are "being inserted".
__device__ __forceinline__ int volume_find(const uint64_t* __restrict__ slot_keys,
                                           const int32_t* __restrict__ slot_rows, uint32_t mask,
                                           uint64_t key) {
  uint32_t s = mix64(key) & mask;
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    const uint64_t k = slot_keys[s];
    if (k == key) return slot_rows[s];
    if (k == kEmptyKey) return -1;
    s = (s + 1) & mask;
  }
  return -1;
}

// ---- dense row index of the grid (bnv_volume_t::brick) -----------------------------------------------------------
__device__ __forceinline__ bool brick_index(const bnv_volume_t& v, int64_t x, int64_t y, int64_t z, int64_t* idx) {
  if ((uint64_t)x >= (uint64_t)v.brick_dims[0] || (uint64_t)y >= (uint64_t)v.brick_dims[1] ||
      (uint64_t)z >= (uint64_t)v.brick_dims[2])
    return false;
  *idx = (x * v.brick_dims[1] + y) * v.brick_dims[2] + z;
  return true;
}

// a row was created for voxel (x, y, z)
__device__ __forceinline__ void brick_set(const bnv_volume_t& v, int64_t x, int64_t y, int64_t z, int32_t row) {
  int64_t idx;
  if (v.brick && brick_index(v, x, y, z, &idx)) v.brick[idx] = row;
}

// Row of voxel (x, y, z) or -1: from the brick when it is kept and the voxel is inside the grid, else from the hash.
__device__ __forceinline__ int volume_row(const bnv_volume_t& v, int64_t x, int64_t y, int64_t z) {
  int64_t idx;
  if (v.brick && brick_index(v, x, y, z, &idx)) return v.brick[idx];
  uint64_t key;
  if (!pack_key(x, y, z, &key)) return -1;
  return volume_find(v.slot_keys, v.slot_rows, (uint32_t)(v.n_slots - 1), key);
}

// ---- block-wide exclusive scan of one uint32 per thread (256 or 1024 threads) ---------------
template <int THREADS>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds_wave_totals,
                                                         uint32_t* block_total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) lds_wave_totals[wave] = incl;
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) {
    const uint32_t t = lds_wave_totals[w];
    if (w < wave) base += t;
    total += t;
  }
  __syncthreads();
  *block_total = total;
  return base + incl - v;
}

// ---- single-pass ordered prefix over the workgroups of ONE launch (decoupled look-back) -------------------------
// Replaces the three launches of a two-level scan (partial sums, top-level scan, apply).  state[tile] is ONE 64-bit
// word: bits 63..34 launch epoch | 33..32 flag (1 = this tile's aggregate, 2 = inclusive prefix up to and including
// this tile) | 31..0 value.  Words left behind by earlier launches carry older epochs and read as "not published
// yet", so the array is never cleared; every launch takes a fresh epoch from next_epoch().  Workgroups are
// dispatched in index order, so a predecessor is always resident or finished when a tile waits for it.
// lookback_exclusive is called by ALL lanes of the first wave of the block with the tile's aggregate (wave-uniform);
// `seed` is added to tile 0's prefix (e.g. the first free row).  Returns the exclusive prefix, in every lane.
uint32_t next_epoch();

__device__ __forceinline__ uint64_t lb_word(uint32_t epoch, uint32_t flag, uint32_t v) {
  return ((uint64_t)(epoch & 0x3fffffffu) << 34) | ((uint64_t)flag << 32) | (uint64_t)v;
}

__device__ __forceinline__ uint32_t lookback_exclusive(uint64_t* __restrict__ state, int tile, uint32_t agg,
                                                       uint32_t epoch, uint32_t seed = 0) {
  const int lane = threadIdx.x & 63;
  if (tile == 0) {
    if (lane == 0) __hip_atomic_store(&state[0], lb_word(epoch, 2, seed + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return seed;
  }
  if (lane == 0) __hip_atomic_store(&state[tile], lb_word(epoch, 1, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t want = epoch & 0x3fffffffu;
  uint32_t excl = 0;
  for (int hi = tile - 1;; hi -= 64) {
    const int p = hi - lane;   // lane 0 looks at the nearest predecessor
    uint64_t w = 0;
    if (p >= 0) {
      do {
        w = __hip_atomic_load(&state[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } while ((uint32_t)(w >> 34) != want);
    }
    const bool incl = p >= 0 && ((uint32_t)(w >> 32) & 3u) == 2u;
    const unsigned long long m = __ballot(incl);
    const int first = m ? (int)__ffsll((long long)m) - 1 : 64;   // lane of the nearest inclusive prefix
    uint32_t v = (p >= 0 && lane <= first) ? (uint32_t)w : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    excl += v;
    if (m) break;   // (tile 0 always publishes an inclusive prefix, so the walk ends at the latest there)
  }
  if (lane == 0)
    __hip_atomic_store(&state[tile], lb_word(epoch, 2, excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}

}  // namespace bnv
