/*
 * bnv_fusion.h -- C ABI of the MI355X (gfx950) implementation of BNV-Fusion's per-frame
 * local-fusion + SDF-decode hot path.
 *
 * The reference (likojack/bnv_fusion) has no FFI for this path: the boundary there is a Python
 * duck type (LitFusionPointNet / SparseVolume).  The entry points below are what a binding of
 * those methods needs; each cites the reference code it replaces (paths relative to the
 * reference repo).  Conventions:
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns all memory (the library never allocates device memory);
 *   - every function enqueues work on `stream` (a hipStream_t passed as void*) and returns
 *     immediately; it returns 0 on success or a negative bnv_status code, never throws;
 *   - counts that are only known on the device are written to caller-provided device structs;
 *     the caller synchronises the stream before reading them.
 *   - build: hipcc --offload-arch=gfx950 (bnv_fusion_amd/csrc/build.py); no CPU fallback exists.
 */
#ifndef BNV_FUSION_H
#define BNV_FUSION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* bnv_stream_t; /* hipStream_t */

enum bnv_status {
  BNV_OK = 0,
  BNV_ERR_INVALID_ARGUMENT = -1,
  BNV_ERR_WORKSPACE_TOO_SMALL = -2,
  BNV_ERR_HIP = -3,          /* a HIP runtime call failed; see bnv_last_hip_error() */
  BNV_ERR_NOT_INITIALISED = -4,
  BNV_ERR_CAPACITY = -5
};

/* Voxel grid of one SparseVolume (sparse_volume.py:485-500, voxel_utils.py:83-88).  The float
 * members are the float32 values the reference holds: bound_min = float32(min_coords);
 * bound_lo = float32(min_coords) + float32(voxel), bound_hi = float32(max_coords) - float32(voxel)
 * evaluated in float32 (local_point_fusion.py:94-100). */
typedef struct bnv_grid {
  float bound_min[3];
  float bound_lo[3];
  float bound_hi[3];
  float voxel_size;
  int32_t n_xyz[3];
  int32_t min_pts_in_grid;
  /* spatial sharding of the active-voxel set: a voxel is owned by
   * hash(block coordinate) % shard_world, blocks of (1 << shard_block_log2)^3 voxels.
   * shard_world = 1 disables sharding. */
  int32_t shard_rank;
  int32_t shard_world;
  int32_t shard_block_log2;
  /* Arithmetic of the MLP kernels for calls made with this grid: 0 = the process default (bnv_set_mlp_mode),
   * BNV_GRID_MLP_MODE(m) = 1 + m selects mode m for these calls alone.  The mode belongs to the model (its weight
   * packs are laid out for it), so callers that hold several models -- or drive the library from several host
   * threads -- state it here instead of flipping the process default around their calls. */
  int32_t mlp_mode;
  /* Ownership rule of the sharding.  NULL: owner = hash(block coordinate) % shard_world, a pure function.  Else a
   * device buffer of bnv_shard_state_bytes() (zeroed once, the SAME evolving content on every rank -- the kernels
   * that update it work on the replicated voxelisation): FIRST-TOUCH ownership.  A block gets its owner in the frame
   * that first touches it: the new blocks of a frame, in ascending block order, go one by one to the rank that carries
   * the least load so far (load = touched voxels of a block in that frame); the 26 neighbour blocks of a newly
   * touched block that have no owner yet are pinned at the same moment to (bx + 5 by + 7 bz) % shard_world, so that
   * whenever a voxel is emitted every block of its 3x3x3 neighbourhood has an owner that never changes afterwards
   * (the boundary / ghost-row tests of the exchange rely on that).  Set by bnv_encode_begin* of the frame; every other
   * call only reads it.
   * That is BNV_SHARD_RULE_GREEDY, what a zeroed buffer means: blocks interleave finely, every rank holds an even
   * sample of ANY view (max / mean load 1.01-1.06) and ~1.2 x the single-GPU decode work is done in total (the halo).
   * bnv_shard_state_configure selects BNV_SHARD_RULE_REGION instead: contiguous regions.  cur[r] = the voxels THIS
   * frame touches in blocks rank r owns; a new block without an owner, in walk order (bands stacked along `axis`), goes
   * to the least-loaded not-full owner among its 26 neighbour blocks, else to the RECEIVER (the least-loaded rank, which
   * keeps that role until it is full: cur * world >= the frame's touched voxels); then the untouched neighbours of the
   * new blocks are pinned to the owner of the first new block (walk order) that reaches them -- regions grow outwards
   * -- or to the receiver when that owner carries more than 9/8 of its share.  Total decode work ~1.03-1.07 x single
   * GPU, a quarter of the boundary records; balanced while the view stays put or moves ACROSS the bands (1.04 on the
   * benchmark's pan with bands stacked along the vertical), not for a camera that sweeps a room (1.4-1.5).  Same
   * invariant, same determinism: every rank computes the same table. */
  void* shard_state;
} bnv_grid_t;
#define BNV_GRID_MLP_MODE(m) ((m) + 1)
#define BNV_SHARD_RULE_GREEDY 0
#define BNV_SHARD_RULE_REGION 1
/* Bytes of bnv_grid_t.shard_state for a grid of n_xyz voxels in blocks of (1 << block_log2)^3. */
size_t bnv_shard_state_bytes(const int32_t n_xyz[3], int32_t block_log2);
/* Selects the first-touch rule of a ZEROED shard_state buffer (before its first frame; the same call on every rank):
 * rule BNV_SHARD_RULE_*, axis 0 / 1 / 2 = the grid axis the region rule stacks its first bands along (the axis the
 * camera moves least along: the vertical).  Ordered on `stream` (and waited for: a set-up call). */
int bnv_shard_state_configure(void* shard_state, int32_t rule, int32_t axis, bnv_stream_t stream);
/* Byte offsets inside it: the per-rank loads (uint64[64]) and the owner table (one byte per block, index
 * (bx * nby + by) * nbz + bz; bits 0..5 owner, bit 6 assigned, bit 7 touched), for tools and tests. */
size_t bnv_shard_state_loads_offset(void);
size_t bnv_shard_state_table_offset(void);

/* Device-side result counters of bnv_encode_pointcloud. */
typedef struct bnv_encode_counters {
  int32_t n_valid_points; /* points that passed the bounds mask                                 */
  int32_t n_unique;       /* U : voxels touched by >= 1 (point, corner) pair                    */
  int32_t n_out;          /* U': rows written (count >= min_pts_in_grid and owned by this shard) */
  float n_avg_pts;        /* mean pair count over all U voxels (local_point_fusion.py:143)       */
  int32_t error;          /* != 0: an output capacity was exceeded                               */
  int32_t reserved[3];    /* [0]: sharded encode: (point, corner) pairs this shard encoded; others 0 */
} bnv_encode_counters_t;

/* Sparse feature volume = open-addressing hash (packed 3x21-bit key -> row) + dense row arrays
 * (replaces open3d.core.HashMap, sparse_volume.py:588-594).  All arrays are caller-owned. */
typedef struct bnv_volume {
  uint64_t* slot_keys;   /* [n_slots]  packed key, ~0 = empty                 */
  int32_t* slot_rows;    /* [n_slots]  row index                              */
  int64_t n_slots;       /* power of two                                      */
  int64_t* row_coords;   /* [row_capacity, 3] voxel coordinates (int64)       */
  float* features;       /* [row_capacity, n_feats]                           */
  float* weights;        /* [row_capacity]                                    */
  float* num_hits;       /* [row_capacity]                                    */
  int64_t row_capacity;
  int32_t* n_rows;       /* device int32[2]: {rows in use, sticky error of the upsert kernels: 0 ok, 1 slot table
                            full, 2 voxel coordinate outside the 21-bit key range, 3 row capacity exceeded};
                            bnv_volume_clear zeroes both                      */
  int32_t n_feats;       /* 8                                                 */
  /* Optional dense index of the volume's grid ("brick"): brick[(x * dims[1] + y) * dims[2] + z] = row of voxel
   * (x, y, z), or -1.  Maintained by every kernel that creates rows, for keys inside [0, dims); read by the decode
   * kernels in place of 27 hash probes per voxel (neighbouring voxels are neighbouring words: one cache line serves
   * a whole z-run, where every hash probe pulls a line of its own).  NULL: not kept (the hash is always complete). */
  int32_t* brick;
  int32_t brick_dims[3];
  /* Optional PERSISTENT lattice tables of the per-frame decode (bnv_frame_pipe_t; NULL: none).  lattice_table
   * [row_capacity * 27]: the SDF table entry of (row, lattice offset l), as bnv_lattice_table computes it;
   * lattice_have [row_capacity]: bit l set <=> that entry was computed from the row's CURRENT features.  Every kernel
   * of this library that writes a row's features clears the row's word (the upserts, the ghost-row install, insert);
   * the per-frame decode then evaluates only the entries a frame reads that are not there yet -- entries in rows the
   * frame did not update are carried over from earlier frames -- and the blend reads lattice_table.  A caller that
   * changes features by any other means zeroes lattice_have (the optimiser does not: it steps a COPY, to_tensor(), and
   * writes it back through insert, which clears the words); so does a caller that changes the SDF network's weights
   * (the entries are functions of both: SparseVolume.invalidate_tables, which the frame pipe calls when the model's
   * pack version moves).
   * bnv_volume_clear / bnv_volume_rehash leave both alone: the owner re-makes them with the row arrays. */
  float* lattice_table;
  uint32_t* lattice_have;
  /* 1: THIS call's lattice decode reads and extends the persistent tables (the frame pipeline sets it for its own
   * calls): its `features` argument must then be the volume's own array and the table stage must work on listed
   * entries -- anything else is rejected with BNV_ERR_INVALID_ARGUMENT (the three stages share ONE predicate);
   * 0: the call keeps to its workspace (every other decode: its features argument need not be the volume's). */
  int32_t lattice_persist;
} bnv_volume_t;

/* ------------------------------------------------------------------------------------------ */
int bnv_init(int device);           /* query the device, opt kernels in to large LDS          */
int bnv_num_compute_units(void);
const char* bnv_status_string(int status);
int bnv_last_hip_error(void);

/* Arithmetic of the two MLPs (point encoder, SDF decoder).  Both modes take fp32 inputs/weights and
 * accumulate in fp32:
 *   0  exact fp32: v_mfma_f32_32x32x2_f32 (bitwise an fp32 fmaf chain), 157 TFLOP/s peak;
 *   1  (default) split operands: every fp32 operand x = hi + lo with hi, lo in f16 (about 22 significant
 *      bits; f16 subnormals are kept), a.b ~ ah.bh + ah.bl + al.bh on the f16 MFMA (v_mfma_f32_16x16x32_f16 in
 *      the per-frame kernels, 32x32x16 in the generic decode and backward kernels): fp32-class accuracy
 *      (differences at the level of fp32 summation order) at 16/3 the MFMA rate;
 *   2  the tiny-cuda-nn networks of the reference's default checkpoint (pointnet_tcnn.ckpt): inputs
 *      padded with 1.0, 64-wide, no bias, fp16 weights and activations, f16 MFMA with fp32
 *      accumulation.  The pack buffers then hold the tcnn layouts (weights.py: pack_*_tcnn);
 *   3  the fp32 checkpoint with weights and activations rounded to f16 (the hi halves of mode 1's packs),
 *      ONE product per multiply-add on the f16 MFMA, fp32 accumulation: a third of mode 1's MFMAs and no lo
 *      conversions; per-layer relative error ~2^-11, SDF error ~1e-5 against the 1e-4 bar (features ~1e-3).
 *      The backward of decode_pts keeps the split arithmetic. */
/* The process DEFAULT: used by calls whose grid says mlp_mode = 0 (bnv_decode_dense: whose mlp_mode argument is 0).
 * A plain atomic word: setting it while another thread launches with mlp_mode = 0 changes that thread's arithmetic --
 * state the mode in the grid where that matters. */
int bnv_set_mlp_mode(int mode);
int bnv_get_mlp_mode(void);

/* Tuning switches (A/B experiments; defaults are the measured-best):
 *   "lattice_pipe"  1 (default): lattice-table MLP of modes 1 and 3 runs k_lattice_table_x (v_mfma_f32_16x16x32_f16,
 *                  operands prefetched across tiles and layers, dynamic tile hand-out); 0: the generic
 *                  k_decode<LATTICE> (32x32x16 MFMA; the same arithmetic in another summation grouping: tables equal
 *                  to ~1e-8, 8-10 % slower);
 *   "fused_mark"   -1 (default): bnv_decode_lattice looks the 27 neighbour rows up inside the live-entry marking kernel
 *                  for calls of up to 49,152 voxels and in a launch of its own above; 1 / 0 force either.
 *   "tcnn_block_encoder"  1 (default): whole-frame encodes with the tiny-cuda-nn networks run k_pointnet_scatter_tb
 *                  (32-point blocks x 8 corners, per-wave LDS accumulation of the voxel sums); 0: the per-tile kernel.
 *   "tcnn_shared_table"  1 (default): that kernel's 8 waves take the 8 blocks of a 16 x 16-pixel patch and sum them in
 *                  ONE LDS table per workgroup (flushed behind a barrier); 0: one table per wave, flushed per block.
 *   "finalize_blocks"  0 (default): the encoder's compaction kernel runs on 2 workgroups of 1,024 threads per CU
 *                  striding over its tiles; n > 0: on n workgroups, clamped to that same 2 per CU (tests force many
 *                  strides per workgroup with a small n).  The clamp is a forward-progress requirement, not a tuning
 *                  choice: a tile waits for its predecessor's look-back word, so every launched workgroup must be
 *                  resident at once.
 *   "reserve_cus"  0 (default); n: the persistent MLP kernels launch on (CUs - n) workgroups, leaving n CUs to
 *                  kernels of other streams (measured on one GPU with the two-stream frame pipeline: no gain for
 *                  n = 4, 8, 16 -- tools/ab_reserve.py; meant for an RCCL collective that must progress beside
 *                  the decode in the multi-GPU mode).
 * Unknown names return BNV_ERR_INVALID_ARGUMENT.  The switches are process-wide atomic words read at launch time:
 * they select between implementations that produce the same results (A/B timing), never the arithmetic. */
int bnv_set_option(const char* name, int value);

/* Optional timing of the dominant kernels with HIP events recorded on their launch stream.
 * kinds: 0 point-encoder MLP + scatter, 1 lattice-table SDF MLP, 2 decode_pts SDF MLP,
 * 3 dense-decode SDF MLP.  bnv_profile_read synchronises the recorded events and returns the
 * summed kernel time (ms) and launch count per kind since bnv_profile_enable(1). */
int bnv_profile_enable(int on);
int bnv_profile_read(double* total_ms_host /*[4]*/, int64_t* launches_host /*[4]*/);

/* Diagnostic (bench.py; no reference counterpart): the rate of f16 MFMAs this GPU sustains when NOTHING but
 * MFMAs is issued -- shape 0 = v_mfma_f32_32x32x16_f16 (`iters` x 12 per wave), 1 = v_mfma_f32_16x16x32_f16
 * (`iters` x 24: the same FLOPs), every CU, two waves per SIMD, operands 0 = zeros, 1 = uniform random f16.  The frame's MLP kernels run at the package power limit, where the clock (and the dense
 * MFMA rate) settles below the 2.4 GHz the 2.5 PFLOP/s peak is quoted at; this gives the ceiling that is actually
 * attainable on the box (the 16x16x32 form sustains ~14 % more than the 32x32x16 form), next to which bench.py
 * reports the dominant kernel.  Synchronises on `stream`; writes the
 * elapsed milliseconds and the FLOPs issued (2 x 32 x 32 x 16 per MFMA) to host memory. */
int bnv_probe_mfma_rate(int shape, int operands, int iters, void* stream, double* ms_host, double* flop_host);
/* Diagnostic: n_blocks single-wave workgroups that spin for `cycles` shader cycles on `stream`.  The runtime maps HIP
 * streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4) and two streams that share one are served
 * strictly in submission order; one spin on each of two streams, timed with events, tells whether they overlap.  The
 * frame pipeline picks its encode stream that way (bnv_fusion_amd/streams.py). */
int bnv_probe_spin(int n_blocks, int64_t cycles, void* stream);

/* ---- front end: depth image -> input_pts (FusionInferenceAbstractDataset.__getitem__,
 * src/datasets/fusion_inference_dataset.py:40-90; geometry.py:150-171; kornia depth_to_normals) --------
 * depth [H,W]: dtype 0 = uint16 millimetres (the dataset's PNG, /1000. as common.py:93), 1 = float32
 * metres, 2 = float64 metres.  intr 3x3 and T_wc 4x4 row-major float64 on the HOST.  Float64
 * arithmetic in the reference's order, rounded once to float32; out_pts [n_out, 6] holds the valid
 * pixels (0 < depth < max_depth) in row-major order, capacity H*W rows; n_out is a device int32. */
size_t bnv_depth_workspace_bytes(int H, int W);
int bnv_depth_to_points(const void* depth, int depth_dtype, int H, int W, const double* intr_host,
                        const double* T_wc_host, double max_depth, void* ws, size_t ws_bytes,
                        float* out_pts, int32_t* n_out, bnv_stream_t stream);
/* The same, and the rows [n_out, H*W) of out_pts are set to NaN by the same launch (a caller that hands the whole
 * H*W-row buffer on without reading n_out -- the encoder's bounds mask drops NaN rows -- needs no separate fill). */
int bnv_depth_to_points_padded(const void* depth, int depth_dtype, int H, int W, const double* intr_host,
                               const double* T_wc_host, double max_depth, void* ws, size_t ws_bytes,
                               float* out_pts, int32_t* n_out, bnv_stream_t stream);

/* ---- TSDF side fusion: TSDFVolume.integrate (third_parties/fusion.py:68-141, called per frame from
 * run_e2e.py:99-109).  tsdf / weight / color [dx,dy,dz] f32 (color may be NULL); depth_im [h,w] f32 metres
 * (0 = invalid); color_im [h,w] f32 folded b*65536+g*256+r or NULL; intr 3x3 and pose 4x4 row-major f32 on
 * the HOST; trunc_margin = 5 * voxel_size in the reference.  max_depth > 0: samples with depth >= max_depth are
 * invalid too -- the reference's loader zeroes them before the frame reaches either fusion (common.py:110-113 with
 * max_depth = model.ray_tracer.ray_max_dist, fusion_inference_dataset.py:28); <= 0: no cut.  gate: device int32 or
 * NULL; when *gate == 0 the launch does nothing -- NeuralMap.integrate returns before the TSDF fusion when the
 * encode found no in-bounds point (run_e2e.py:91-92), and the pipelined host does not know that count yet. */
int bnv_tsdf_integrate(float* tsdf, float* weight, float* color, const int32_t dim_host[3],
                       const float origin_host[3], float voxel_size, float trunc_margin,
                       const float* depth_im, const float* color_im, int im_h, int im_w,
                       const float intr_host[9], const float pose_host[16], float obs_weight,
                       float max_depth, const int32_t* gate, bnv_stream_t stream);
/* The same with the depth image as the dataset stores it: uint16 millimetres, converted per sample as
 * depth_mm / 1000 (common.py:93) inside the kernel -- no float copy of the frame. */
int bnv_tsdf_integrate_u16(float* tsdf, float* weight, float* color, const int32_t dim_host[3],
                       const float origin_host[3], float voxel_size, float trunc_margin,
                       const uint16_t* depth_mm, const float* color_im, int im_h, int im_w,
                       const float intr_host[9], const float pose_host[16], float obs_weight,
                       float max_depth, const int32_t* gate, bnv_stream_t stream);
/* n_frames (<= BNV_TSDF_BATCH_MAX) consecutive uint16 depth frames in ONE launch: identical to n_frames calls of
 * bnv_tsdf_integrate_u16 in order.  depth_mm: HOST array of device pointers; color / color_im: the colour volume and a
 * HOST array of device pointers to the frames' folded colour images (entries may be NULL), or both NULL;
 * intr_host [n_frames, 9] and pose_host [n_frames, 16] row-major f32 on the host. */
#define BNV_TSDF_BATCH_MAX 8
int bnv_tsdf_integrate_batch_u16(float* tsdf, float* weight, float* color, const int32_t dim_host[3],
                                 const float origin_host[3], float voxel_size, float trunc_margin, int n_frames,
                                 const uint16_t* const* depth_mm, const float* const* color_im, int im_h, int im_w,
                                 const float* intr_host,
                                 const float* pose_host, float obs_weight, float max_depth, bnv_stream_t stream);

/* ---- spatially sharded volume: the exchange step (new design; bnv_fusion_amd/distributed.py).
 * Voxels are owned by hash(block coordinate) % shard_world (bnv_grid_t).  Per frame every rank sends the rows it has
 * just updated that are BOUNDARY voxels (a voxel of their 3x3x3 neighbourhood belongs to another rank), one
 * all-gather moves the blocks, every rank installs the records adjacent to voxels it owns as ghost rows.
 * A block is (1 + capacity) records of BNV_SHARD_RECORD_BYTES: record 0 is the header {int32 count, int32 sender
 * rank, int32 overflow flag}; a record is {int32 x, y, z; float weight; float feature[8]}.
 *   bnv_shard_pack     coords [n, 3] i64 = this frame's emitted voxels (n_dev: device count or NULL), values read
 *                      from the volume AFTER the frame's upsert; capacity >= the bound bnv_encode_begin leaves in the
 *                      workspace for this rank;
 *   bnv_shard_install  blocks = world blocks back to back (the all-gather's output); the own block is skipped; a
 *                      sender's overflow flag sets the volume's sticky error word to 4. */
#define BNV_SHARD_RECORD_BYTES 48
int bnv_shard_pack(const bnv_volume_t* vol, const bnv_grid_t* grid, const int64_t* coords, int64_t n,
                   const int32_t* n_dev, void* block, int64_t capacity, bnv_stream_t stream);
int bnv_shard_install(const bnv_volume_t* vol, const bnv_grid_t* grid, const void* blocks, int world,
                      int64_t capacity, bnv_stream_t stream);
/* The same, and the header count of own_send_block (the block this rank contributed to `blocks`; may be NULL) is set
 * back to 0: the block is then ready for the records bnv_volume_integrate_frame appends for the next frame. */
int bnv_shard_install_reset(const bnv_volume_t* vol, const bnv_grid_t* grid, const void* blocks, int world,
                            int64_t capacity, void* own_send_block, bnv_stream_t stream);

/* ---- encode: LitFusionPointNet.encode_pointcloud (local_point_fusion.py:81-165) ------------ */

/* Bytes of scratch bnv_encode_pointcloud needs for up to max_points input points.  The scratch is
 * laid out for the max_points it was sized for: pass that same value as ws_max_points.
 * DESIGN LIMIT: the sorted-unique of the touched voxels is a bitmap rank, not a sort, so the scratch holds one
 * byte + one bit per voxel of the GRID (16.7 MB + 2 MB at 256^3, 134 MB + 16 MB at 512^3) next to the O(points)
 * arrays, flat voxel ids are int32, and grids of 2^31 voxels or more are rejected with BNV_ERR_INVALID_ARGUMENT
 * (about 1290^3).  The reference's torch.unique is O(points) with int64 ids (local_point_fusion.py:116-119) and has
 * no such bound; every BASELINE configuration (<= 512^3) is far inside it. */
size_t bnv_encode_workspace_bytes(int64_t max_points, const int32_t n_xyz[3]);
/* Zero the scratch (once after allocation; encode leaves it clean for the next frame). */
int bnv_encode_workspace_reset(void* ws, size_t ws_bytes, bnv_stream_t stream);

/* Packed-weight sizes (floats) of the fp32 point encoder / SDF decoder; layouts in DESIGN.md. */
size_t bnv_pointnet_pack_floats(void);
size_t bnv_sdfmlp_pack_floats(void);

/* The encode in two halves (bnv_encode_pointcloud below = begin + finish on the same arguments; everything between
 * the halves lives in the workspace):
 *   begin   bounds mask (local_point_fusion.py:94-104), 8-corner voxelisation (:153-165, modules.py:586-655) into a
 *           grid bitmap, sorted-unique of the touched voxels by bitmap rank (torch.unique, :118-119);
 *   finish  the point encoder on every (point, corner) pair, per-voxel mean, min-points filter, ordered repack.
 * bnv_encode_begin_depth fuses the depth front end in front of `begin` (depth arguments as bnv_depth_to_points): it
 * writes input_pts rows IN PIXEL ORDER into out_pts [H*W, 6] (NaN rows for invalid pixels -- nothing is compacted; the
 * bounds mask drops NaN rows wherever they are) and marks the voxels from registers; n_points of the matching
 * finish call is H*W.
 * With spatial sharding (grid.shard_world > 1) `begin` also leaves, at byte offset bnv_encode_shard_counts_offset()
 * of the workspace, int32[shard_world]: the number of touched BOUNDARY voxels each rank owns (voxels with a foreign
 * voxel in their 3x3x3 neighbourhood).  The voxelisation is replicated, so every rank holds the same numbers: an
 * upper bound, known before the encoder runs, of the boundary records each rank exchanges for the frame
 * (bnv_shard_pack).  `finish` clears them. */
int bnv_encode_begin(const float* input_pts, int64_t n_points, const bnv_grid_t* grid_host, void* ws, size_t ws_bytes,
                     int64_t ws_max_points, bnv_stream_t stream);
int bnv_encode_begin_depth(const void* depth, int depth_dtype, int H, int W, const double* intr_host,
                           const double* T_wc_host, double max_depth, const bnv_grid_t* grid_host, void* ws,
                           size_t ws_bytes, int64_t ws_max_points, float* out_pts, bnv_stream_t stream);
int bnv_encode_finish(const float* input_pts, int64_t n_points, const bnv_grid_t* grid_host,
                      const float* pointnet_pack, void* ws, size_t ws_bytes, int64_t ws_max_points,
                      float* out_feats, int64_t* out_pcounts, int64_t* out_flat_ids, int64_t* out_grid_ids,
                      int64_t out_capacity, int emit_all, bnv_encode_counters_t* counters, bnv_stream_t stream);
/* bnv_encode_finish for a frame whose points are the pixels of an image in row-major order (the out_pts of
 * bnv_encode_begin_depth): image_width > 0 with n_points a multiple of it lets the encoder work on 2-D pixel patches
 * (the tiny-cuda-nn encoder accumulates a patch's per-voxel sums on chip before touching the global accumulators:
 * neighbouring pixels in BOTH directions share voxels); 0 = no such structure (bnv_encode_finish).  Results are
 * identical either way (integer sums). */
int bnv_encode_finish_image(const float* input_pts, int64_t n_points, int image_width, const bnv_grid_t* grid_host,
                            const float* pointnet_pack, void* ws, size_t ws_bytes, int64_t ws_max_points,
                            float* out_feats, int64_t* out_pcounts, int64_t* out_flat_ids, int64_t* out_grid_ids,
                            int64_t out_capacity, int emit_all, bnv_encode_counters_t* counters, bnv_stream_t stream);
/* The same with the persistent point-encoder kernel launched on at most max_workgroups workgroups (one per CU;
 * 0 = every CU).  The encoder and the SDF-decoder kernels each fill a CU's LDS, so nothing else runs on a CU one of
 * them holds; the frame pipeline leaves a share of the CUs to the small kernels of the other streams this way. */
int bnv_encode_finish_image_wg(const float* input_pts, int64_t n_points, int image_width, const bnv_grid_t* grid_host,
                               const float* pointnet_pack, void* ws, size_t ws_bytes, int64_t ws_max_points,
                               float* out_feats, int64_t* out_pcounts, int64_t* out_flat_ids, int64_t* out_grid_ids,
                               int64_t out_capacity, int emit_all, bnv_encode_counters_t* counters,
                               int max_workgroups, bnv_stream_t stream);
/* The same in two parts that may go to different streams (the caller orders them: part 2 behind part 1):
 * parts & 1 = the point-encoder MLP + scatter, parts & 2 = finalize (mean, filter, ordered compaction, counters, clean
 * workspace); parts = 3 is bnv_encode_finish_image_wg.  The frame pipeline with CU-masked streams keeps only the
 * persistent MLP kernel on the encoder's share of the CUs. */
int bnv_encode_finish_image_parts(const float* input_pts, int64_t n_points, int image_width,
                                  const bnv_grid_t* grid_host, const float* pointnet_pack, void* ws, size_t ws_bytes,
                                  int64_t ws_max_points, float* out_feats, int64_t* out_pcounts, int64_t* out_flat_ids,
                                  int64_t* out_grid_ids, int64_t out_capacity, int emit_all,
                                  bnv_encode_counters_t* counters, int max_workgroups, int parts, bnv_stream_t stream);
size_t bnv_encode_shard_counts_offset(void);

/* input_pts [n_points, 6] f32 (world xyz, world normal).
 * Runs: bounds mask (:94-104), 8-corner voxelisation (:153-165, modules.py:586-655), the point
 * encoder on every (point, corner) pair (pointnet_utils.py:246-266, BatchNorm folded), sorted
 * unique voxel ids + counts (torch.unique, :118-119), per-voxel mean (torch_scatter.scatter_mean,
 * :125), min-points filter and repack (:143-151).
 * Outputs (capacity rows each): feats [.,8] f32, pcounts [.] i64, flat_ids [.] i64 ascending,
 * grid_ids [.,3] i64.  emit_all != 0 writes all U voxels with features zeroed where
 * count < min_pts (the return_dense=True contract, :126-141). */
int bnv_encode_pointcloud(const float* input_pts, int64_t n_points, const bnv_grid_t* grid_host,
                          const float* pointnet_pack, void* ws, size_t ws_bytes,
                          int64_t ws_max_points, float* out_feats, int64_t* out_pcounts, int64_t* out_flat_ids,
                          int64_t* out_grid_ids, int64_t out_capacity, int emit_all,
                          bnv_encode_counters_t* counters, bnv_stream_t stream);

/* get_relative_xyz + flatten for every pair, pair index p = corner * n_points + i
 * (local_point_fusion.py:106-117, voxel_utils.py:62-65).  No bounds mask (the caller compacts).
 * grid_ids [8n,3] i32, flat_ids [8n] i64, rel_xyz [8n,3] f32 = (xn - grid_id) * voxel_size.
 * Any output may be NULL. */
int bnv_voxelize_pairs(const float* input_pts, int64_t n_points, const bnv_grid_t* grid_host,
                       int32_t* grid_ids, int64_t* flat_ids, float* rel_xyz, uint8_t* bound_mask,
                       bnv_stream_t stream);

/* ---- volume: SparseVolume (sparse_volume.py:484-695) ---------------------------------------- */

int bnv_volume_clear(const bnv_volume_t* vol_host, bnv_stream_t stream);
/* Rebuild the slot table from row_coords[0:n_rows) (after the caller enlarged it). */
int bnv_volume_rehash(const bnv_volume_t* vol_host, bnv_stream_t stream);
size_t bnv_volume_workspace_bytes(int64_t max_keys);

/* LitFusionPointNet._integrate + _update (local_point_fusion.py:647-673) fused with
 * SparseVolume.query/insert (sparse_volume.py:561-585, 661-695): for n UNIQUE keys,
 * w = min(count/32, 1); f = (f_old*w_old + f*w)/(w_old + w); upsert.  coords [n,3] i64,
 * feats [n,8], pcounts [n] i64.  If n_dev (a device int32, e.g. &counters->n_out of the encode that
 * produced the inputs) is not NULL the element count is read on the device and n is only the
 * capacity the launch is sized for -- no host synchronisation between encode and integrate. */
int bnv_volume_integrate(const bnv_volume_t* vol_host, const int64_t* coords, const float* feats,
                         const int64_t* pcounts, int64_t n, const int32_t* n_dev, void* ws,
                         size_t ws_bytes, bnv_stream_t stream);

/* bnv_volume_integrate for the per-frame pipeline: the same single launch also (a) marks the rows it touches as the
 * ORIGINS of the frame's lattice decode in `lattice_ws` (the workspace bnv_decode_lattice_stamped is then called with,
 * same stamp_epoch != 0; NULL: no stamps) and (b) with a sharded volume appends the boundary voxels among the
 * upserted keys -- with the values just written -- to shard_block as bnv_shard_pack would (header count must be 0 or
 * hold the records of this frame so far: bnv_shard_install_reset leaves it so; NULL: no records).  Replaces the launches
 * of k_lattice_stamp and bnv_shard_pack (local_point_fusion.py:647-673 has no counterpart of either: new design). */
typedef struct bnv_integrate_extras {
  void* shard_block;             /* (1 + shard_block_capacity) records of BNV_SHARD_RECORD_BYTES, or NULL */
  int64_t shard_block_capacity;
  const bnv_grid_t* grid_host;   /* ownership predicates; needed with shard_block */
  void* lattice_ws;              /* lattice-decode workspace sized for vol->row_capacity, or NULL */
  int32_t stamp_epoch;
} bnv_integrate_extras_t;
int bnv_volume_integrate_frame(const bnv_volume_t* vol_host, const int64_t* coords, const float* feats,
                               const int64_t* pcounts, int64_t n, const int32_t* n_dev, void* ws, size_t ws_bytes,
                               const bnv_integrate_extras_t* extras_host, bnv_stream_t stream);

/* The same for up to BNV_VOLUME_BATCH_MAX consecutive frames in ONE call (4 launches): the result -- row order, row
 * count, features, weights -- is identical to calling bnv_volume_integrate once per frame in order (the replay loop of
 * the frame-parallel multi-GPU mode, and any caller that holds several encoded frames; the reference integrates one
 * frame per call, local_point_fusion.py:647-673).  coords / feats / pcounts / n / n_dev: HOST arrays of n_frames
 * device pointers / counts (n_dev may be NULL, or hold NULLs).  slot_mask [n_slots] u32 and slot_items
 * [n_slots * BNV_VOLUME_BATCH_MAX] i32 are side tables that belong to the volume: slot_mask must be all zeros
 * before the first call (it is all zeros again after every call) and both must be re-made when the slot table is.
 * Workspace: bnv_volume_workspace_bytes(sum over frames of n rounded up to 256). */
#define BNV_VOLUME_BATCH_MAX 8
int bnv_volume_integrate_batch(const bnv_volume_t* vol_host, int n_frames, const int64_t* const* coords,
                               const float* const* feats, const int64_t* const* pcounts, const int64_t* n,
                               const int32_t* const* n_dev, uint32_t* slot_mask, int32_t* slot_items,
                               void* workspace, size_t workspace_bytes, bnv_stream_t stream);
/* SparseVolume.insert (upsert with explicit values), sparse_volume.py:561-585. */
int bnv_volume_insert(const bnv_volume_t* vol_host, const int64_t* coords, const float* feats,
                      const float* weights, const float* num_hits, int64_t n, void* ws,
                      size_t ws_bytes, bnv_stream_t stream);
/* SparseVolume.query / _query_tensor (sparse_volume.py:625-695): zeros for absent keys.
 * Values are read from (features, weights, num_hits) given here -- the live arrays or a
 * to_tensor() snapshot; rows >= row_limit count as absent.  out_rows (optional) gets the row or -1. */
int bnv_volume_query(const bnv_volume_t* vol_host, const int64_t* coords, int64_t n,
                     const float* features, const float* weights, const float* num_hits,
                     int64_t row_limit, float* out_feats, float* out_weights, float* out_hits,
                     int32_t* out_rows, bnv_stream_t stream);
/* SparseVolume.count_optim (sparse_volume.py:602-622): weights[row] += 1 once per distinct row. */
int bnv_volume_count_optim(const bnv_volume_t* vol_host, const int64_t* coords, int64_t n,
                           float* weights, int64_t row_limit, int32_t* stamp, int32_t epoch,
                           bnv_stream_t stream);

/* ---- decode --------------------------------------------------------------------------------- */

/* Optional TSDF prior sampled by nearest grid_sample (sparse_volume.py:819-832). */
typedef struct bnv_sdf_delta {
  const float* data; /* [dx, dy, dz] or NULL */
  int32_t dims[3];
} bnv_sdf_delta_t;

/* SparseVolume.decode_pts (sparse_volume.py:768-833) at arbitrary points: coords [n,3] f32,
 * voxel units if is_coords else world; out [n] f32. */
int bnv_decode_pts(const bnv_volume_t* vol_host, const bnv_grid_t* grid_host, const float* features,
                   const float* weights, int64_t row_limit, const float* sdfmlp_pack,
                   const float* coords, int64_t n, int is_coords, const bnv_sdf_delta_t* delta_host,
                   float* out_sdf, bnv_stream_t stream);

/* Backward of bnv_decode_pts with respect to the volume features -- the autograd edge the global
 * optimiser relies on: run_e2e.py:111-162 turns volume.features into an nn.Parameter and
 * back-propagates render_utils.py:551-590 (calculate_loss) through SparseVolume.decode_pts
 * (sparse_volume.py:768-833).  grad_sdf [n] = d loss / d out_sdf; grad_features [row_limit, 8] is
 * ACCUMULATED into (the caller zeroes it).  The decoder weights are frozen and the points are data, so
 * nothing else receives gradient; sdf_delta is additive and does not enter.  sdfmlp_bwd_pack holds the
 * transposed layers: bnv_sdfmlp_bwd_pack_floats() floats for the fp32 decoder (weights.py: pack_sdf_mlp_bwd),
 * bnv_sdfmlp_tcnn_bwd_pack_floats() for the tiny-cuda-nn decoder of MLP mode 2 (pack_sdf_tcnn_bwd). */
size_t bnv_sdfmlp_bwd_pack_floats(void);
size_t bnv_sdfmlp_tcnn_bwd_pack_floats(void);
int bnv_decode_pts_backward(const bnv_volume_t* vol_host, const bnv_grid_t* grid_host,
                            const float* features, const float* weights, int64_t row_limit,
                            const float* sdfmlp_pack, const float* sdfmlp_bwd_pack, const float* coords,
                            int64_t n, int is_coords, const float* grad_sdf, float* grad_features,
                            bnv_stream_t stream);

/* ---- dataset formats (host only) ----------------------------------------------------------------- */

/* Reverses the PNG scanline filters of an inflated IDAT stream (the reference reads its 16-bit depth PNGs with
 * cv2.imread(path, -1), src/utils/common.py:93; bnv_fusion_amd/datasets.py parses the container).  raw: height
 * scanlines of 1 filter byte + row_bytes data bytes; bpp: bytes per pixel; out: height * row_bytes bytes. */
int bnv_png_unfilter(const uint8_t* raw, int height, int row_bytes, int bpp, uint8_t* out);

/* ---- global optimiser: ray sampling + SDF ray loss, fused ------------------------------------------ */

/* One ray split of render_utils.py:461-549 up to the decode.  Per ray: the ray through pixel uv
 * (get_camera_params / lift, :411-458; T_wc 4x4 and intr 3x3 row-major on the host), n_fine stratified
 * samples within +-truncated_dist of the observed surface point gt_pts and n_coarse between the camera and
 * it (hierarchical_sampling, :191-233; u_fine / u_coarse are the uniforms in [0,1) of the strata, supplied
 * by the caller), merged by distance -> pts [n, S, 3] (S = n_fine + n_coarse <= 64).  Per sample the L1
 * target of compute_sdf_loss (:508-549): +-distance to the nearest valid neighbour point (nb_pts
 * [n, n_nb, 3], nb_mask [n, n_nb]) clipped to +-truncated_dist, and weight = valid * ray_mask. */
int bnv_ray_samples(const float* uv, const float* gt_pts, const float* ray_mask, const float* nb_pts,
                    const float* nb_mask, int n_nb, const float T_wc[16], const float intr[9],
                    const float* u_fine, const float* u_coarse, int n, int n_fine, int n_coarse,
                    float truncated_dist, float* pts, float* target, float* weight, bnv_stream_t stream);
/* loss += sum_i weight_i |pred_i - target_i| / *n_valid (device scalars; loss is accumulated atomically)
 * and grad_i = d loss / d pred_i. */
int bnv_ray_loss(const float* pred, const float* target, const float* weight, const float* n_valid,
                 int64_t m, float* loss, float* grad, bnv_stream_t stream);
/* SparseVolume.count_optim on the 8 corner voxels of m sample points (render_utils.py:491-493,
 * sparse_volume.py:602-622): weights[row] += 1 once per distinct row. */
int bnv_volume_count_optim_pts(const bnv_volume_t* vol_host, const bnv_grid_t* grid_host, const float* pts,
                               int64_t m, int is_coords, float* weights, int64_t row_limit,
                               int32_t* stamp, int32_t epoch, bnv_stream_t stream);

/* ---- one optimiser step in a handful of launches (round 6).  NeuralMap.optimize (run_e2e.py:111-162) cuts a step's
 * 5,000 rays into splits of train_ray_splits = 1,000 and runs render_with_rays -> count_optim -> decode_pts -> loss ->
 * backward split by split (render_utils.py:461-590).  What couples the splits is count_optim: a split's mask
 * decisions (min over the 8 corner weights >= min_pts, sparse_volume.py:816-818) see the +1s of the splits before it
 * and its own.  These entries run ALL splits of a step together and keep every decision: sample q belongs to split
 * q / split_samples (at most 31 splits);
 *   bnv_volume_count_optim_splits  bit s of split_mask[row] (uint32 per row of the to_tensor() snapshot, zeroed once by
 *                                  the caller) = split s touches the row; weights are NOT changed yet;
 *   bnv_decode_pts_splits / bnv_decode_pts_backward_splits   bnv_decode_pts / _backward whose corner weights are
 *                                  weights[row] + 1 per split <= the query's that touches the row (exact fp32 adds);
 *   bnv_ray_loss_splits            bnv_ray_loss with one n_valid per split;
 *   bnv_optim_step                 forward + loss + backward of all samples in ONE kernel (fp32 decoder, split-f16
 *                                  arithmetic as bnv_decode_pts_backward): the L1 loss is elementwise, so a query's
 *                                  gradient is known in the tile that computes its value.  loss_and_counter: float[2],
 *                                  zeroed by the caller -- [0] accumulates the sum of the splits' losses, [1] is the
 *                                  kernel's chunk counter; pred (optional) [n] gets the decoded SDF; grad_features is
 *                                  accumulated into like bnv_decode_pts_backward's;
 *   bnv_volume_apply_split_counts  weights[row] += 1 per set bit (sequential adds), masks cleared: the state the
 *                                  split-by-split sequence leaves. */
int bnv_volume_count_optim_splits(const bnv_volume_t* vol_host, const bnv_grid_t* grid_host, const float* pts, int64_t m,
                                  int is_coords, int64_t row_limit, int64_t split_samples, uint32_t* split_mask,
                                  bnv_stream_t stream);
int bnv_volume_apply_split_counts(float* weights, uint32_t* split_mask, int64_t n_rows, bnv_stream_t stream);
int bnv_decode_pts_splits(const bnv_volume_t* vol_host, const bnv_grid_t* grid_host, const float* features,
                          const float* weights, int64_t row_limit, const float* sdfmlp_pack, const float* coords,
                          int64_t n, int is_coords, const bnv_sdf_delta_t* delta_host, const uint32_t* split_mask,
                          int64_t split_samples, float* out_sdf, bnv_stream_t stream);
int bnv_decode_pts_backward_splits(const bnv_volume_t* vol_host, const bnv_grid_t* grid_host, const float* features,
                                   const float* weights, int64_t row_limit, const float* sdfmlp_pack,
                                   const float* sdfmlp_bwd_pack, const float* coords, int64_t n, int is_coords,
                                   const uint32_t* split_mask, int64_t split_samples, const float* grad_sdf,
                                   float* grad_features, bnv_stream_t stream);
int bnv_ray_loss_splits(const float* pred, const float* target, const float* weight, const float* n_valid, int64_t m,
                        int64_t split_samples, float* loss, float* grad, bnv_stream_t stream);
int bnv_optim_step(const bnv_volume_t* vol_host, const bnv_grid_t* grid_host, const float* features,
                   const float* weights, int64_t row_limit, const float* sdfmlp_pack, const float* sdfmlp_bwd_pack,
                   const float* pts, int64_t n, int is_coords, const bnv_sdf_delta_t* delta_host,
                   const uint32_t* split_mask, int64_t split_samples, const float* target, const float* sample_weight,
                   const float* n_valid, float* loss_and_counter, float* pred, float* grad_features,
                   bnv_stream_t stream);

/* ---- per-voxel marching cubes on decoded lattices -------------------------------------------- */

/* SparseVolume.meshlize after the decode (sparse_volume.py:740-756): for every voxel whose 3x3x3
 * lattice sdf [n, 27] straddles `level` (max > level and min < level) the 8 cells are triangulated;
 * vertex = level crossing of a lattice edge (linear), in world units:
 * ((index + t) * 0.5 + origin - 0.5) * voxel_size + min_coords.  tri_table: int8 [256, 16] edge triples,
 * -1 terminated (bnv_fusion_amd/mc_tables.py; scikit-image's Lewiner tables are not available here, so
 * the triangulation inside ambiguous cells is that table's).  Two passes: bnv_mc_count writes the
 * triangle count of every voxel; the caller prefix-sums them (exclusive, int64) and sizes `vertices`
 * [3 T, 3] f32; bnv_mc_emit writes the triangle soup in voxel / cell / table order. */
int bnv_mc_count(const float* sdf, int64_t n, const int32_t* n_dev, float level, const int8_t* tri_table,
                 int32_t* counts, bnv_stream_t stream);
int bnv_mc_emit(const float* sdf, const int64_t* origins, int64_t n, const int32_t* n_dev, float level,
                float voxel_size, const float min_coords[3], const int8_t* tri_table,
                const int64_t* tri_offsets, float* vertices, bnv_stream_t stream);
/* The same meshes with shared vertices, laid out as SparseVolume.meshlize concatenates them
 * (sparse_volume.py:740-756): per voxel one vertex per sign-changing edge of its 3x3x3 lattice (ascending lattice-edge
 * order: x edges, y edges, z edges, node-major) and faces that index them, offset by the vertex counts of the voxels
 * before it (the reference's `faces + last_face_id; last_face_id += max(faces) + 1`).  count -> the caller's
 * exclusive prefix sums (vert_offsets, tri_offsets [n] i64) -> emit into vertices [V, 3] f32, faces [T, 3] i64. */
int bnv_mc_count_indexed(const float* sdf, int64_t n, const int32_t* n_dev, float level, const int8_t* tri_table,
                         int32_t* n_verts, int32_t* n_tris, bnv_stream_t stream);
int bnv_mc_emit_indexed(const float* sdf, const int64_t* origins, int64_t n, const int32_t* n_dev, float level,
                        float voxel_size, const float min_coords[3], const int8_t* tri_table,
                        const int64_t* vert_offsets, const int64_t* tri_offsets, float* vertices, int64_t* faces,
                        bnv_stream_t stream);

size_t bnv_decode_lattice_workspace_bytes(int64_t n_voxels, int64_t row_capacity);
/* Byte offset, inside that workspace, of two int32 device counters: [0] rows listed by
 * bnv_lattice_neighbors(build_list), [1] table entries (= MLP evaluations) listed by bnv_lattice_mark /
 * evaluated by the last bnv_decode_lattice (for FLOP accounting). */
size_t bnv_decode_lattice_count_offset(int64_t row_capacity);
/* The same decode for the 3x3x3 lattice {-0.5,0,0.5}^3 around n integer voxel origins
 * (SparseVolume.meshlize's decode_pts call, sparse_volume.py:717-738): out [n,27] f32.
 * Evaluates the MLP once per (corner voxel, local offset) instead of 8x per lattice point.
 * n_dev: optional device-side count as in bnv_volume_integrate (n = capacity). */
int bnv_decode_lattice(const bnv_volume_t* vol_host, const bnv_grid_t* grid_host,
                       const float* features, const float* weights, int64_t row_limit,
                       const float* sdfmlp_pack, const int64_t* origins, int64_t n, const int32_t* n_dev,
                       const bnv_sdf_delta_t* delta_host, void* ws, size_t ws_bytes, int32_t epoch,
                       float* out_sdf, bnv_stream_t stream);

/* bnv_decode_lattice for origins that bnv_volume_integrate_frame has already stamped in `ws` with this `epoch` (the
 * origins are exactly the keys of that upsert): one launch less. */
int bnv_decode_lattice_stamped(const bnv_volume_t* vol_host, const bnv_grid_t* grid_host,
                               const float* features, const float* weights, int64_t row_limit,
                               const float* sdfmlp_pack, const int64_t* origins, int64_t n, const int32_t* n_dev,
                               const bnv_sdf_delta_t* delta_host, void* ws, size_t ws_bytes, int32_t epoch,
                               float* out_sdf, bnv_stream_t stream);

/* bnv_decode_lattice_stamped without its last stage: neighbour rows, live entries and the table MLP on `stream`; the
 * caller finishes with bnv_lattice_blend(vol, grid, origins, n, n_dev, delta, ws, ws_bytes, out_sdf, other_stream)
 * behind an event -- the blend reads nothing but `ws`, so on a stream of its own it leaves `stream` free for the next
 * frame's upsert (the frame pipeline does this; that next frame must then decode into another workspace). */
int bnv_decode_lattice_stamped_tables(const bnv_volume_t* vol_host, const bnv_grid_t* grid_host,
                                      const float* features, const float* weights, int64_t row_limit,
                                      const float* sdfmlp_pack, const int64_t* origins, int64_t n,
                                      const int32_t* n_dev, void* ws, size_t ws_bytes, int32_t epoch,
                                      bnv_stream_t stream);

/* The three stages of bnv_decode_lattice, callable separately so that a sharded volume can exchange
 * corner-voxel tables between them (bnv_fusion_amd/distributed.py).  They share one workspace:
 *   neighbors: row of each of the 27 neighbour voxels of every origin (-1: absent or weight below
 *              min_pts); with build_list, appends each distinct usable row (not flagged in
 *              row_skip) to the work list;
 *   mark:      (single-volume path) flags the (row, l) table entries read by LIVE lattice points --
 *              all 8 corner voxels usable -- and lists them; masked points cost no MLP work;
 *   table:     table[row][l] = SDF-MLP(enc(l), features[row]) * voxel for every listed entry
 *              (use_entries) or all 27 l of every listed row;
 *   blend:     out[n,27] from the neighbour rows and the table. */
int bnv_lattice_neighbors(const bnv_volume_t* vol_host, const bnv_grid_t* grid_host, const float* weights,
                          int64_t row_limit, const int64_t* origins, int64_t n, const int32_t* n_dev,
                          const uint8_t* row_skip, int build_list, void* ws, size_t ws_bytes, int32_t epoch,
                          bnv_stream_t stream);
int bnv_lattice_mark(const bnv_volume_t* vol_host, int64_t n, const int32_t* n_dev, void* ws, size_t ws_bytes,
                     int32_t epoch, bnv_stream_t stream);
int bnv_lattice_table(const bnv_volume_t* vol_host, const bnv_grid_t* grid_host, const float* features,
                      const float* sdfmlp_pack, int64_t n_voxels, int use_entries, void* ws, size_t ws_bytes,
                      bnv_stream_t stream);
int bnv_lattice_blend(const bnv_volume_t* vol_host, const bnv_grid_t* grid_host, const int64_t* origins,
                      int64_t n, const int32_t* n_dev, const bnv_sdf_delta_t* delta_host, void* ws,
                      size_t ws_bytes, float* out_sdf, bnv_stream_t stream);
/* Byte offsets of the table [row_capacity,27] f32 and of the row work list inside the workspace. */
size_t bnv_decode_lattice_table_offset(int64_t row_capacity);
size_t bnv_decode_lattice_list_offset(int64_t n_voxels, int64_t row_capacity);

/* LitFusionPointNet.decode_feature_grid_w_pts (local_point_fusion.py:265-370) on dense grids feat_grid [8,X,Y,Z],
 * pts_weight [X,Y,Z]; voxel_coords [n,3] f32 (voxel units) -> out_sdf [n].  `variant` selects the reference's branch:
 *   BNV_DENSE_CORNERS  global_coords=False, interpolate_decode=True (:281-329, the yaml configuration): 8 corner
 *                      evaluations per query, blended; a corner counts when its weight >= min_pts_in_grid, the query is
 *                      valid when ANY corner counts, otherwise out = voxel_size.  out_feats must be NULL.
 *   BNV_DENSE_NEAREST  global_coords=False, interpolate_decode=False (:288-292, 331-343): ONE evaluation at the
 *                      nearest voxel (torch.round, half to even), valid when that voxel's weight >= min_pts_in_grid.
 *   BNV_DENSE_GLOBAL   global_coords=True, the signature default (:345-367): features by trilinear grid_sample
 *                      (align_corners=True, zero padding), weight by nearest; the MLP sees voxel_coords / (res - 1),
 *                      the prediction is NOT scaled by voxel_size; valid when the nearest weight >= min_pts_in_grid.
 * out_feats (optional, NEAREST / GLOBAL) [n,8]: the features the evaluation used (the reference's second return value).
 * status (optional, device int32[2]): status[1] = 5 when a feature leaves the certified range of the split arithmetic
 * (the range guard of bnv_decode_pts, which reports through the volume's error word).
 * gradient=True is not offered: the reference's branch (:367-369) reads an undefined name and cannot run. */
enum { BNV_DENSE_CORNERS = 0, BNV_DENSE_NEAREST = 1, BNV_DENSE_GLOBAL = 2 };
int bnv_decode_dense(const float* feat_grid, const float* pts_weight, const int32_t dims_host[3],
                     float voxel_size, int32_t min_pts_in_grid, const float* sdfmlp_pack,
                     const float* voxel_coords, int64_t n, int32_t variant, float* out_sdf, float* out_feats,
                     int32_t* status, bnv_stream_t stream);
/* The same with the arithmetic mode stated per call (bnv_grid_t.mlp_mode encoding: 0 = process default,
 * BNV_GRID_MLP_MODE(m) = mode m); bnv_decode_dense = mlp_mode 0. */
int bnv_decode_dense_mode(const float* feat_grid, const float* pts_weight, const int32_t dims_host[3],
                          float voxel_size, int32_t min_pts_in_grid, const float* sdfmlp_pack,
                          const float* voxel_coords, int64_t n, int32_t variant, int32_t mlp_mode, float* out_sdf,
                          float* out_feats, int32_t* status, bnv_stream_t stream);

/* ---- the per-frame chain as one object: NeuralMap.integrate (run_e2e.py:78-109: encode_pointcloud -> track_n_pts ->
 * _integrate -> TSDF side fusion) followed by the lattice decode of the frame's voxels (sparse_volume.py:697-738),
 * for one GPU or for one shard of a spatially sharded volume.  Replaces the per-stage calls
 * above on the per-frame path: a frame in flight occupies a SLOT of caller-owned persistent buffers and runs on two
 * streams -- the encode (a function of the frame only) on encode_stream, the volume-dependent part on main_stream --
 * so frame t+1's encode overlaps frame t's exchange + decode, and no stage needs a host-side allocation, event object
 * or size read.  Per frame and slot, in this order (bnv_frame_begin_* of the NEXT frame may come before
 * bnv_frame_bound of this one: the encode side then runs a frame ahead of the host):
 *   bnv_frame_begin_depth | _points   front_stream (encode_stream if none): front end + voxelise + rank; exchange bound
 *                                     -> pinned memory; encode_stream: point encoder + reduction + filter; TSDF side
 *                                     fusion (depth frames, if configured)
 *   bnv_frame_upsert                  main_stream: upsert + running average (+ boundary records into the slot's send
 *                                     block, + decode-origin stamps when lattice_ws is given)
 *   bnv_frame_bound                   HOST wait for the frame's exchange bound (max over ranks; 0 unsharded)
 *   [sharded: the caller all-gathers the first (1 + capacity) records of every rank's send block on main_stream]
 *   bnv_frame_finish                  main_stream: install ghost rows from `blocks`, neighbour rows + live entries +
 *                                     table MLP of the lattice decode (when lattice_ws is given); blend_stream
 *                                     (main_stream if none): blend into the slot's sdf, read-backs into the slot's
 *                                     pinned words
 *   bnv_frame_result                  HOST wait for the frame; copies the slot's pinned words out; frees the slot
 * The volume and its workspaces are passed per call (they are re-made when the volume grows).  The object owns HIP
 * events only; every buffer is the caller's and must outlive it. */
#define BNV_PIPE_MAX_SLOTS 8
#define BNV_PIPE_HOST_WORDS 96
/* int32 words of a slot's pinned area */
#define BNV_PIPE_WORD_COUNTERS 0  /* [8]  bnv_encode_counters_t of the frame                                     */
#define BNV_PIPE_WORD_STATUS 8    /* [2]  {volume row count, sticky upsert error} behind the frame               */
#define BNV_PIPE_WORD_EVALS 10    /* [1]  SDF-MLP evaluations of the frame's decode                              */
#define BNV_PIPE_WORD_BOUNDS 16   /* [shard_world] touched boundary voxels per rank (the exchange bound)         */

typedef struct bnv_frame_slot {
  float* input_pts;                 /* [max_points, 6]: written by the depth front end (may be NULL: point frames only) */
  float* feats;                     /* [out_capacity, 8]  */
  int64_t* pcounts;                 /* [out_capacity]     */
  int64_t* flat_ids;                /* [out_capacity]     */
  int64_t* grid_ids;                /* [out_capacity, 3]  */
  bnv_encode_counters_t* counters;  /* device             */
  float* sdf;                       /* [out_capacity, 27] or NULL (never decoding)                               */
  void* send_block;                 /* sharded: (1 + send_capacity) records, header {0, rank, 0} before first use */
  int32_t* host_words;              /* PINNED HOST memory, BNV_PIPE_HOST_WORDS int32                             */
} bnv_frame_slot_t;

typedef struct bnv_tsdf_desc {      /* TSDFVolume (third_parties/fusion.py:19-66); tsdf == NULL: no side fusion   */
  float* tsdf;
  float* weight;
  float* color;
  int32_t dim[3];
  float origin[3];
  float voxel_size, trunc_margin;
} bnv_tsdf_desc_t;

typedef struct bnv_frame_pipe_config {
  bnv_grid_t grid;
  int64_t max_points;               /* largest frame (H * W)                                                      */
  int64_t out_capacity;             /* rows of the slots' output arrays (>= 8 * max_points / min_pts + 1)         */
  int64_t send_capacity;            /* records of the slots' send blocks.  out_capacity is enough for what a rank
                                       SENDS (its emitted boundary voxels); the exchange bound of bnv_frame_bound
                                       counts TOUCHED boundary voxels and can be larger (small or sparse frames): the
                                       caller exchanges min(bound rounded up, send_capacity) records per rank, and
                                       bnv_frame_finish rejects a block_capacity above send_capacity             */
  const float* pointnet_pack;
  void* enc_ws;                     /* bnv_encode_workspace_bytes(enc_ws_max_points, grid.n_xyz), zeroed once     */
  size_t enc_ws_bytes;
  int64_t enc_ws_max_points;
  double max_depth;                 /* depth cut-off of the loader (common.py:110-113)                            */
  bnv_tsdf_desc_t tsdf;
  int32_t n_slots;
  bnv_frame_slot_t slots[BNV_PIPE_MAX_SLOTS];
  bnv_stream_t encode_stream, main_stream;
  /* Round 4 (all optional; zero = the two-stream pipeline of round 3):
   *   front_stream  the front end (voxelise + rank + exchange bound) runs here instead of on encode_stream, so the
   *                 encoder of the next frame is ready as soon as this frame's has finished;
   *   enc_ws2       a second encode workspace (same size as enc_ws, zeroed once): frames alternate, and the front end
   *                 of a frame only waits for the encoder of the frame before the last one;
   *   blend_stream  the decode's blend + the frame's read-backs run here instead of on main_stream, which goes
   *                 straight on to the next frame's upsert; the caller must then pass bnv_frame_upsert /
   *                 bnv_frame_finish alternating decode workspaces (the pipe orders a workspace's reuse itself);
   *   encoder_workgroups  the persistent point encoder is launched on this many workgroups (one per CU; 0 = all
   *                 CUs): the CUs it leaves are where the small kernels of the other streams run meanwhile. */
  void* enc_ws2;
  bnv_stream_t front_stream, blend_stream;
  int32_t encoder_workgroups;
} bnv_frame_pipe_config_t;

typedef struct bnv_frame_pipe bnv_frame_pipe_t;
/* A frame's read-backs in one launch for callers that drive the stages themselves: the encode's counters (8 int32)
 * and the volume's {row count, sticky error} (bnv_volume_t.n_rows) -> host_words[0..9], PINNED host memory (written
 * through its device mapping; two small copies if it has none).  Valid once `stream` has passed this point. */
int bnv_readback_words(const int32_t* counters, const int32_t* status, int32_t* host_words, bnv_stream_t stream);

int bnv_frame_pipe_create(const bnv_frame_pipe_config_t* config_host, bnv_frame_pipe_t** out);
int bnv_frame_pipe_destroy(bnv_frame_pipe_t* pipe);
/* Arithmetic mode (bnv_grid_t.mlp_mode encoding) of the frames begun from now on; a frame in flight keeps the mode it
 * was begun with through its decode.  The mode config.grid carried at creation applies until this is called. */
int bnv_frame_pipe_set_mlp_mode(bnv_frame_pipe_t* pipe, int32_t grid_mlp_mode);
/* depth as bnv_encode_begin_depth (dtype 0 = uint16 mm, 1 = float32 m); color_im: folded colour image or NULL.  The
 * device buffers of a frame (depth, color_im, input_pts) are read by kernels enqueued up to the frame's bnv_frame_upsert
 * (the TSDF side fusion): they must stay valid until the frame's result is in. */
int bnv_frame_begin_depth(bnv_frame_pipe_t* pipe, int slot, const void* depth, int depth_dtype, int H, int W,
                          const double* intr_host, const double* T_wc_host, const float* color_im);
int bnv_frame_begin_points(bnv_frame_pipe_t* pipe, int slot, const float* input_pts, int64_t n_points);
/* The TSDF side fusion (run_e2e.py:99-109) of a frame begun with bnv_frame_begin_points that ALSO carries its depth
 * image -- the reference dataset's frames hold input_pts, rgbd, intr_mat and T_wc together (run_e2e.py:78-109) -- gated
 * on the device by the frame's in-bounds point count like the depth path's.  Between the frame's begin and its upsert;
 * a no-op without a TSDF volume in the config.  depth_dtype 0 = uint16 mm, 1 = float32 m. */
int bnv_frame_side_depth(bnv_frame_pipe_t* pipe, int slot, const void* depth, int depth_dtype, int H, int W,
                         const double* intr_host, const double* T_wc_host, const float* color_im);
/* Abandons a frame that was begun but not upserted (an announced frame that never came, an error on the way): its
 * encode runs to its end and leaves the workspaces clean, nothing of it reaches the feature volume, the slot is free
 * again (the next frame begun in it waits for the abandoned work).  What the encode side has already done stays done:
 * a TSDF side fusion enqueued by begin, the owners a sharded volume's first-touch table gave to new blocks (the same
 * on every rank that began the frame). */
int bnv_frame_cancel(bnv_frame_pipe_t* pipe, int slot);
/* The caller has re-made its decode workspaces (the volume grew): the pipe forgets the pointers it has seen, after
 * ordering main_stream behind the frames that used them.  No frame may sit between its upsert and its finish. */
int bnv_frame_pipe_forget_workspaces(bnv_frame_pipe_t* pipe);
int bnv_frame_upsert(bnv_frame_pipe_t* pipe, int slot, const bnv_volume_t* vol_host, void* vol_ws, size_t vol_ws_bytes,
                     void* lattice_ws, int32_t lattice_epoch);
int bnv_frame_bound(bnv_frame_pipe_t* pipe, int slot, int32_t* max_bound_host);
int bnv_frame_finish(bnv_frame_pipe_t* pipe, int slot, const bnv_volume_t* vol_host, const void* blocks,
                     int64_t block_capacity, const float* sdfmlp_pack, const bnv_sdf_delta_t* delta_host,
                     void* lattice_ws, size_t lattice_ws_bytes, int32_t lattice_epoch);
int bnv_frame_result(bnv_frame_pipe_t* pipe, int slot, int32_t* words_host /* [BNV_PIPE_HOST_WORDS] or NULL */);
/* 1: bnv_frame_result would not block; 0: the frame is still running */
int bnv_frame_ready(bnv_frame_pipe_t* pipe, int slot);
/* Diagnostic: GPU timestamps of a frame's stages (tools/spatial_single_rank.py --timeline).  _enable(1) creates timing
 * events and records a base event on main_stream; every frame BEGUN afterwards records BNV_PIPE_TIMELINE_POINTS events
 * on its streams; bnv_frame_timeline, called after bnv_frame_result of that frame and before its slot is begun again,
 * returns their times in ms since the base (NaN: the frame did not pass that point).  Points: 0 front end starts,
 * 1 exchange bound copied (front stream), 2 encoder starts, 3 encoder done, 4 finalize done, 5 upsert starts (the
 * slot's encode has arrived on main_stream), 6 upsert done, 7 finish starts (the caller's all-gather is through),
 * 8 ghost rows installed, 9 table MLP done (marking + table), 10 blend + read-back done.  The extra marker packets
 * cost a few us per frame: off by default. */
#define BNV_PIPE_TIMELINE_POINTS 11
int bnv_frame_pipe_timeline_enable(bnv_frame_pipe_t* pipe, int on);
int bnv_frame_timeline(bnv_frame_pipe_t* pipe, int slot, float* ms_host /* [BNV_PIPE_TIMELINE_POINTS] */);

#ifdef __cplusplus
}
#endif
#endif /* BNV_FUSION_H */
