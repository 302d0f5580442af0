// pipeline.hip -- one frame of the hot path behind a handful of C calls (host-side orchestration only; the kernels
// live in encode.hip / volume.hip / shard.hip / decode.hip / tsdf.hip).
//
// What NeuralMap.integrate (run_e2e.py:78-109) + the per-frame lattice decode (sparse_volume.py:697-738) cost the
// host when every stage is a Python call of its own: ~15 kernel launches through ~8 ctypes calls, a dozen tensor
// allocations, event objects, a pinned allocation -- and, in the sharded mode, one host wait in the MIDDLE of the
// frame (the exchange bound) behind which the GPU ran dry.  Here a frame in flight lives in a SLOT of caller-owned,
// persistent buffers and moves through up to FOUR HIP streams (round 4; two in round 3):
//
//   F (front stream)   begin:   front end + voxel marking, rank  ->  exchange bound to pinned memory (event)
//   E (encode stream)  begin:   point-encoder MLP + scatter, finalize  ->  TSDF side fusion
//   M (main stream)    upsert:  waits for the slot's encode; ONE launch = upsert + running average + decode-origin
//                               stamps + (sharded) boundary records appended to the slot's send block
//                      [the caller runs the frame's ONE all-gather on M: RCCL through torch.distributed]
//                      finish:  install ghost rows (resets the send block), neighbours + mark, table MLP
//   B (blend stream)   finish:  blend, counters / row count / evaluation count to pinned memory, done event
//
// Why four.  The two MLP kernels each fill a CU's LDS, so they can only take turns, and M is one dependent chain:
// upsert(t) -> all-gather -> install -> mark -> table(t) -> [blend(t)] -> upsert(t+1) ...  With two streams a rank's
// frame of an 8-GPU run was table 150 us + encoder 60 us + ~90 us in which only small kernels ran: the encoder of
// frame t+1 became ready exactly when table(t) did (its front end sat behind finalize + TSDF of frame t on the same
// stream) and the blend sat between table(t) and upsert(t+1).  Now the front end runs on a stream of its own, a
// frame or two ahead (the encode workspace is double-buffered), so the encoder of frame t+1 is ready as soon as
// frame t's has finished and runs BESIDE the M chain of frame t; the blend leaves M's chain for B (the decode
// workspace is double-buffered by the caller); and the encoder is launched on a share of the CUs
// (encoder_workgroups) so that the chain's small kernels find free CUs while it runs -- they cannot share a CU with
// it (LDS, registers).  A cycle is then ~ [encoder(t+1) beside chain(t)] + table(t).
//
// E of frame t+1 depends on the frame only; the host wait for the bound of frame t+1 (bnv_frame_bound) returns
// while M still holds most of frame t, and the GPU never waits for the host.  Nothing here allocates device memory;
// the object owns HIP events only.
#include <new>

#include "bnv_common.hpp"

using namespace bnv;

// the frame's read-backs in ONE small launch that writes the slot's pinned words directly (three hipMemcpyAsync of
// 4-32 bytes were three blit kernels of ~4.3 us each on the main stream)
__global__ void k_frame_readback(const int32_t* __restrict__ counters, const int32_t* __restrict__ status,
                                 const int32_t* __restrict__ evals, int32_t* __restrict__ host_words) {
  const int t = threadIdx.x;
  if (t < 8) host_words[BNV_PIPE_WORD_COUNTERS + t] = counters[t];
  else if (t < 10) host_words[BNV_PIPE_WORD_STATUS + t - 8] = status[t - 8];
  else if (t == 10) host_words[BNV_PIPE_WORD_EVALS] = evals ? *evals : 0;
  __threadfence_system();
}

// the same for callers that drive the stages themselves (NeuralMap): counters[8] | status[2] -> host_words[0..9]
__global__ void k_readback_words(const int32_t* __restrict__ counters, const int32_t* __restrict__ status,
                                 int32_t* __restrict__ host_words) {
  const int t = threadIdx.x;
  if (t < 8) host_words[t] = counters[t];
  else if (t < 10) host_words[t] = status[t - 8];
  __threadfence_system();
}

struct bnv_frame_pipe {
  bnv_frame_pipe_config_t cfg;
  int32_t* host_dev[BNV_PIPE_MAX_SLOTS];   // device-side address of the slots' pinned words (null: copy instead)
  hipStream_t F, E, M, B;                  // F == E, B == M when the config names no stream for them
  hipEvent_t ev_bound[BNV_PIPE_MAX_SLOTS], ev_enc[BNV_PIPE_MAX_SLOTS], ev_side[BNV_PIPE_MAX_SLOTS],
      ev_table[BNV_PIPE_MAX_SLOTS], ev_done[BNV_PIPE_MAX_SLOTS];
  hipEvent_t ev_encws[2];               // the encode workspace is free again (behind finalize of its last frame)
  bool encws_used[2];
  int enc_next;                         // encode workspace of the next frame begun
  int enc_buf[BNV_PIPE_MAX_SLOTS];
  void* lws_ptr[4];                     // decode workspaces seen and the slot whose frame used each one last
  int lws_slot[4];
  uint64_t lws_serial[4];               // serial of the frame that used the workspace last (0: none)
  uint64_t serial[BNV_PIPE_MAX_SLOTS];  // serial of the frame the slot holds (or held last)
  uint64_t next_serial;
  int last_decoded;                     // slot of the frame whose decode was enqueued last (-1: none)
  int state[BNV_PIPE_MAX_SLOTS];        // 0 free, 1 begun, 2 upserted, 3 finished (result pending)
  bool used[BNV_PIPE_MAX_SLOTS];        // ev_done has been recorded at least once
  int64_t n_points[BNV_PIPE_MAX_SLOTS];
  int32_t mlp_mode[BNV_PIPE_MAX_SLOTS];   // bnv_grid_t.mlp_mode a slot's frame was begun with (its decode uses the same)
  size_t bound_off;
  // diagnostic timeline (bnv_frame_pipe_timeline_enable): timing events per slot and point, a base event
  bool tl_on;
  hipEvent_t tl_base;
  hipEvent_t tl[BNV_PIPE_MAX_SLOTS][BNV_PIPE_TIMELINE_POINTS];
  bool tl_set[BNV_PIPE_MAX_SLOTS][BNV_PIPE_TIMELINE_POINTS];
  int encws_slot[2];                       // the slot whose frame used the encode workspace last
};

// `stream` waits for `ev` -- unless the event has already fired at the host's call (everything recorded in front of it is
// then complete: the barrier packet would only cost the stream a few microseconds of its serial chain; round 6,
// profiles/r06_experiments.txt [s1]: 0.304 -> 0.301 ms per frame for a rank of 8)
static hipError_t wait_event(const bnv_frame_pipe* p, hipStream_t stream, hipEvent_t ev) {
  (void)p;
  if (hipEventQuery(ev) == hipSuccess) return hipSuccess;
  (void)hipGetLastError();                 // (hipErrorNotReady is not an error)
  return hipStreamWaitEvent(stream, ev, 0);
}

// the pipe's grid with the arithmetic mode of the slot's frame
static bnv_grid_t slot_grid(const bnv_frame_pipe* p, int slot) {
  bnv_grid_t g = p->cfg.grid;
  g.mlp_mode = p->mlp_mode[slot];
  return g;
}

static void* slot_encws(const bnv_frame_pipe* p, int slot) {
  return p->enc_buf[slot] ? p->cfg.enc_ws2 : p->cfg.enc_ws;
}

static bool slot_ok(const bnv_frame_pipe* p, int slot) { return p && slot >= 0 && slot < p->cfg.n_slots; }

// timeline point k of the slot's frame on `st` (nothing unless the diagnostic is on)
static void tl_mark(bnv_frame_pipe* p, int slot, int k, hipStream_t st) {
  if (!p->tl_on) return;
  if (hipEventRecord(p->tl[slot][k], st) == hipSuccess) p->tl_set[slot][k] = true;
  else (void)hipGetLastError();
}

extern "C" {

int bnv_readback_words(const int32_t* counters, const int32_t* status, int32_t* host_words, bnv_stream_t stream) {
  if (!counters || !status || !host_words) return BNV_ERR_INVALID_ARGUMENT;
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, host_words, 0) == hipSuccess && d) {
    hipLaunchKernelGGL(k_readback_words, dim3(1), dim3(64), 0, (hipStream_t)stream, counters, status, (int32_t*)d);
    BNV_LAUNCH_CHECK();
    return BNV_OK;
  }
  (void)hipGetLastError();   // not mapped into the device's address space: two copies instead
  BNV_HIP_CHECK(hipMemcpyAsync(host_words, counters, 32, hipMemcpyDeviceToHost, (hipStream_t)stream));
  BNV_HIP_CHECK(hipMemcpyAsync(host_words + 8, status, 8, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return BNV_OK;
}

int bnv_frame_pipe_create(const bnv_frame_pipe_config_t* cfg, bnv_frame_pipe_t** out) {
  if (!cfg || !out || cfg->n_slots < 1 || cfg->n_slots > BNV_PIPE_MAX_SLOTS || !cfg->pointnet_pack || !cfg->enc_ws ||
      cfg->max_points < 1 || cfg->enc_ws_max_points < cfg->max_points || cfg->out_capacity < 1)
    return BNV_ERR_INVALID_ARGUMENT;
  if (cfg->grid.shard_world < 1 || cfg->grid.shard_world > 64 || !mlp_mode_field_ok(cfg->grid.mlp_mode))
    return BNV_ERR_INVALID_ARGUMENT;
  for (int s = 0; s < cfg->n_slots; ++s) {
    const bnv_frame_slot_t& b = cfg->slots[s];
    if (!b.feats || !b.pcounts || !b.flat_ids || !b.grid_ids || !b.counters || !b.host_words)
      return BNV_ERR_INVALID_ARGUMENT;
    if (cfg->grid.shard_world > 1 && (!b.send_block || cfg->send_capacity < 1)) return BNV_ERR_INVALID_ARGUMENT;
  }
  bnv_frame_pipe* p = new (std::nothrow) bnv_frame_pipe();
  if (!p) return BNV_ERR_CAPACITY;
  p->cfg = *cfg;
  p->E = (hipStream_t)cfg->encode_stream;
  p->M = (hipStream_t)cfg->main_stream;
  p->F = cfg->front_stream ? (hipStream_t)cfg->front_stream : p->E;
  p->B = cfg->blend_stream ? (hipStream_t)cfg->blend_stream : p->M;
  p->tl_on = false;
  p->tl_base = nullptr;
  for (int s = 0; s < BNV_PIPE_MAX_SLOTS; ++s)
    for (int k = 0; k < BNV_PIPE_TIMELINE_POINTS; ++k) {
      p->tl[s][k] = nullptr;
      p->tl_set[s][k] = false;
    }
  p->bound_off = bnv_encode_shard_counts_offset();
  p->enc_next = 0;
  for (int k = 0; k < 2; ++k) {
    p->ev_encws[k] = nullptr;
    p->encws_used[k] = false;
    p->encws_slot[k] = -1;
  }
  for (int k = 0; k < 4; ++k) {
    p->lws_ptr[k] = nullptr;
    p->lws_slot[k] = -1;
    p->lws_serial[k] = 0;
  }
  p->next_serial = 1;
  p->last_decoded = -1;
  for (int s = 0; s < BNV_PIPE_MAX_SLOTS; ++s) {
    p->serial[s] = 0;
    p->state[s] = 0;
    p->used[s] = false;
    p->n_points[s] = 0;
    p->enc_buf[s] = 0;
    p->mlp_mode[s] = cfg->grid.mlp_mode;
    p->ev_bound[s] = p->ev_enc[s] = p->ev_side[s] = p->ev_table[s] = p->ev_done[s] = nullptr;
    p->host_dev[s] = nullptr;
  }
  for (int k = 0; k < 2; ++k)
    if (hipEventCreateWithFlags(&p->ev_encws[k], hipEventDisableTiming) != hipSuccess) {
      bnv_frame_pipe_destroy(p);
      return BNV_ERR_HIP;
    }
  for (int s = 0; s < cfg->n_slots; ++s) {
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, cfg->slots[s].host_words, 0) == hipSuccess) p->host_dev[s] = (int32_t*)d;
    else (void)hipGetLastError();
  }
  for (int s = 0; s < cfg->n_slots; ++s) {
    hipEvent_t* evs[5] = {&p->ev_bound[s], &p->ev_enc[s], &p->ev_side[s], &p->ev_table[s], &p->ev_done[s]};
    for (hipEvent_t* e : evs)
      if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) {
        bnv_frame_pipe_destroy(p);
        return BNV_ERR_HIP;
      }
  }
  *out = p;
  return BNV_OK;
}

int bnv_frame_pipe_set_mlp_mode(bnv_frame_pipe_t* p, int32_t grid_mlp_mode) {
  if (!p || !mlp_mode_field_ok(grid_mlp_mode)) return BNV_ERR_INVALID_ARGUMENT;
  p->cfg.grid.mlp_mode = grid_mlp_mode;
  return BNV_OK;
}

int bnv_frame_pipe_destroy(bnv_frame_pipe_t* p) {
  if (!p) return BNV_OK;
  if (p->tl_base) (void)hipEventDestroy(p->tl_base);
  for (int s = 0; s < BNV_PIPE_MAX_SLOTS; ++s)
    for (int k = 0; k < BNV_PIPE_TIMELINE_POINTS; ++k)
      if (p->tl[s][k]) (void)hipEventDestroy(p->tl[s][k]);
  for (int s = 0; s < BNV_PIPE_MAX_SLOTS; ++s) {
    hipEvent_t evs[5] = {p->ev_bound[s], p->ev_enc[s], p->ev_side[s], p->ev_table[s], p->ev_done[s]};
    for (hipEvent_t e : evs)
      if (e) (void)hipEventDestroy(e);
  }
  for (int k = 0; k < 2; ++k)
    if (p->ev_encws[k]) (void)hipEventDestroy(p->ev_encws[k]);
  delete p;
  return BNV_OK;
}

// the part of `begin` behind the voxelisation, common to depth and point frames
static int begin_tail(bnv_frame_pipe* p, int slot, const float* pts, int64_t n, int image_width) {
  const bnv_frame_pipe_config_t& c = p->cfg;
  const bnv_frame_slot_t& b = c.slots[slot];
  void* enc_ws = slot_encws(p, slot);
  if (c.grid.shard_world > 1) {
    // the exchange bound: touched boundary voxels per rank, identical on every rank, known before the encoder MLP
    BNV_HIP_CHECK(hipMemcpyAsync(b.host_words + BNV_PIPE_WORD_BOUNDS, (const char*)enc_ws + p->bound_off,
                                 4 * (size_t)c.grid.shard_world, hipMemcpyDeviceToHost, p->F));
  }
  tl_mark(p, slot, 1, p->F);   // (in front of the event E waits for: the timeline's point 2 can then never precede it)
  BNV_HIP_CHECK(hipEventRecord(p->ev_bound[slot], p->F));
  if (p->E != p->F) BNV_HIP_CHECK(hipStreamWaitEvent(p->E, p->ev_bound[slot], 0));
  tl_mark(p, slot, 2, p->E);
  const bnv_grid_t g = slot_grid(p, slot);
  // (with the timeline on, the two parts are enqueued separately so that a mark fits between them: the same launches)
  const bool two = p->tl_on;
  int rc = bnv_encode_finish_image_parts(pts, n, image_width, &g, c.pointnet_pack, enc_ws, c.enc_ws_bytes,
                                         c.enc_ws_max_points, b.feats, b.pcounts, b.flat_ids, b.grid_ids,
                                         c.out_capacity, 0, b.counters, c.encoder_workgroups, two ? 1 : 3, p->E);
  if (rc != BNV_OK) return rc;
  tl_mark(p, slot, 3, p->E);
  if (two) {
    rc = bnv_encode_finish_image_parts(pts, n, image_width, &g, c.pointnet_pack, enc_ws, c.enc_ws_bytes,
                                       c.enc_ws_max_points, b.feats, b.pcounts, b.flat_ids, b.grid_ids, c.out_capacity,
                                       0, b.counters, c.encoder_workgroups, 2, p->E);
    if (rc != BNV_OK) return rc;
  }
  tl_mark(p, slot, 4, p->E);
  BNV_HIP_CHECK(hipEventRecord(p->ev_enc[slot], p->E));
  BNV_HIP_CHECK(hipEventRecord(p->ev_encws[p->enc_buf[slot]], p->E));   // finalize has left the workspace clean
  p->encws_used[p->enc_buf[slot]] = true;
  p->encws_slot[p->enc_buf[slot]] = slot;
  p->n_points[slot] = n;
  p->state[slot] = 1;
  return BNV_OK;
}

static int begin_head(bnv_frame_pipe* p, int slot) {
  if (!slot_ok(p, slot) || p->state[slot] != 0) return BNV_ERR_INVALID_ARGUMENT;
  // the slot's buffers are still read by the blend of the frame that used it last
  if (p->used[slot]) BNV_HIP_CHECK(hipStreamWaitEvent(p->F, p->ev_done[slot], 0));
  // the encode workspace alternates when there are two: the front end of this frame then only waits for the encoder
  // of the frame before the last one
  const int buf = p->cfg.enc_ws2 ? p->enc_next : 0;
  if (p->cfg.enc_ws2) p->enc_next ^= 1;
  p->enc_buf[slot] = buf;
  if (p->encws_used[buf] && p->F != p->E) BNV_HIP_CHECK(hipStreamWaitEvent(p->F, p->ev_encws[buf], 0));
  p->mlp_mode[slot] = p->cfg.grid.mlp_mode;   // the frame keeps the mode it starts under (bnv_frame_pipe_set_mlp_mode)
  p->serial[slot] = p->next_serial++;
  for (int k = 0; k < BNV_PIPE_TIMELINE_POINTS; ++k) p->tl_set[slot][k] = false;
  tl_mark(p, slot, 0, p->F);
  return BNV_OK;
}

// The TSDF side fusion of the slot's frame (run_e2e.py:99-109), gated on the device by the frame's in-bounds point
// count: enqueued on E behind the frame's encode, and the slot's side event recorded behind it.
static int side_depth(bnv_frame_pipe* p, int slot, const void* depth, int depth_dtype, int H, int W,
                      const double* intr_host, const double* T_wc_host, const float* color_im) {
  const bnv_frame_pipe_config_t& c = p->cfg;
  const bnv_frame_slot_t& b = c.slots[slot];
  if (c.tsdf.tsdf) {
    float K[9], T[16];
    for (int i = 0; i < 9; ++i) K[i] = (float)intr_host[i];
    for (int i = 0; i < 16; ++i) T[i] = (float)T_wc_host[i];
    const int32_t* gate = &b.counters->n_valid_points;
    int rc;
    if (depth_dtype == 0)
      rc = bnv_tsdf_integrate_u16(c.tsdf.tsdf, c.tsdf.weight, color_im ? c.tsdf.color : nullptr, c.tsdf.dim,
                                  c.tsdf.origin, c.tsdf.voxel_size, c.tsdf.trunc_margin, (const uint16_t*)depth,
                                  color_im, H, W, K, T, 1.0f, (float)c.max_depth, gate, p->E);
    else
      rc = bnv_tsdf_integrate(c.tsdf.tsdf, c.tsdf.weight, color_im ? c.tsdf.color : nullptr, c.tsdf.dim, c.tsdf.origin,
                              c.tsdf.voxel_size, c.tsdf.trunc_margin, (const float*)depth, color_im, H, W, K, T, 1.0f,
                              (float)c.max_depth, gate, p->E);
    if (rc != BNV_OK) return rc;
  }
  BNV_HIP_CHECK(hipEventRecord(p->ev_side[slot], p->E));
  return BNV_OK;
}

int bnv_frame_begin_depth(bnv_frame_pipe_t* p, int slot, const void* depth, int depth_dtype, int H, int W,
                          const double* intr_host, const double* T_wc_host, const float* color_im) {
  if (!slot_ok(p, slot)) return BNV_ERR_INVALID_ARGUMENT;
  const bnv_frame_pipe_config_t& c = p->cfg;
  const bnv_frame_slot_t& b = c.slots[slot];
  if (!b.input_pts || (int64_t)H * W > c.max_points || !intr_host || !T_wc_host) return BNV_ERR_INVALID_ARGUMENT;
  if (c.tsdf.tsdf && depth_dtype == 2) return BNV_ERR_INVALID_ARGUMENT;   // the TSDF kernel reads f32 / u16 images
  int rc = begin_head(p, slot);
  if (rc != BNV_OK) return rc;
  const bnv_grid_t g = slot_grid(p, slot);
  rc = bnv_encode_begin_depth(depth, depth_dtype, H, W, intr_host, T_wc_host, c.max_depth, &g, slot_encws(p, slot),
                              c.enc_ws_bytes, c.enc_ws_max_points, b.input_pts, p->F);
  if (rc != BNV_OK) return rc;
  rc = begin_tail(p, slot, b.input_pts, (int64_t)H * W, W);
  if (rc != BNV_OK) return rc;
  rc = side_depth(p, slot, depth, depth_dtype, H, W, intr_host, T_wc_host, color_im);
  if (rc != BNV_OK) return rc;
  return BNV_OK;
}

int bnv_frame_begin_points(bnv_frame_pipe_t* p, int slot, const float* input_pts, int64_t n_points) {
  int rc = begin_head(p, slot);
  if (rc != BNV_OK) return rc;
  const bnv_frame_pipe_config_t& c = p->cfg;
  if (!input_pts || n_points < 0 || n_points > c.max_points) return BNV_ERR_INVALID_ARGUMENT;
  const bnv_grid_t g = slot_grid(p, slot);
  rc = bnv_encode_begin(input_pts, n_points, &g, slot_encws(p, slot), c.enc_ws_bytes, c.enc_ws_max_points, p->F);
  if (rc != BNV_OK) return rc;
  rc = begin_tail(p, slot, input_pts, n_points, 0);
  if (rc != BNV_OK) return rc;
  BNV_HIP_CHECK(hipEventRecord(p->ev_side[slot], p->E));
  return BNV_OK;
}

int bnv_frame_side_depth(bnv_frame_pipe_t* p, int slot, const void* depth, int depth_dtype, int H, int W,
                         const double* intr_host, const double* T_wc_host, const float* color_im) {
  if (!slot_ok(p, slot) || p->state[slot] != 1 || !depth || !intr_host || !T_wc_host || H < 1 || W < 1)
    return BNV_ERR_INVALID_ARGUMENT;
  if (depth_dtype != 0 && depth_dtype != 1) return BNV_ERR_INVALID_ARGUMENT;   // the TSDF kernel reads u16 / f32 images
  if (!p->cfg.tsdf.tsdf) return BNV_OK;
  return side_depth(p, slot, depth, depth_dtype, H, W, intr_host, T_wc_host, color_im);
}

int bnv_frame_upsert(bnv_frame_pipe_t* p, int slot, const bnv_volume_t* vol, void* vol_ws, size_t vol_ws_bytes,
                     void* lattice_ws, int32_t lattice_epoch) {
  if (!slot_ok(p, slot) || p->state[slot] != 1 || !vol) return BNV_ERR_INVALID_ARGUMENT;
  const bnv_frame_pipe_config_t& c = p->cfg;
  const bnv_frame_slot_t& b = c.slots[slot];
  BNV_HIP_CHECK(wait_event(p, p->M, p->ev_enc[slot]));
  tl_mark(p, slot, 5, p->M);
  if (lattice_ws && p->B != p->M) {
    // the upsert stamps the decode's origins into lattice_ws and clears its control words: the blend (other stream)
    // of the frame that decoded into this workspace last must be through.  Callers alternate two workspaces, so
    // this is the blend of the frame before the last one.
    int k = 0, free_k = -1, lru_k = 0;
    for (; k < 4; ++k) {
      if (p->lws_ptr[k] == lattice_ws) break;
      if (!p->lws_ptr[k] && free_k < 0) free_k = k;
      if (p->lws_serial[k] < p->lws_serial[lru_k]) lru_k = k;
    }
    if (k == 4) {
      // a workspace not seen before; with more than four, the entry used longest ago makes room (after waiting for
      // the frame that used it)
      k = free_k >= 0 ? free_k : lru_k;
      if (free_k < 0 && p->lws_slot[k] >= 0 && p->used[p->lws_slot[k]])
        BNV_HIP_CHECK(hipStreamWaitEvent(p->M, p->ev_done[p->lws_slot[k]], 0));
      p->lws_ptr[k] = lattice_ws;
      p->lws_slot[k] = -1;
      p->lws_serial[k] = 0;
    }
    const int prev = p->lws_slot[k];
    if (prev >= 0 && prev != slot) {
      // The frame that used this workspace last sat in slot `prev`.  If that very frame is still in the slot and its
      // finish has not been enqueued, its done event does not exist yet: a caller's ordering error.  If the slot has
      // moved on to a later frame (collected and begun again, in any state), the old frame's finish was enqueued
      // long ago and ev_done[prev] -- recorded by it, or by a later finish in that slot -- covers its blend.
      if (p->serial[prev] == p->lws_serial[k] && (p->state[prev] == 1 || p->state[prev] == 2))
        return BNV_ERR_INVALID_ARGUMENT;
      if (p->used[prev]) BNV_HIP_CHECK(wait_event(p, p->M, p->ev_done[prev]));
    }
    p->lws_slot[k] = slot;
    p->lws_serial[k] = p->serial[slot];
  }
  bnv_integrate_extras_t x = {};
  if (c.grid.shard_world > 1) {
    x.shard_block = b.send_block;
    x.shard_block_capacity = c.send_capacity;
    x.grid_host = &c.grid;
  }
  x.lattice_ws = lattice_ws;
  x.stamp_epoch = lattice_epoch;
  const int rc = bnv_volume_integrate_frame(vol, b.grid_ids, b.feats, b.pcounts, c.out_capacity, &b.counters->n_out,
                                            vol_ws, vol_ws_bytes, &x, p->M);
  if (rc != BNV_OK) return rc;
  tl_mark(p, slot, 6, p->M);
  p->state[slot] = 2;
  return BNV_OK;
}

int bnv_frame_pipe_forget_workspaces(bnv_frame_pipe_t* p) {
  if (!p) return BNV_ERR_INVALID_ARGUMENT;
  for (int s = 0; s < p->cfg.n_slots; ++s)
    if (p->state[s] == 2) return BNV_ERR_INVALID_ARGUMENT;   // an upserted frame still needs its workspace
  for (int k = 0; k < 4; ++k) {
    if (p->lws_ptr[k] && p->lws_slot[k] >= 0 && p->used[p->lws_slot[k]])
      BNV_HIP_CHECK(hipStreamWaitEvent(p->M, p->ev_done[p->lws_slot[k]], 0));
    p->lws_ptr[k] = nullptr;
    p->lws_slot[k] = -1;
    p->lws_serial[k] = 0;
  }
  return BNV_OK;
}

int bnv_frame_cancel(bnv_frame_pipe_t* p, int slot) {
  if (!slot_ok(p, slot) || p->state[slot] != 1) return BNV_ERR_INVALID_ARGUMENT;
  // the encode side of the frame is enqueued and runs to its end (finalize leaves the encode workspace clean; a
  // sharded volume's owner table keeps the owners the frame gave to new blocks -- they are the same on every rank);
  // nothing of it reaches the volume.  The slot's done event covers the encode + side fusion, so the next frame begun
  // in the slot waits for them before it overwrites the slot's buffers.
  BNV_HIP_CHECK(hipStreamWaitEvent(p->B, p->ev_enc[slot], 0));
  BNV_HIP_CHECK(hipStreamWaitEvent(p->B, p->ev_side[slot], 0));
  BNV_HIP_CHECK(hipEventRecord(p->ev_done[slot], p->B));
  p->used[slot] = true;
  p->state[slot] = 0;
  return BNV_OK;
}

int bnv_frame_bound(bnv_frame_pipe_t* p, int slot, int32_t* max_bound_host) {
  if (!slot_ok(p, slot) || p->state[slot] < 1 || !max_bound_host) return BNV_ERR_INVALID_ARGUMENT;
  *max_bound_host = 0;
  if (p->cfg.grid.shard_world <= 1) return BNV_OK;
  BNV_HIP_CHECK(hipEventSynchronize(p->ev_bound[slot]));
  const int32_t* w = p->cfg.slots[slot].host_words + BNV_PIPE_WORD_BOUNDS;
  int32_t m = 0;
  for (int r = 0; r < p->cfg.grid.shard_world; ++r) m = w[r] > m ? w[r] : m;
  *max_bound_host = m;
  return BNV_OK;
}

int bnv_frame_finish(bnv_frame_pipe_t* p, int slot, const bnv_volume_t* vol, const void* blocks, int64_t block_capacity,
                     const float* sdfmlp_pack, const bnv_sdf_delta_t* delta, void* lattice_ws, size_t lattice_ws_bytes,
                     int32_t lattice_epoch) {
  if (!slot_ok(p, slot) || p->state[slot] != 2 || !vol) return BNV_ERR_INVALID_ARGUMENT;
  const bnv_frame_pipe_config_t& c = p->cfg;
  const bnv_frame_slot_t& b = c.slots[slot];
  // the exchanged blocks are prefixes of the slots' send blocks: more records than those hold were never sent
  if (blocks && block_capacity > c.send_capacity) return BNV_ERR_INVALID_ARGUMENT;
  int rc;
  tl_mark(p, slot, 7, p->M);
  if (c.grid.shard_world > 1 && blocks && block_capacity > 0) {
    rc = bnv_shard_install_reset(vol, &c.grid, blocks, c.grid.shard_world, block_capacity, b.send_block, p->M);
    if (rc != BNV_OK) return rc;
  }
  tl_mark(p, slot, 8, p->M);
  if (lattice_ws) {   // decode of the voxels the frame's upsert stamped, from the live rows
    if (!b.sdf || !sdfmlp_pack) return BNV_ERR_INVALID_ARGUMENT;
    const bnv_grid_t g = slot_grid(p, slot);
    if (vol->lattice_persist && vol->lattice_table && p->B != p->M && p->last_decoded >= 0 &&
        p->last_decoded != slot && p->used[p->last_decoded])
      // ONE table serves consecutive frames: this frame's table kernel overwrites entries the blend of the frame
      // decoded before it (blend stream) may still be reading
      BNV_HIP_CHECK(wait_event(p, p->M, p->ev_done[p->last_decoded]));
    p->last_decoded = slot;
    rc = bnv_decode_lattice_stamped_tables(vol, &g, vol->features, vol->weights, vol->row_capacity, sdfmlp_pack,
                                           b.grid_ids, c.out_capacity, &b.counters->n_out, lattice_ws,
                                           lattice_ws_bytes, lattice_epoch, p->M);
    if (rc != BNV_OK) return rc;
    tl_mark(p, slot, 9, p->M);
    if (p->B != p->M) {
      BNV_HIP_CHECK(hipEventRecord(p->ev_table[slot], p->M));
      BNV_HIP_CHECK(hipStreamWaitEvent(p->B, p->ev_table[slot], 0));
    }
    // the blend reads the workspace only: on B it leaves M to the next frame's upsert
    rc = bnv_lattice_blend(vol, &g, b.grid_ids, c.out_capacity, &b.counters->n_out, delta, lattice_ws, lattice_ws_bytes,
                           b.sdf, p->B);
    if (rc != BNV_OK) return rc;
  } else if (p->B != p->M) {
    BNV_HIP_CHECK(hipEventRecord(p->ev_table[slot], p->M));
    BNV_HIP_CHECK(hipStreamWaitEvent(p->B, p->ev_table[slot], 0));
  }
  int32_t *stamp = nullptr, *ctl = nullptr;
  if (lattice_ws) lattice_ws_frame_words(lattice_ws, vol->row_capacity, &stamp, &ctl);
  // (on B the row count may already include rows of the NEXT frame's upsert: the host's row bound stays an upper
  // bound -- it adds that frame's whole reservation on top until its own read-back settles it)
  if (p->host_dev[slot]) {
    hipLaunchKernelGGL(k_frame_readback, dim3(1), dim3(64), 0, p->B, (const int32_t*)b.counters,
                       (const int32_t*)vol->n_rows, ctl ? (const int32_t*)(ctl + 1) : (const int32_t*)nullptr,
                       p->host_dev[slot]);
    BNV_LAUNCH_CHECK();
  } else {
    if (ctl) BNV_HIP_CHECK(hipMemcpyAsync(b.host_words + BNV_PIPE_WORD_EVALS, ctl + 1, 4, hipMemcpyDeviceToHost, p->B));
    BNV_HIP_CHECK(hipMemcpyAsync(b.host_words + BNV_PIPE_WORD_COUNTERS, b.counters, sizeof(bnv_encode_counters_t),
                                 hipMemcpyDeviceToHost, p->B));
    BNV_HIP_CHECK(hipMemcpyAsync(b.host_words + BNV_PIPE_WORD_STATUS, vol->n_rows, 8, hipMemcpyDeviceToHost, p->B));
  }
  tl_mark(p, slot, 10, p->B);
  BNV_HIP_CHECK(hipStreamWaitEvent(p->B, p->ev_side[slot], 0));   // the frame's event covers its TSDF update too
  BNV_HIP_CHECK(hipEventRecord(p->ev_done[slot], p->B));
  p->used[slot] = true;
  p->state[slot] = 3;
  return BNV_OK;
}

int bnv_frame_result(bnv_frame_pipe_t* p, int slot, int32_t* words_host) {
  if (!slot_ok(p, slot) || p->state[slot] != 3) return BNV_ERR_INVALID_ARGUMENT;
  BNV_HIP_CHECK(hipEventSynchronize(p->ev_done[slot]));
  if (words_host)
    for (int i = 0; i < BNV_PIPE_HOST_WORDS; ++i) words_host[i] = p->cfg.slots[slot].host_words[i];
  p->state[slot] = 0;
  return BNV_OK;
}

int bnv_frame_ready(bnv_frame_pipe_t* p, int slot) {
  if (!slot_ok(p, slot) || p->state[slot] != 3) return BNV_ERR_INVALID_ARGUMENT;
  const hipError_t e = hipEventQuery(p->ev_done[slot]);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) return 0;
  g_last_hip_error = (int)e;
  return BNV_ERR_HIP;
}

int bnv_frame_pipe_timeline_enable(bnv_frame_pipe_t* p, int on) {
  if (!p) return BNV_ERR_INVALID_ARGUMENT;
  if (!on) {
    p->tl_on = false;
    return BNV_OK;
  }
  if (!p->tl_base) {
    BNV_HIP_CHECK(hipEventCreate(&p->tl_base));
    for (int s = 0; s < p->cfg.n_slots; ++s)
      for (int k = 0; k < BNV_PIPE_TIMELINE_POINTS; ++k) BNV_HIP_CHECK(hipEventCreate(&p->tl[s][k]));
  }
  BNV_HIP_CHECK(hipEventRecord(p->tl_base, p->M));
  p->tl_on = true;
  return BNV_OK;
}

int bnv_frame_timeline(bnv_frame_pipe_t* p, int slot, float* ms_host) {
  if (!slot_ok(p, slot) || !ms_host || !p->tl_base || p->state[slot] != 0) return BNV_ERR_INVALID_ARGUMENT;
  for (int k = 0; k < BNV_PIPE_TIMELINE_POINTS; ++k) {
    ms_host[k] = __builtin_nanf("");
    if (!p->tl_set[slot][k]) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p->tl_base, p->tl[slot][k]) == hipSuccess) ms_host[k] = ms;
    else (void)hipGetLastError();
  }
  return BNV_OK;
}

}  // extern "C"
