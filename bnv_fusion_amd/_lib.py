"""ctypes binding of libbnv_fusion_hip.so (C ABI: include/bnv_fusion.h).

There is no CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BNV_FUSION_LIB") or os.path.join(_HERE, "libbnv_fusion_hip.so")

# every symbol include/bnv_fusion.h declares
SYMBOLS = [
    "bnv_init", "bnv_num_compute_units", "bnv_status_string", "bnv_last_hip_error",
    "bnv_encode_workspace_bytes", "bnv_encode_workspace_reset", "bnv_pointnet_pack_floats",
    "bnv_sdfmlp_pack_floats", "bnv_encode_pointcloud", "bnv_encode_begin", "bnv_encode_begin_depth",
    "bnv_encode_finish", "bnv_encode_finish_image", "bnv_encode_shard_counts_offset", "bnv_shard_pack", "bnv_shard_install", "bnv_voxelize_pairs",
    "bnv_volume_clear", "bnv_volume_rehash", "bnv_volume_workspace_bytes", "bnv_volume_integrate",
    "bnv_volume_integrate_batch",
    "bnv_volume_insert", "bnv_volume_query", "bnv_volume_count_optim",
    "bnv_depth_workspace_bytes", "bnv_depth_to_points", "bnv_depth_to_points_padded", "bnv_tsdf_integrate", "bnv_tsdf_integrate_u16", "bnv_tsdf_integrate_batch_u16", "bnv_set_mlp_mode", "bnv_get_mlp_mode", "bnv_set_option", "bnv_profile_enable", "bnv_profile_read", "bnv_probe_mfma_rate", "bnv_probe_spin", "bnv_decode_lattice_count_offset",
    "bnv_decode_lattice_table_offset", "bnv_decode_lattice_list_offset", "bnv_lattice_neighbors",
    "bnv_lattice_mark", "bnv_lattice_table", "bnv_lattice_blend",
    "bnv_decode_pts", "bnv_sdfmlp_bwd_pack_floats", "bnv_sdfmlp_tcnn_bwd_pack_floats", "bnv_decode_pts_backward",
    "bnv_png_unfilter", "bnv_mc_count", "bnv_mc_emit", "bnv_mc_count_indexed", "bnv_mc_emit_indexed", "bnv_ray_samples", "bnv_ray_loss", "bnv_volume_count_optim_pts",
    "bnv_volume_count_optim_splits", "bnv_volume_apply_split_counts", "bnv_decode_pts_splits",
    "bnv_decode_pts_backward_splits", "bnv_ray_loss_splits", "bnv_optim_step",
    "bnv_decode_lattice_workspace_bytes", "bnv_decode_lattice", "bnv_decode_dense",
    "bnv_shard_install_reset", "bnv_volume_integrate_frame", "bnv_decode_lattice_stamped", "bnv_readback_words",
    "bnv_decode_dense_mode", "bnv_frame_pipe_set_mlp_mode", "bnv_decode_lattice_stamped_tables",
    "bnv_encode_finish_image_wg", "bnv_encode_finish_image_parts", "bnv_shard_state_bytes", "bnv_shard_state_loads_offset", "bnv_shard_state_table_offset",
    "bnv_frame_pipe_create", "bnv_frame_pipe_destroy", "bnv_frame_begin_depth", "bnv_frame_begin_points",
    "bnv_frame_upsert", "bnv_frame_bound", "bnv_frame_finish", "bnv_frame_result", "bnv_frame_ready", "bnv_frame_pipe_timeline_enable", "bnv_frame_timeline",
    "bnv_frame_side_depth", "bnv_frame_cancel", "bnv_frame_pipe_forget_workspaces", "bnv_shard_state_configure",
    
]


class Grid(C.Structure):
    _fields_ = [("bound_min", C.c_float * 3), ("bound_lo", C.c_float * 3), ("bound_hi", C.c_float * 3),
                ("voxel_size", C.c_float), ("n_xyz", C.c_int32 * 3), ("min_pts_in_grid", C.c_int32),
                ("shard_rank", C.c_int32), ("shard_world", C.c_int32), ("shard_block_log2", C.c_int32),
                ("mlp_mode", C.c_int32),      # 0: process default; 1 + m: arithmetic mode m for calls with this grid
                ("shard_state", C.c_void_p)]  # NULL: hash ownership; else the first-touch owner table (device)


class EncodeCounters(C.Structure):
    _fields_ = [("n_valid_points", C.c_int32), ("n_unique", C.c_int32), ("n_out", C.c_int32),
                ("n_avg_pts", C.c_float), ("error", C.c_int32), ("reserved", C.c_int32 * 3)]


class Volume(C.Structure):
    _fields_ = [("slot_keys", C.c_void_p), ("slot_rows", C.c_void_p), ("n_slots", C.c_int64),
                ("row_coords", C.c_void_p), ("features", C.c_void_p), ("weights", C.c_void_p),
                ("num_hits", C.c_void_p), ("row_capacity", C.c_int64), ("n_rows", C.c_void_p),
                ("n_feats", C.c_int32), ("brick", C.c_void_p), ("brick_dims", C.c_int32 * 3),
                ("lattice_table", C.c_void_p), ("lattice_have", C.c_void_p), ("lattice_persist", C.c_int32)]


class SdfDelta(C.Structure):
    _fields_ = [("data", C.c_void_p), ("dims", C.c_int32 * 3)]


class IntegrateExtras(C.Structure):
    _fields_ = [("shard_block", C.c_void_p), ("shard_block_capacity", C.c_int64), ("grid_host", C.POINTER(Grid)),
                ("lattice_ws", C.c_void_p), ("stamp_epoch", C.c_int32)]


class FrameSlot(C.Structure):
    _fields_ = [("input_pts", C.c_void_p), ("feats", C.c_void_p), ("pcounts", C.c_void_p), ("flat_ids", C.c_void_p),
                ("grid_ids", C.c_void_p), ("counters", C.c_void_p), ("sdf", C.c_void_p), ("send_block", C.c_void_p),
                ("host_words", C.c_void_p)]


class TsdfDesc(C.Structure):
    _fields_ = [("tsdf", C.c_void_p), ("weight", C.c_void_p), ("color", C.c_void_p), ("dim", C.c_int32 * 3),
                ("origin", C.c_float * 3), ("voxel_size", C.c_float), ("trunc_margin", C.c_float)]


class FramePipeConfig(C.Structure):
    _fields_ = [("grid", Grid), ("max_points", C.c_int64), ("out_capacity", C.c_int64), ("send_capacity", C.c_int64),
                ("pointnet_pack", C.c_void_p), ("enc_ws", C.c_void_p), ("enc_ws_bytes", C.c_size_t),
                ("enc_ws_max_points", C.c_int64), ("max_depth", C.c_double), ("tsdf", TsdfDesc),
                ("n_slots", C.c_int32), ("slots", FrameSlot * 8), ("encode_stream", C.c_void_p),
                ("main_stream", C.c_void_p), ("enc_ws2", C.c_void_p), ("front_stream", C.c_void_p),
                ("blend_stream", C.c_void_p), ("encoder_workgroups", C.c_int32)]


class BnvError(RuntimeError):
    pass


_lib = None
_initialised_device = None
fp32_mode = 1   # MLP mode of fp32-checkpoint models that do not name one themselves (0 exact fp32, 1 split-f16, 3 f16); see set_mlp_mode


def model_mode(m):
    """Arithmetic mode of a network object (a LitFusionPointNet or its ``nerf``): its own ``mlp_mode`` when it names
    one (tiny-cuda-nn networks: always 2; ``model.set_mlp_mode``), else the package default for fp32 checkpoints."""
    mode = getattr(m, "mlp_mode", None)
    return fp32_mode if mode is None else int(mode)


def grid_with_mode(grid, mode, cache=None):
    """A copy of a bnv_grid_t whose calls run in arithmetic mode ``mode`` (bnv_grid_t.mlp_mode = 1 + mode).  The mode
    travels with every call instead of living in a process global: two models of different arithmetic, or two host
    threads, do not interfere.  ``cache``: a dict owned by whoever owns ``grid`` (re-made when ``grid`` is)."""
    if cache is not None:
        if cache.get("base") is not grid:
            cache.clear()
            cache["base"] = grid
        g = cache.get(mode)
        if g is not None:
            return g
    g = Grid.from_buffer_copy(grid)
    g.mlp_mode = int(mode) + 1
    if cache is not None:
        cache[mode] = g
    return g


def load():
    """Loads the shared library (building is the job of __graft_entry__.build / csrc/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python bnv_fusion_amd/csrc/build.py` "
            "(hipcc, gfx950).  bnv_fusion_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i64, i32, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_size_t
    sig = {
        "bnv_init": (C.c_int, [C.c_int]),
        "bnv_num_compute_units": (C.c_int, []),
        "bnv_status_string": (C.c_char_p, [C.c_int]),
        "bnv_last_hip_error": (C.c_int, []),
        "bnv_encode_workspace_bytes": (sz, [i64, C.POINTER(i32)]),
        "bnv_encode_workspace_reset": (C.c_int, [vp, sz, vp]),
        "bnv_pointnet_pack_floats": (sz, []),
        "bnv_sdfmlp_pack_floats": (sz, []),
        "bnv_encode_pointcloud": (C.c_int, [vp, i64, C.POINTER(Grid), vp, vp, sz, i64, vp, vp, vp, vp, i64, C.c_int,
                                            vp, vp]),
        "bnv_encode_begin": (C.c_int, [vp, i64, C.POINTER(Grid), vp, sz, i64, vp]),
        "bnv_encode_begin_depth": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                             C.POINTER(C.c_double), C.c_double, C.POINTER(Grid), vp, sz, i64, vp, vp]),
        "bnv_encode_finish": (C.c_int, [vp, i64, C.POINTER(Grid), vp, vp, sz, i64, vp, vp, vp, vp, i64, C.c_int,
                                        vp, vp]),
        "bnv_encode_finish_image": (C.c_int, [vp, i64, C.c_int, C.POINTER(Grid), vp, vp, sz, i64, vp, vp, vp, vp, i64,
                                              C.c_int, vp, vp]),
        "bnv_encode_shard_counts_offset": (sz, []),
        "bnv_shard_pack": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, i64, vp, vp, i64, vp]),
        "bnv_shard_install": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, C.c_int, i64, vp]),
        "bnv_voxelize_pairs": (C.c_int, [vp, i64, C.POINTER(Grid), vp, vp, vp, vp, vp]),
        "bnv_volume_clear": (C.c_int, [C.POINTER(Volume), vp]),
        "bnv_volume_rehash": (C.c_int, [C.POINTER(Volume), vp]),
        "bnv_volume_workspace_bytes": (sz, [i64]),
        "bnv_volume_integrate": (C.c_int, [C.POINTER(Volume), vp, vp, vp, i64, vp, vp, sz, vp]),
        "bnv_volume_integrate_batch": (C.c_int, [C.POINTER(Volume), C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
        "bnv_volume_insert": (C.c_int, [C.POINTER(Volume), vp, vp, vp, vp, i64, vp, sz, vp]),
        "bnv_volume_query": (C.c_int, [C.POINTER(Volume), vp, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp]),
        "bnv_volume_count_optim": (C.c_int, [C.POINTER(Volume), vp, i64, vp, i64, vp, i32, vp]),
        "bnv_decode_pts": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, vp, i64, vp, vp, i64, C.c_int,
                                     C.POINTER(SdfDelta), vp, vp]),
        "bnv_sdfmlp_bwd_pack_floats": (sz, []),
        "bnv_sdfmlp_tcnn_bwd_pack_floats": (sz, []),
        "bnv_decode_pts_backward": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, vp, i64, vp, vp, vp, i64,
                                              C.c_int, vp, vp, vp]),
        "bnv_png_unfilter": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
        "bnv_ray_samples": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), vp, vp,
                                      C.c_int, C.c_int, C.c_int, C.c_float, vp, vp, vp, vp]),
        "bnv_ray_loss": (C.c_int, [vp, vp, vp, vp, i64, vp, vp, vp]),
        "bnv_ray_loss_splits": (C.c_int, [vp, vp, vp, vp, i64, i64, vp, vp, vp]),
        "bnv_volume_count_optim_splits": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, i64, C.c_int, i64, i64, vp, vp]),
        "bnv_volume_apply_split_counts": (C.c_int, [vp, vp, i64, vp]),
        "bnv_decode_pts_splits": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, vp, i64, vp, vp, i64, C.c_int,
                                            C.POINTER(SdfDelta), vp, i64, vp, vp]),
        "bnv_decode_pts_backward_splits": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, vp, i64, vp, vp, vp, i64,
                                                     C.c_int, vp, i64, vp, vp, vp]),
        "bnv_optim_step": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, vp, i64, vp, vp, vp, i64, C.c_int,
                                     C.POINTER(SdfDelta), vp, i64, vp, vp, vp, vp, vp, vp, vp]),
        "bnv_volume_count_optim_pts": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, i64, C.c_int, vp, i64, vp, i32,
                                                 vp]),
        "bnv_mc_count": (C.c_int, [vp, i64, vp, C.c_float, vp, vp, vp]),
        "bnv_mc_emit": (C.c_int, [vp, vp, i64, vp, C.c_float, C.c_float, C.POINTER(C.c_float), vp, vp, vp, vp]),
        "bnv_mc_count_indexed": (C.c_int, [vp, i64, vp, C.c_float, vp, vp, vp, vp]),
        "bnv_mc_emit_indexed": (C.c_int, [vp, vp, i64, vp, C.c_float, C.c_float, C.POINTER(C.c_float), vp, vp, vp, vp, vp,
                                          vp]),
        "bnv_decode_lattice_workspace_bytes": (sz, [i64, i64]),
        "bnv_encode_finish_image_parts": (C.c_int, [vp, i64, C.c_int, C.POINTER(Grid), vp, vp, sz, i64, vp, vp, vp, vp, i64,
                                                    C.c_int, vp, C.c_int, C.c_int, vp]),
        "bnv_decode_lattice_count_offset": (sz, [i64]),
        "bnv_decode_lattice_table_offset": (sz, [i64]),
        "bnv_decode_lattice_list_offset": (sz, [i64, i64]),
        "bnv_lattice_neighbors": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, i64, vp, i64, vp, vp, C.c_int, vp,
                                            sz, i32, vp]),
        "bnv_lattice_mark": (C.c_int, [C.POINTER(Volume), i64, vp, vp, sz, i32, vp]),
        "bnv_lattice_table": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, vp, i64, C.c_int, vp, sz, vp]),
        "bnv_lattice_blend": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, i64, vp, C.POINTER(SdfDelta), vp, sz,
                                        vp, vp]),
        "bnv_depth_workspace_bytes": (sz, [C.c_int, C.c_int]),
        "bnv_depth_to_points": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.c_double, vp, sz, vp, vp, vp]),
        "bnv_depth_to_points_padded": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.c_double, vp, sz, vp, vp, vp]),
        "bnv_tsdf_integrate": (C.c_int, [vp, vp, vp, C.POINTER(i32), C.POINTER(C.c_float), C.c_float, C.c_float, vp, vp,
                                         C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float,
                                         vp, vp]),
        "bnv_tsdf_integrate_u16": (C.c_int, [vp, vp, vp, C.POINTER(i32), C.POINTER(C.c_float), C.c_float, C.c_float, vp, vp,
                                         C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float,
                                         vp, vp]),
        "bnv_tsdf_integrate_batch_u16": (C.c_int, [vp, vp, vp, C.POINTER(i32), C.POINTER(C.c_float), C.c_float, C.c_float,
                                                   C.c_int, vp, vp, C.c_int, C.c_int, C.POINTER(C.c_float),
                                                   C.POINTER(C.c_float), C.c_float, C.c_float, vp]),
        "bnv_set_mlp_mode": (C.c_int, [C.c_int]),
        "bnv_get_mlp_mode": (C.c_int, []),
        "bnv_set_option": (C.c_int, [C.c_char_p, C.c_int]),
        "bnv_profile_enable": (C.c_int, [C.c_int]),
        "bnv_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(i64)]),
        "bnv_probe_mfma_rate": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "bnv_probe_spin": (C.c_int, [C.c_int, i64, vp]),
        "bnv_decode_lattice": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, vp, i64, vp, vp, i64, vp,
                                         C.POINTER(SdfDelta), vp, sz, i32, vp, vp]),
        "bnv_decode_dense": (C.c_int, [vp, vp, C.POINTER(i32), C.c_float, i32, vp, vp, i64, i32, vp, vp, vp, vp]),
        "bnv_decode_lattice_stamped_tables": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, vp, i64, vp, vp, i64, vp,
                                                        vp, sz, i32, vp]),
        "bnv_encode_finish_image_wg": (C.c_int, [vp, i64, C.c_int, C.POINTER(Grid), vp, vp, sz, i64, vp, vp, vp, vp, i64,
                                                 C.c_int, vp, C.c_int, vp]),
        "bnv_shard_state_bytes": (sz, [C.POINTER(i32), i32]),
        "bnv_shard_state_loads_offset": (sz, []),
        "bnv_shard_state_table_offset": (sz, []),
        "bnv_decode_dense_mode": (C.c_int, [vp, vp, C.POINTER(i32), C.c_float, i32, vp, vp, i64, i32, i32, vp, vp, vp, vp]),
        "bnv_frame_pipe_set_mlp_mode": (C.c_int, [vp, i32]),
        "bnv_shard_install_reset": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, C.c_int, i64, vp, vp]),
        "bnv_volume_integrate_frame": (C.c_int, [C.POINTER(Volume), vp, vp, vp, i64, vp, vp, sz,
                                                 C.POINTER(IntegrateExtras), vp]),
        "bnv_readback_words": (C.c_int, [vp, vp, vp, vp]),
        "bnv_decode_lattice_stamped": (C.c_int, [C.POINTER(Volume), C.POINTER(Grid), vp, vp, i64, vp, vp, i64, vp,
                                                 C.POINTER(SdfDelta), vp, sz, i32, vp, vp]),
        "bnv_frame_pipe_create": (C.c_int, [C.POINTER(FramePipeConfig), C.POINTER(vp)]),
        "bnv_frame_pipe_destroy": (C.c_int, [vp]),
        "bnv_frame_begin_depth": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                            C.POINTER(C.c_double), vp]),
        "bnv_frame_begin_points": (C.c_int, [vp, C.c_int, vp, i64]),
        "bnv_frame_upsert": (C.c_int, [vp, C.c_int, C.POINTER(Volume), vp, sz, vp, i32]),
        "bnv_frame_bound": (C.c_int, [vp, C.c_int, C.POINTER(i32)]),
        "bnv_frame_finish": (C.c_int, [vp, C.c_int, C.POINTER(Volume), vp, i64, vp, C.POINTER(SdfDelta), vp, sz, i32]),
        "bnv_frame_result": (C.c_int, [vp, C.c_int, C.POINTER(i32)]),
        "bnv_frame_ready": (C.c_int, [vp, C.c_int]),
        "bnv_frame_pipe_timeline_enable": (C.c_int, [vp, C.c_int]),
        "bnv_frame_timeline": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float)]),
        "bnv_frame_side_depth": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), vp]),
        "bnv_frame_cancel": (C.c_int, [vp, C.c_int]),
        "bnv_frame_pipe_forget_workspaces": (C.c_int, [vp]),
        "bnv_shard_state_configure": (C.c_int, [vp, i32, i32, vp]),
    }
    for name in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype, fn.argtypes = sig[name]
    _lib = lib
    # A/B switches from the environment (tools, experiments): BNV_OPTIONS="name=value,name=value" -> bnv_set_option.
    # They choose between implementations with identical results.
    for item in filter(None, os.environ.get("BNV_OPTIONS", "").split(",")):
        name, _, value = item.partition("=")
        rc = lib.bnv_set_option(name.strip().encode(), int(value))
        if rc != 0:
            raise BnvError(f"BNV_OPTIONS: bnv_set_option({name.strip()!r}, {value}) failed ({rc})")
    return lib


def check(status, what):
    if status != 0:
        lib = load()
        msg = lib.bnv_status_string(status).decode()
        raise BnvError(f"{what} failed: {msg} (status {status}, hip error {lib.bnv_last_hip_error()})")


def require_device(device_index):
    """bnv_init on first use; raises if no MI355X-class device / library is available."""
    global _initialised_device
    lib = load()
    if _initialised_device != device_index:
        check(lib.bnv_init(int(device_index)), "bnv_init")
        _initialised_device = device_index
    return lib


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
