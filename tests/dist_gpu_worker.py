"""One rank of the multi-process GPU tests (tests/test_gpu_multiprocess.py): started by torch.distributed.run, several
ranks SHARE one GPU (BNV_DIST_BACKEND=gloo; RCCL needs a GPU per rank), runs a few synthetic frames through one of
the two multi-GPU modes of bnv_fusion_amd.distributed and saves this rank's per-frame outputs."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["spatial", "frame"], required=True)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--checkpoint", default="fp32", choices=["fp32", "tcnn"])
    ap.add_argument("--ownership", default=None, choices=["hash", "first_touch", "region"])
    ap.add_argument("--ahead", action="store_true", help="spatial: announce every next frame (next_frame=), and at the "
                    "end a frame that never comes: abandon() must free its slot")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import bnv_fusion_amd
    bnv_fusion_amd.configure_runtime()
    torch.cuda.set_device(0)
    dist.init_process_group(os.environ.get("BNV_DIST_BACKEND", "gloo"))
    import bnv_fusion_amd as bnv
    from bnv_fusion_amd import synthetic
    from bnv_fusion_amd.distributed import FrameParallelNeuralMap, ShardedNeuralMap
    dims, voxel = synthetic.GRID_DIMS[args.grid]
    model = bnv.load_pretrained(device="cuda:0", voxel_size=voxel, tiny_cuda=args.checkpoint == "tcnn")
    H, W = args.height, args.width
    frames = [{"depth": torch.from_numpy(synthetic.depth_u16(t, H, W)).cuda(), "intr_mat": synthetic.intrinsics(H, W),
               "T_wc": synthetic.pose(t)} for t in range(args.frames)]
    out = {}
    if args.mode == "spatial":
        nm = ShardedNeuralMap(np.array([dims] * 3), voxel, model, device="cuda:0", tsdf=True, ownership=args.ownership)
        pending = None
        evals = {}
        for t, fr in enumerate(frames):                 # pipelined: frame t is enqueued before t-1 is collected
            if args.ahead:
                # the next frame's encode is enqueued before this frame's bound is waited for; the last frame announces
                # one that is never passed
                h = nm.fuse_and_decode_async(fr, next_frame=frames[t + 1] if t + 1 < len(frames) else frames[0])
                if t + 1 == len(frames):
                    try:
                        nm.fuse_and_decode_async(frames[1])        # not the announced frame: refused, nothing changes
                        raise SystemExit("a frame other than the announced one was accepted")
                    except bnv.BnvError:
                        pass
                    nm.abandon()
            else:
                h = nm.fuse_and_decode_async(fr)
            if pending is not None:
                c, s = pending[1].result()
                evals[pending[0]] = nm.backend._last_evals
                out[pending[0]] = (None if c is None else c.cpu(), None if s is None else s.cpu())
            pending = (t, h)
        c, s = pending[1].result()
        evals[pending[0]] = nm.backend._last_evals
        out[pending[0]] = (None if c is None else c.cpu(), None if s is None else s.cpu())
        assert nm.flush() == [] and nm._pre is None and not any(nm.backend.pipe._busy)
        table, loads = nm.backend.owner_table()
        meta = {"host_waits": nm.host_waits, "exchanged_bytes": nm.exchanged_bytes, "rows": nm.volume.num_rows(),
                "tsdf": nm.backend.tsdf_vol.tsdf.cpu(), "ownership": nm.backend.ownership, "owner_table": table,
                "owner_loads": loads, "mlp_evals": evals, "block_log2": nm.backend.block_log2, "axis": nm.backend.axis}
    else:
        nm = FrameParallelNeuralMap(np.array([dims] * 3), voxel, model, device="cuda:0", tsdf=True)
        batches = [frames[b0: b0 + world] for b0 in range(0, len(frames), world)]
        for k, handle in enumerate(nm.process_stream(batches)):
            c, s = handle.result()
            if c is not None:
                out[k * world + rank] = (c.cpu(), s.cpu())
        nm.flush()
        meta = {"exchanged_bytes": nm.exchanged_bytes, "rows": nm.volume.num_rows(),
                "tsdf": nm.backend.tsdf_vol.tsdf.cpu()}
    torch.save({"out": out, "meta": meta}, os.path.join(args.out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
