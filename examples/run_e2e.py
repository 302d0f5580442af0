"""The reference's end-to-end loop (src/run_e2e.py:196-293) on this package: local fusion of every frame of a
sequence directory, periodic global optimisation, mesh extraction, final artefacts.

    python examples/run_e2e.py --data-dir DATA --scan-id scene3d/lounge --out OUT          # the reference's layout
    python examples/run_e2e.py --synthetic 24 --out /tmp/bnv_demo                          # writes a synthetic scene first
    python examples/run_e2e.py --sweep 600 --grid 512 --decode-frames --pipelined --no-optimize --out /tmp/sweep
                                                    # a moving-camera room sweep (bnv_fusion_amd/sequence.py), per-frame
                                                    # SDF decode of the touched voxels, two frames in flight

Frames are read from ``<data-dir>/<scan-id>/{depth/<i>.png, pose/T_wc_<i>.txt, pose/intr_mat_<i>.txt,
pose/dimensions.txt}`` (bnv_fusion_amd/datasets.py), the volume extent comes from ``dimensions.txt`` exactly as in
the reference; checkpoints default to the converted weights shipped with the package.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bnv_fusion_amd as bnv                                   # noqa: E402
bnv.configure_runtime()                                       # optional: 8 hardware queues for the pipelined streams
from bnv_fusion_amd import datasets, synthetic                # noqa: E402
from bnv_fusion_amd.mesh import post_process_mesh             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data-dir")
    ap.add_argument("--scan-id", default="synthetic/scene0")
    ap.add_argument("--synthetic", type=int, default=0, help="write this many synthetic 640x480 frames and use them")
    ap.add_argument("--out", required=True)
    ap.add_argument("--voxel-size", type=float, default=0.01)
    ap.add_argument("--tiny-cuda", action="store_true", help="the reference's default tiny-cuda-nn checkpoint")
    ap.add_argument("--skip-images", type=int, default=1)
    ap.add_argument("--optim-interval", type=int, default=100)        # fusion_pointnet_model.yaml:48
    ap.add_argument("--mode", default="offline", choices=["demo", "offline"])
    ap.add_argument("--no-optimize", action="store_true")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--sweep", type=int, default=0, help="write this many frames of the moving-camera room sweep "
                                                          "(bnv_fusion_amd/sequence.py) and use them")
    ap.add_argument("--grid", type=int, default=512, choices=[256, 512], help="--sweep: volume of the sweep")
    ap.add_argument("--decode-frames", action="store_true",
                    help="decode the SDF lattice of every frame's touched voxels (the unit of the benchmark metric)")
    ap.add_argument("--pipelined", action="store_true",
                    help="with --decode-frames: two frames in flight (fuse_and_decode_async) instead of one "
                         "synchronous call per frame")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    dev = "cuda:0"

    if args.synthetic:
        args.data_dir = args.data_dir or os.path.join(args.out, "data")
        H, W = args.height, args.width
        dims = {0.01: 2.54, 0.02: 2.52}.get(args.voxel_size, 2.54)
        datasets.write_sequence(args.data_dir, args.scan_id,
                                [synthetic.depth_u16(t, H, W) for t in range(args.synthetic)],
                                synthetic.intrinsics(H, W), [synthetic.pose(t) for t in range(args.synthetic)],
                                [dims] * 3)
    if args.sweep:
        from bnv_fusion_amd import sequence
        args.data_dir = args.data_dir or os.path.join(args.out, "data")
        args.scan_id = "sweep/room"
        dims_m, args.voxel_size, scale = sequence.DIMS[args.grid]
        datasets.write_sequence(args.data_dir, args.scan_id,
                                (sequence.depth_u16(t, scale=scale, device=dev).cpu().numpy() for t in range(args.sweep)),
                                sequence.intrinsics(), (sequence.sweep_pose(t, scale) for t in range(args.sweep)),
                                [dims_m] * 3, filter_type=0, level=1)
    data = datasets.FusionInferenceDataset(args.data_dir, args.scan_id, skip_images=args.skip_images, device=dev)
    model = bnv.load_pretrained(device=dev, voxel_size=args.voxel_size, tiny_cuda=args.tiny_cuda)
    nm = bnv.NeuralMap(data.dimensions, args.voxel_size, model, capacity=1 << 20, device=dev, tsdf=True,
                       max_depth=data.max_depth)
    t_local = t_global = 0.0
    max_depth = data.max_depth
    if args.decode_frames:
        # the per-frame loop of the benchmark metric: fuse + decode of the touched voxels, synchronous or pipelined
        from bnv_fusion_amd import sequence
        nm.volume.reset(100000)                  # the reference's initial capacity: the tables grow on demand
        st = sequence.run(nm, data, pipelined=args.pipelined, in_flight=2, checksums=False,
                          on_frame=lambda k, fr, c, s: nm.frames.append(fr))
        print(f"fused + decoded {st['frames']} frames ({st['empty_frames']} without a point inside the volume) at "
              f"{st['frames'] / st['seconds']:.1f} frames/s incl. file reading; {nm.volume.num_rows()} voxels")
        data = []
        t_local = st["seconds"]
    for idx, frame in enumerate(data):                                   # run_e2e.py:243-279
        t0 = time.perf_counter()
        nm.integrate(frame)
        torch.cuda.synchronize()
        t_local += time.perf_counter() - t0
        if np.isnan(frame["T_wc"]).any():
            continue
        nm.frames.append(frame)
        if args.mode == "demo" and not args.no_optimize and idx % args.optim_interval == 0:
            last = max(0, len(nm.frames) - args.optim_interval)
            n_iters = min(len(nm.frames), args.optim_interval) * args.skip_images
            t0 = time.perf_counter()
            nm.optimize(n_iters=n_iters, last_frame=last, ray_max_dist=max_depth)
            torch.cuda.synchronize()
            t_global += time.perf_counter() - t0
            mesh = nm.extract_mesh()
            if mesh is not None:                                             # :277-280
                post_process_mesh(mesh).export(os.path.join(args.out, f"{idx}.ply"))
    mesh = nm.extract_mesh(os.path.join(args.out, "before_optim.ply"))   # :280-282
    steps = int(len(nm.frames) * args.skip_images) * (1 if args.mode == "demo" else 2)   # :283-284
    if not args.no_optimize:
        t0 = time.perf_counter()
        nm.optimize(n_iters=steps, last_frame=-1, ray_max_dist=max_depth)
        torch.cuda.synchronize()
        t_global += time.perf_counter() - t0
    print(f"speed on local fusion: {len(nm.frames) / max(t_local, 1e-9):.1f} fps"
          + ("" if args.no_optimize else f"; speed on global fusion: {steps / max(t_global, 1e-9):.1f} fps"))
    mesh = nm.extract_mesh()                                             # :291-294
    if mesh is not None:
        mesh = post_process_mesh(mesh, vertex_threshold=nm.voxel_size / 4)
        mesh.export(os.path.join(args.out, "final.ply"))
    nm.save(args.out, scan_id=args.scan_id.split("/")[-1])
    print(f"{len(nm.frames)} frames, {nm.volume.num_rows()} voxels, "
          f"{0 if mesh is None else len(mesh.faces)} triangles -> {args.out}")


if __name__ == "__main__":
    main()
