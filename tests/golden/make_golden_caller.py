"""Golden TRACE of the reference's own caller: src/run_e2e.py's ``NeuralMap`` (run_e2e.py:27-194) run on CPU in this
container with recording proxies around the three objects it drives -- the point-net model (``pointnet``), the
``SparseVolume`` and the TSDF volume -- so that the exact call surface that class touches is pinned by a test instead
of by reading: which methods, in which order, with which argument names / dtypes / shapes, which attributes are read
and written (``volume.features = nn.Parameter(...)``, run_e2e.py:114), what comes back (arity, dtypes, shapes, values).

Build-container only (needs /root/reference):  python tests/golden/make_golden_caller.py
Scenario (64^3 volume, voxel 0.02): ``NeuralMap.__init__`` -> ``integrate`` x 10 frames (one of them without a point
inside the volume: the reference returns before _integrate) -> ``extract_mesh`` (``volume.meshlize`` up to marching
cubes, which scikit-image would run: stubbed, its INPUT lattices are recorded) -> ``optimize`` (2 iterations of 2 ray
splits; the DataLoader is replaced by two prepared ray batches -- the dataset needs cv2 and image files) ->
``extract_mesh`` -> ``save``.  Only DATA is written (tests/golden/caller_64.npz): a JSON event list + the arrays it
refers to.  tests/test_gpu_caller_trace.py replays the depth-0 events against bnv_fusion_amd's classes on the GPU.

Event kinds (``depth`` = number of recorded calls on the stack; the replay drives depth 0 -- what the CALLER does --
and ignores what the reference's own methods do among themselves):
  new   an object of the boundary is constructed            {cls, args, kwargs}
  get   attribute read                                      {obj, attr, value}
  set   attribute write                                     {obj, attr, value}
  call  method call                                         {obj, method, args{name: value}, ret | raised}
  grad  a gradient reached a tensor the caller made a leaf  {obj, value}   (autograd edge of decode_pts)
  mc    marching cubes was asked for (meshlize)             {n_calls, origins, lattices}
Values: {t: none|py|ndarray|tensor|tuple|dict|obj|opaque, ...}; arrays by key into the npz; tensors carry ``ref``
(identity across events), ``param``, ``requires_grad`` and ``role`` ("device": the reference put it on its CUDA device).
"""
import hashlib
import inspect
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import ref_shims  # noqa: E402
from make_golden import surface_points  # noqa: E402
from make_golden_grad import make_rays, surface_z  # noqa: E402

VOXEL = 0.02
DIMS = np.array([1.24, 1.24, 1.24])
N_FRAMES = 10
EMPTY_FRAME = 4            # this frame's points all lie outside the volume
N_PTS = 4000
IMG_H, IMG_W = 60, 80
N_RAYS, RAY_SPLIT = 240, 120
SAMPLE_LIMIT = 1 << 18     # arrays above this many elements are stored as SHA-256 + a strided sample


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------------------------------------------------------
# the recorder
# ------------------------------------------------------------------------------------------------------------------
class Recorder:
    def __init__(self):
        self.events, self.arrays = [], {}
        self.depth = 0
        self.refs = {}       # id(tensor) -> ref name
        self.last_sha = {}   # ref -> sha of the data at its last sighting
        self.keep = []       # keeps recorded tensors alive (ids stay unique)
        self.caller_owned = set()
        self.proxies = {}    # id(real object) -> name
        self.by_sha = {}     # (sha, dtype, shape) -> array key
        self.live = {}       # ref -> the tensor object (caller-owned ones are looked at before every depth-0 call)
        self.sha_ref = {}    # sha of a tensor's data -> the ref that carried it last

    def store(self, a):
        shape = list(np.asarray(a).shape)
        a = np.ascontiguousarray(a).reshape(shape)      # (ascontiguousarray alone turns 0-d arrays into 1-d ones)
        out = {"dtype": str(a.dtype), "shape": shape}
        if a.size > SAMPLE_LIMIT:
            stride = -(-a.size // 4096)
            out.update(sha=sha(a), stride=stride, sample=self._put(a.reshape(-1)[::stride]))
        else:
            out["data"] = self._put(a)
        return out

    def _put(self, a):
        # a COPY: tensor.numpy() shares the tensor's memory, and the caller goes on changing some tensors in place
        a = np.array(np.ascontiguousarray(a).reshape(np.asarray(a).shape), copy=True)
        h = (sha(a), str(a.dtype), a.shape)
        if h in self.by_sha:                      # the same bytes are stored once
            return self.by_sha[h]
        key = f"a{len(self.arrays)}"
        self.arrays[key] = a
        self.by_sha[h] = key
        return key

    def desc(self, v):
        if v is None:
            return {"t": "none"}
        if isinstance(v, Proxy):
            return {"t": "obj", "name": object.__getattribute__(v, "_n")}
        if id(v) in self.proxies:
            return {"t": "obj", "name": self.proxies[id(v)]}
        if isinstance(v, (bool, int, float, str)):
            return {"t": "py", "py": type(v).__name__, "v": v}
        if isinstance(v, (np.floating, np.integer)):
            return {"t": "py", "py": type(v).__name__, "v": v.item()}
        light = self.depth > 0       # what the reference's methods pass among themselves: structure only, no data
        if isinstance(v, np.ndarray):
            return {"t": "ndarray", "dtype": str(v.dtype), "shape": list(v.shape)} if light else \
                {"t": "ndarray", **self.store(v)}
        if isinstance(v, torch.Tensor) and light:
            return {"t": "tensor", "param": isinstance(v, torch.nn.Parameter), "requires_grad": bool(v.requires_grad),
                    "dtype": str(v.detach().cpu().numpy().dtype), "shape": list(v.shape)}
        if isinstance(v, torch.Tensor):
            a = v.detach().cpu().numpy()
            ref = self.refs.get(id(v))
            if ref is None:
                ref = f"t{len(self.refs)}"
                self.refs[id(v)] = ref
                self.keep.append(v)
                self.live[ref] = v
            d = {"t": "tensor", "ref": ref, "param": isinstance(v, torch.nn.Parameter),
                 "requires_grad": bool(v.requires_grad), "role": "device" if getattr(v, "_was_cuda", True) else "host"}
            h = sha(a)
            if self.last_sha.get(ref) == h:
                d.update(dtype=str(a.dtype), shape=list(a.shape), same=True)        # unchanged since its last sighting
            else:
                d.update(self.store(a), changed=ref in self.last_sha,
                         caller_owned=ref in self.caller_owned)
                if ref not in self.last_sha and self.sha_ref.get(h, ref) != ref:
                    d["alias_of"] = self.sha_ref[h]      # a new object with the very data of a tensor seen before
                self.last_sha[ref] = h
            self.sha_ref[h] = ref
            return d
        if isinstance(v, (tuple, list)):
            return {"t": "tuple" if isinstance(v, tuple) else "list", "items": [self.desc(x) for x in v]}
        if isinstance(v, dict):
            return {"t": "dict", "items": {str(k): self.desc(x) for k, x in v.items()}}
        return {"t": "opaque", "cls": type(v).__name__}

    def event(self, **kw):
        kw["depth"] = self.depth
        self.events.append(kw)
        return kw


class Proxy:
    """Forwards everything to the wrapped reference object and records it."""

    def __init__(self, rec, obj, name):
        object.__setattr__(self, "_r", rec)
        object.__setattr__(self, "_o", obj)
        object.__setattr__(self, "_n", name)
        object.__setattr__(self, "_sub", {})
        rec.proxies[id(obj)] = name

    def __getattr__(self, k):
        rec, obj, name = (object.__getattribute__(self, a) for a in ("_r", "_o", "_n"))
        v = getattr(obj, k)
        if isinstance(v, torch.nn.Module):                      # pointnet.nerf: a boundary object of its own
            sub = object.__getattribute__(self, "_sub")
            if k not in sub:
                sub[k] = Proxy(rec, v, f"{name}.{k}")
            rec.event(op="get", obj=name, attr=k, value={"t": "obj", "name": f"{name}.{k}"})
            return sub[k]
        if callable(v) and not isinstance(v, torch.Tensor):
            return self._wrap(k, v)
        rec.event(op="get", obj=name, attr=k, value=rec.desc(v))
        return v

    def __setattr__(self, k, v):
        rec, obj, name = (object.__getattribute__(self, a) for a in ("_r", "_o", "_n"))
        d = rec.desc(v)
        if isinstance(v, torch.Tensor):
            rec.caller_owned.add(d["ref"])
            if v.requires_grad:      # the caller made it a leaf (run_e2e.py:114): record what autograd delivers to it
                def hook(g, n=f"{name}.{k}"):
                    rec.event(op="grad", obj=n, value=rec.desc(g.clone()))
                v.register_hook(hook)
        rec.event(op="set", obj=name, attr=k, value=d)
        setattr(obj, k, v)

    def _wrap(self, k, fn):
        rec, name = object.__getattribute__(self, "_r"), object.__getattribute__(self, "_n")

        def call(*a, **kw):
            try:
                bound = inspect.signature(fn).bind(*a, **kw)
                named = dict(bound.arguments)
            except (TypeError, ValueError):
                named = {f"arg{i}": x for i, x in enumerate(a)}
                named.update(kw)
            # what the caller's own code did, in place, to tensors it owns since the boundary last saw them (the
            # optimiser's step on volume.features happens between two calls without touching the boundary)
            state = {}
            if rec.depth == 0:
                for ref in sorted(rec.caller_owned):
                    d = rec.desc(rec.live[ref])
                    if "same" not in d:
                        state[ref] = d
            ev = rec.event(op="call", obj=name, method=k, args={n: rec.desc(x) for n, x in named.items()},
                           positional=len(a), keywords=sorted(kw), caller_state=state)
            rec.depth += 1
            try:
                ret = fn(*a, **kw)
            except BaseException as e:
                ev["raised"] = type(e).__name__
                raise
            finally:
                rec.depth -= 1
            ev["ret"] = rec.desc(ret)
            if isinstance(ret, torch.Tensor) and ret.requires_grad:
                ret.register_hook(lambda g, ev=ev: ev.__setitem__("grad_in", rec.store(g.detach().cpu().numpy())))
            return ret

        return call


# ------------------------------------------------------------------------------------------------------------------
# the scenario
# ------------------------------------------------------------------------------------------------------------------
def camera():
    T_wc = torch.eye(4)
    T_wc[:3, :3] = torch.tensor([[1.0, 0, 0], [0, -1.0, 0], [0, 0, -1.0]])
    T_wc[:3, 3] = torch.tensor([0.01, -0.02, 0.45])
    intr = torch.tensor([[100.0, 0, 40.0], [0, 100.0, 30.0], [0, 0, 1.0]])
    return T_wc, intr


def depth_image(shift):
    """z-depth of the analytic surface z = surface_z(x, y, shift) seen from camera(): fixed-point iteration."""
    T_wc, intr = camera()
    v, u = torch.meshgrid(torch.arange(IMG_H, dtype=torch.float32), torch.arange(IMG_W, dtype=torch.float32),
                          indexing="ij")
    dx, dy = (u - intr[0, 2]) / intr[0, 0], (v - intr[1, 2]) / intr[1, 1]
    depth = torch.full((IMG_H, IMG_W), 0.45)
    for _ in range(30):
        pc = torch.stack([dx * depth, dy * depth, depth], -1)
        pw = pc @ T_wc[:3, :3].T + T_wc[:3, 3]
        depth = 0.45 - surface_z(pw[..., 0], pw[..., 1], shift)
    return depth.numpy().astype(np.float32)


def make_frame(t):
    """What the reference's DataLoader hands NeuralMap.integrate (fusion_inference_dataset.py:40-90, batch dim added
    by the default collate; run_e2e.py:247-249 moves the tensors to the device as float32)."""
    shift = 0.02 * t
    pts = surface_points(N_PTS, 100 + t, VOXEL, 0.2, shift=shift)[None]
    if t == EMPTY_FRAME:
        pts = pts.clone()
        pts[..., :3] += 100.0
    T_wc, intr = camera()
    d = depth_image(shift)
    rgb = np.full((3, IMG_H, IMG_W), 0.1, np.float32)          # (rgb - 0.5 normalised values; any constant will do)
    rgbd = np.concatenate([rgb, d[None]], 0)[None]
    return {"input_pts": pts.float(), "rgbd": torch.from_numpy(rgbd), "intr_mat": intr[None].clone(),
            "T_wc": T_wc[None].clone()}


def make_cfg():
    cfg = ref_shims.make_cfg(VOXEL)
    cfg["model"].update(train_ray_splits=RAY_SPLIT, sdf_delta_weight=0.1,
                        ray_tracer=dict(ray_max_dist=3, truncated_units=10))
    cfg["dataset"] = dict(scan_id="scene3d/caller", num_pixels=N_RAYS, skip_images=1, confidence_level=0)
    return ref_shims.AttrDict(cfg)


def main():
    torch.set_num_threads(8)
    torch.manual_seed(0)
    np.random.seed(0)
    ref_shims.install()
    ref_shims.install_run_e2e()
    work = "/tmp/refwork_caller"
    os.makedirs(work, exist_ok=True)
    os.chdir(work)
    import src.run_e2e as R
    import src.models.sparse_volume as SVM
    import third_parties.fusion as TF

    # ---- the reference runs on "cuda"; here that is the CPU: every request for a CUDA device lands on it
    orig_to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) or
                  (isinstance(x, torch.device) and x.type == "cuda") else x for x in a)
        if "device" in k and "cuda" in str(k["device"]):
            k["device"] = "cpu"
        return orig_to(self, *a, **k)

    torch.Tensor.to = to
    torch.Tensor.cuda = lambda self, *a, **k: self

    rec = Recorder()
    cfg = make_cfg()
    sd = ref_shims.load_checkpoint_state_dict(os.path.join(ref_shims.REFERENCE_ROOT, "pretrained", "pointnet.ckpt"))
    model = R.LitFusionPointNet(cfg)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    model.eval()
    model.freeze()
    pointnet = Proxy(rec, model, "pointnet")

    # ---- constructors of the boundary objects, as NeuralMap.__init__ calls them
    real_sv, real_tsdf = R.SparseVolume, TF.TSDFVolume

    def new_volume(*a, **k):
        rec.event(op="new", cls="SparseVolume", args=[rec.desc(x) for x in a], kwargs={n: rec.desc(x) for n, x in k.items()})
        return Proxy(rec, real_sv(*a, **{**k, "device": "cpu"}), "volume")

    def new_tsdf(*a, **k):
        rec.event(op="new", cls="TSDFVolume", args=[rec.desc(x) for x in a], kwargs={n: rec.desc(x) for n, x in k.items()})
        return Proxy(rec, real_tsdf(*a, **k), "tsdf_vol")

    R.SparseVolume = new_volume
    TF.TSDFVolume = new_tsdf                     # (run_e2e refers to it as fusion.TSDFVolume)

    # ---- meshlize: scikit-image's marching cubes is absent; record what it is asked for and hand back one triangle
    mc = {"origins": [], "lattices": []}

    def marching_cubes(vol, level=0.0, spacing=(1.0, 1.0, 1.0), **k):
        mc["lattices"].append(np.asarray(vol, np.float32).copy())
        v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float64) * np.asarray(spacing)
        return v, np.array([[0, 1, 2]]), None, None

    class Trimesh:
        def __init__(self, vertices=None, faces=None, process=True, **k):
            self.vertices, self.faces = vertices, faces

        def export(self, path):
            pass

    SVM.marching_cubes = marching_cubes
    SVM.trimesh.Trimesh = Trimesh

    # ---- optimize(): the DataLoader over IterableInferenceDataset needs cv2 + image files; two prepared batches
    ray_batches = [make_rays(N_RAYS, 11), make_rays(N_RAYS, 12)]

    def loader(dataset, **k):
        assert k.get("batch_size", 1) is None and dataset.n_iters == len(ray_batches), (k, dataset.n_iters)
        return [{n: v.clone() for n, v in b.items()} for b in ray_batches]

    R.torch.utils.data.DataLoader = loader
    R.tqdm = lambda x: x

    nm = R.NeuralMap(DIMS, cfg, pointnet, work)
    frames = [make_frame(t) for t in range(N_FRAMES)]
    marks = {}
    for t, f in enumerate(frames):
        marks[f"integrate_{t}"] = len(rec.events)
        nm.integrate(f)
    marks["extract_mesh_0"] = len(rec.events)

    def mesh_pass(tag):
        n0 = len(mc["lattices"])
        mesh = nm.extract_mesh()
        lat = np.stack(mc["lattices"][n0:]) if len(mc["lattices"]) > n0 else np.zeros((0, 3, 3, 3), np.float32)
        rec.event(op="mc", tag=tag, n_calls=int(len(lat)), lattices=rec.store(lat),
                  mesh_vertices=int(len(mesh.vertices)) if mesh is not None else 0)

    nm.volume.to_tensor()
    mesh_pass("before_optim")
    marks["optimize"] = len(rec.events)
    nm.optimize(n_iters=len(ray_batches), last_frame=-1)
    marks["extract_mesh_1"] = len(rec.events)
    mesh_pass("after_optim")
    marks["save"] = len(rec.events)
    nm.save()
    saved = torch.load(os.path.join(work, "final_sparse_volume.pth"), weights_only=False)
    tsdf_saved = np.load(os.path.join(work, nm.scan_id + ".npy"))
    rec.event(op="saved", files={"final_sparse_volume.pth": rec.desc({k: v for k, v in saved.items()}),
                                 nm.scan_id + ".npy": rec.desc(tsdf_saved)})
    marks["end"] = len(rec.events)

    meta = {"voxel_size": VOXEL, "dims": DIMS.tolist(), "n_frames": N_FRAMES, "empty_frame": EMPTY_FRAME,
            "n_pts": N_PTS, "img_hw": [IMG_H, IMG_W], "n_rays": N_RAYS, "ray_split": RAY_SPLIT, "marks": marks,
            "min_pts_in_grid": 8, "scan_id": nm.scan_id}
    out = dict(rec.arrays)
    out["events_json"] = np.frombuffer(json.dumps({"meta": meta, "events": rec.events}).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "caller_64.npz")
    np.savez_compressed(path, **out)
    d0 = [e for e in rec.events if e["depth"] == 0]
    print(f"{len(rec.events)} events ({len(d0)} at depth 0), {len(rec.arrays)} arrays, "
          f"{os.path.getsize(path) / 1e6:.2f} MB -> {path}")
    from collections import Counter
    print(Counter((e["op"], e.get("obj"), e.get("method") or e.get("attr")) for e in d0).most_common(60))


if __name__ == "__main__":
    main()
