"""The on-disk sequence layout the reference's ``fusion_inference_dataset`` reads
(src/datasets/fusion_inference_dataset.py:105-146; SURVEY.md section 8 f: data formats on the input side):

    <data_dir>/<scan_id>/depth/<i>.png          16-bit greyscale, millimetres (cv2.imread(path, -1) / 1000., common.py:93)
    <data_dir>/<scan_id>/pose/T_wc_<i>.txt      16 numbers on one line: camera-to-world, row-major
    <data_dir>/<scan_id>/pose/intr_mat_<i>.txt  9 (or 16) numbers on one line
    <data_dir>/<scan_id>/pose/dimensions.txt    3 numbers: the metric extent of the volume
    <data_dir>/<scan_id>/image/<i>.jpg          colour (only counted, never decoded on this path)

``FusionInferenceDataset`` yields the frame dicts ``NeuralMap.integrate`` / ``fuse_and_decode_async`` take
(``depth`` as a uint16 tensor on the device: the GPU front end replaces the dataset's numpy unprojection).
OpenCV is not a dependency: the PNG container is parsed here (chunks + zlib), the scanline filters are reversed by
``bnv_png_unfilter`` in the shared library.  ``write_sequence`` produces the same layout (used by the tests and
``examples/run_e2e.py`` for a synthetic scene).
"""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import torch

from . import _lib

_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def read_png16(path):
    """16-bit (or 8-bit) greyscale PNG -> numpy [H, W] uint16, like cv2.imread(path, -1)."""
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:8] != _PNG_SIG:
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos: pos + 8])
        body = data[pos + 8: pos + 8 + n]
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
        pos += 12 + n
    if hdr is None:
        raise ValueError(f"{path}: no IHDR chunk")
    w, h, depth, colour, _, _, interlace = hdr
    if colour != 0 or depth not in (8, 16) or interlace != 0:
        raise ValueError(f"{path}: expected a non-interlaced 8/16-bit greyscale PNG (colour type {colour}, "
                         f"bit depth {depth}, interlace {interlace})")
    bpp = depth // 8
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
    row = w * bpp
    if raw.size != h * (row + 1):
        raise ValueError(f"{path}: truncated image data")
    out = np.empty(h * row, dtype=np.uint8)
    lib = _lib.load()
    _lib.check(lib.bnv_png_unfilter(raw.ctypes.data_as(C.c_void_p), h, row, bpp, out.ctypes.data_as(C.c_void_p)),
               "bnv_png_unfilter")
    if bpp == 2:
        return out.view(">u2").reshape(h, w).astype(np.uint16)
    return out.reshape(h, w).astype(np.uint16)


def write_png16(path, image, filter_type=0, level=6):
    """numpy [H, W] uint16 -> 16-bit greyscale PNG.  ``filter_type`` 0..4 selects the scanline filter of every
    row (the tests write all five); ``level``: zlib compression level."""
    img = np.ascontiguousarray(np.asarray(image, dtype=np.uint16))
    h, w = img.shape
    if filter_type == 0:      # (fast path: no predictor arithmetic)
        raw = np.concatenate([np.zeros((h, 1), np.uint8), img.astype(">u2").view(np.uint8).reshape(h, 2 * w)],
                             axis=1).tobytes()
        return _write_png(path, w, h, raw, level)
    rows = img.astype(">u2").view(np.uint8).reshape(h, 2 * w).astype(np.int32)
    bpp = 2
    left = np.zeros_like(rows)
    left[:, bpp:] = rows[:, :-bpp]
    up = np.zeros_like(rows)
    up[1:] = rows[:-1]
    ul = np.zeros_like(rows)
    ul[1:, bpp:] = rows[:-1, :-bpp]
    if filter_type == 0:
        pred = 0
    elif filter_type == 1:
        pred = left
    elif filter_type == 2:
        pred = up
    elif filter_type == 3:
        pred = (left + up) >> 1
    elif filter_type == 4:
        p = left + up - ul
        pa, pb, pc = abs(p - left), abs(p - up), abs(p - ul)
        pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
    else:
        raise ValueError("filter_type must be 0..4")
    filt = ((rows - pred) & 0xFF).astype(np.uint8)
    raw = np.concatenate([np.full((h, 1), filter_type, np.uint8), filt], axis=1).tobytes()
    return _write_png(path, w, h, raw, level)


def _write_png(path, w, h, raw, level):
    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    with open(path, "wb") as fh:
        fh.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0))
                 + chunk(b"IDAT", zlib.compress(raw, level)) + chunk(b"IEND", b""))
    return path


def _read_matrix(path):
    """read_pose (fusion_inference_dataset.py:123-128): one line of numbers -> square float32 matrix."""
    with open(path, "r") as fh:
        vals = np.asarray([float(t) for t in fh.read().splitlines()[0].split()])
    n = int(np.sqrt(len(vals)))
    return vals.reshape(n, n).astype(np.float32)


class FusionInferenceDataset:
    """fusion_inference_dataset.py:105-146 for the per-frame path: ``dimensions`` and frames in order."""

    def __init__(self, data_dir, scan_id, skip_images=1, downsample_scale=1.0, max_depth=3.0, device="cuda:0",
                 num_images=None):
        # max_depth: cfg.model.ray_tracer.ray_max_dist (fusion_inference_dataset.py:28; 3 m in
        # fusion_pointnet_model.yaml:43).  The frames carry the raw depth image; the cut-off is applied by the
        # kernels (NeuralMap(max_depth=dataset.max_depth)), where the reference's load_depth zeroes the image.
        self.root = os.path.join(data_dir, scan_id)
        self.scan_id = scan_id
        self.device = device
        self.max_depth = max_depth
        self.downsample_scale = float(downsample_scale)
        with open(os.path.join(self.root, "pose", "dimensions.txt"), "r") as fh:
            self.dimensions = np.asarray([float(v) for v in fh.read().splitlines()[0].split()])
        n = len([f for f in os.listdir(os.path.join(self.root, "depth")) if f.endswith(".png")])
        if num_images is not None:
            n = min(n, int(num_images))
        self.indices = list(range(0, n, max(int(skip_images), 1)))

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, k):
        i = self.indices[k]
        depth = read_png16(os.path.join(self.root, "depth", f"{i}.png"))
        intr = _read_matrix(os.path.join(self.root, "pose", f"intr_mat_{i}.txt"))[:3, :3].copy()
        if self.downsample_scale != 1.0:
            # load_depth's dense mode (common.py:96-103): nearest-neighbour resize, intrinsics scaled (:135)
            h, w = depth.shape
            rh, rw = int(h * self.downsample_scale), int(w * self.downsample_scale)
            ys = np.minimum((np.arange(rh) * (h / rh)).astype(np.int64), h - 1)     # cv2.INTER_NEAREST: floor(dst * scale)
            xs = np.minimum((np.arange(rw) * (w / rw)).astype(np.int64), w - 1)
            depth = depth[ys][:, xs]
            intr[:2, :3] *= self.downsample_scale
        return {
            "frame_id": i, "scene_id": self.scan_id, "max_depth": self.max_depth,
            "depth": torch.from_numpy(depth).to(self.device),
            "depth_path": os.path.join(self.root, "depth", f"{i}.png"),
            "intr_mat": intr.astype(np.float64),
            "T_wc": _read_matrix(os.path.join(self.root, "pose", f"T_wc_{i}.txt")).astype(np.float64),
        }

    def __iter__(self):
        for k in range(len(self)):
            yield self[k]


def write_sequence(data_dir, scan_id, depths_u16, intrinsics, poses, dimensions, filter_type=4, level=6):
    """Writes a sequence in the reference's layout (depth PNGs, pose / intrinsics / dimensions text files; one empty
    placeholder per colour image so that the reference's frame count -- len(os.listdir("image")) -- agrees).
    ``depths_u16`` / ``poses`` may be generators (a 2,000-frame sequence is never held in memory)."""
    root = os.path.join(data_dir, scan_id)
    for sub in ("depth", "pose", "image"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    with open(os.path.join(root, "pose", "dimensions.txt"), "w") as fh:
        fh.write(" ".join(repr(float(v)) for v in dimensions) + "\n")
    for i, (d, T) in enumerate(zip(depths_u16, poses)):
        write_png16(os.path.join(root, "depth", f"{i}.png"), d, filter_type, level)
        K = np.asarray(intrinsics[i] if np.ndim(intrinsics) == 3 else intrinsics, dtype=np.float64)
        with open(os.path.join(root, "pose", f"intr_mat_{i}.txt"), "w") as fh:
            fh.write(" ".join(repr(float(v)) for v in K.reshape(-1)) + "\n")
        with open(os.path.join(root, "pose", f"T_wc_{i}.txt"), "w") as fh:
            fh.write(" ".join(repr(float(v)) for v in np.asarray(T, dtype=np.float64).reshape(-1)) + "\n")
        open(os.path.join(root, "image", f"{i}.jpg"), "wb").close()
    return root
