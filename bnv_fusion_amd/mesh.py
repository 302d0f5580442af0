"""Mesh extraction from decoded SDF lattices: per-voxel marching cubes on the GPU (csrc/mesh.hip), the
last stage of SparseVolume.meshlize (src/models/sparse_volume.py:740-766).

``marching_cubes_lattice_indexed`` produces what the reference's loop produces: per voxel a vertex list without
duplicates and faces indexing it, concatenated with ``faces + last_face_id`` / ``last_face_id += max(faces) + 1``
(:748-751).  ``marching_cubes_lattice`` is the older triangle-soup form (3 vertices per face).

``TriMesh`` stands in for the ``trimesh.Trimesh(vertices, faces, process=False)`` the reference returns:
``.vertices`` [V, 3] float32, ``.faces`` [T, 3] int64 (numpy, like trimesh), ``.export(path)`` (binary PLY).
"""
import ctypes as C
import struct

import numpy as np
import torch

from . import _lib
from .mc_tables import TRI_TABLE

_TABLES = {}


def _table(device):
    key = str(device)
    if key not in _TABLES:
        _TABLES[key] = torch.from_numpy(np.ascontiguousarray(TRI_TABLE)).to(device)
    return _TABLES[key]


class TriMesh:
    def __init__(self, vertices, faces):
        self.vertices = np.asarray(vertices, dtype=np.float32).reshape(-1, 3)
        self.faces = np.asarray(faces, dtype=np.int64).reshape(-1, 3)

    def export(self, path):
        """Binary little-endian PLY (what trimesh writes for a ``.ply`` path)."""
        v, f = self.vertices, self.faces
        header = ("ply\nformat binary_little_endian 1.0\n"
                  f"element vertex {len(v)}\nproperty float x\nproperty float y\nproperty float z\n"
                  f"element face {len(f)}\nproperty list uchar int vertex_indices\nend_header\n")
        rec = np.empty(len(f), dtype=[("n", "u1"), ("i", "<i4", 3)])
        rec["n"] = 3
        rec["i"] = f.astype(np.int32)
        with open(path, "wb") as fh:
            fh.write(header.encode("ascii"))
            fh.write(v.astype("<f4").tobytes())
            fh.write(rec.tobytes())
        return path

    def merge_vertices(self):
        """Weld coincident vertices (trimesh's ``merge_vertices``): meshlize shares vertices inside a voxel only,
        neighbouring voxels repeat the vertices on their common lattice edges."""
        u, inv = np.unique(np.ascontiguousarray(self.vertices).view([("x", "<f4"), ("y", "<f4"), ("z", "<f4")]).reshape(-1),
                           return_inverse=True)
        self.vertices = u.view(np.float32).reshape(-1, 3).copy()
        self.faces = inv.reshape(-1)[self.faces].astype(np.int64)
        return self


def marching_cubes_lattice(sdf, origins, voxel_size, min_coords, level=0.0, n_dev=None):
    """sdf [n, 27] (or [n, 3, 3, 3]) float32 on the GPU: the lattice {-.5, 0, .5}^3 of every voxel;
    origins [n, 3] int64 voxel coordinates.  -> (vertices [3T, 3] f32 world coordinates, faces [T, 3] i64)
    on the device, as a triangle soup (faces = arange)."""
    lib = _lib.load()
    sdf = sdf.detach().reshape(-1, 27).float().contiguous()
    origins = origins.detach().reshape(-1, 3).long().contiguous()
    n = int(sdf.shape[0])
    dev = sdf.device
    assert origins.shape[0] == n
    if n == 0:
        return torch.zeros((0, 3), device=dev), torch.zeros((0, 3), dtype=torch.int64, device=dev)
    table = _table(dev)
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(lib.bnv_mc_count(_lib.ptr(sdf), n, _lib.ptr(n_dev), float(level), _lib.ptr(table), _lib.ptr(counts),
                                _lib.stream_ptr()), "bnv_mc_count")
    ends = torch.cumsum(counts.long(), 0)
    total = int(ends[-1])                       # the one host read: the mesh has to be allocated
    offsets = (ends - counts.long()).contiguous()
    verts = torch.empty((3 * total, 3), dtype=torch.float32, device=dev)
    if total:
        mn = (C.c_float * 3)(*[float(x) for x in torch.as_tensor(min_coords).reshape(-1)[:3].tolist()])
        _lib.check(lib.bnv_mc_emit(_lib.ptr(sdf), _lib.ptr(origins), n, _lib.ptr(n_dev), float(level),
                                   float(voxel_size), mn, _lib.ptr(table), _lib.ptr(offsets), _lib.ptr(verts),
                                   _lib.stream_ptr()), "bnv_mc_emit")
    faces = torch.arange(3 * total, dtype=torch.int64, device=dev).reshape(-1, 3)
    return verts, faces


def marching_cubes_lattice_indexed(sdf, origins, voxel_size, min_coords, level=0.0, n_dev=None):
    """sdf [n, 27] (or [n, 3, 3, 3]) float32 on the GPU; origins [n, 3] int64.  -> (vertices [V, 3] f32 world
    coordinates, faces [T, 3] i64, n_verts [n] i32, n_tris [n] i32) on the device: the concatenation
    SparseVolume.meshlize builds (sparse_volume.py:740-756) -- voxels that fail the gate contribute nothing, a voxel's
    faces index its own vertices offset by the vertex counts of the voxels before it."""
    lib = _lib.load()
    sdf = sdf.detach().reshape(-1, 27).float().contiguous()
    origins = origins.detach().reshape(-1, 3).long().contiguous()
    n = int(sdf.shape[0])
    dev = sdf.device
    assert origins.shape[0] == n
    if n == 0:
        z = torch.zeros(0, dtype=torch.int32, device=dev)
        return torch.zeros((0, 3), device=dev), torch.zeros((0, 3), dtype=torch.int64, device=dev), z, z
    table = _table(dev)
    nv = torch.empty(n, dtype=torch.int32, device=dev)
    nt = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(lib.bnv_mc_count_indexed(_lib.ptr(sdf), n, _lib.ptr(n_dev), float(level), _lib.ptr(table),
                                        _lib.ptr(nv), _lib.ptr(nt), _lib.stream_ptr()), "bnv_mc_count_indexed")
    ve, te = torch.cumsum(nv.long(), 0), torch.cumsum(nt.long(), 0)
    totals = torch.stack([ve[-1], te[-1]]).tolist()         # the one host read: the mesh has to be allocated
    V, T = int(totals[0]), int(totals[1])
    verts = torch.empty((V, 3), dtype=torch.float32, device=dev)
    faces = torch.empty((T, 3), dtype=torch.int64, device=dev)
    if T:
        mn = (C.c_float * 3)(*[float(x) for x in torch.as_tensor(min_coords).reshape(-1)[:3].tolist()])
        voff, toff = (ve - nv.long()).contiguous(), (te - nt.long()).contiguous()
        _lib.check(lib.bnv_mc_emit_indexed(_lib.ptr(sdf), _lib.ptr(origins), n, _lib.ptr(n_dev), float(level),
                                           float(voxel_size), mn, _lib.ptr(table), _lib.ptr(voff), _lib.ptr(toff),
                                           _lib.ptr(verts), _lib.ptr(faces), _lib.stream_ptr()), "bnv_mc_emit_indexed")
    return verts, faces, nv, nt
