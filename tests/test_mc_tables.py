"""Marching-cubes case table (bnv_fusion_amd/mc_tables.py) against the oracle's independent per-cell
triangulation and against table-independent geometric properties.  CPU only.  (Parity with scikit-image's
Lewiner tables is unpinned -- the package is absent; see oracle/bnv_oracle.py.)"""
from collections import Counter

import numpy as np

from bnv_fusion_amd import mc_tables as M
from oracle import bnv_oracle as orc


def _rot(t):
    k = t.index(min(t))
    return tuple(t[k:] + t[:k])


def test_table_matches_oracle_triangulation_for_all_256_cases():
    assert M.TRI_TABLE.shape == (256, 16) and M.MAX_TRI == 5
    assert [tuple(e) for e in M.EDGES] == [tuple(e) for e in orc._MC_EDGES]
    for case in range(256):
        inside = [bool((case >> c) & 1) for c in range(8)]
        row = [int(x) for x in M.TRI_TABLE[case] if x >= 0]
        assert len(row) == 3 * M.N_TRI[case]
        tris = [tuple(row[i: i + 3]) for i in range(0, len(row), 3)]
        # same crossed-edge set, same loops (cyclic vertex order) as the oracle
        crossed = {e for e, (a, b) in enumerate(orc._MC_EDGES) if inside[a] != inside[b]}
        assert {e for t in tris for e in t} == crossed
        loops = orc._mc_cell_loops(inside) if crossed else []
        assert len(tris) == sum(len(lp) - 2 for lp in loops)
        # every triangle's vertices belong to one loop
        for t in tris:
            assert any(set(t) <= set(lp) for lp in loops)


def test_sphere_mesh_is_closed_oriented_and_has_the_right_area():
    """A table-level end-to-end check with the table driving a numpy mesher: the zero set of a sphere SDF over
    12^3 voxels must come out as a closed, consistently oriented 2-manifold of area ~ 4 pi r^2."""
    R, c = 3.3, np.array([6.2, 6.1, 5.9])
    o = np.stack(np.meshgrid(*[np.arange(12)] * 3, indexing="ij"), -1).reshape(-1, 3)
    r = np.arange(3) * 0.5 - 0.5
    lat = np.stack(np.meshgrid(r, r, r, indexing="ij"), -1)
    sdf = (np.linalg.norm(o[:, None, None, None, :] + lat[None] - c, axis=-1) - R).astype(np.float32)
    tris = []
    for v in range(len(o)):
        s = sdf[v]
        if not (s.max() > 0 and s.min() < 0):
            continue
        for cell in range(8):
            cc = np.array([cell >> 2, (cell >> 1) & 1, cell & 1])
            val = [s[tuple(cc + M.CORNERS[k])] for k in range(8)]
            case = sum(int(val[k] < 0) << k for k in range(8))
            row = [int(x) for x in M.TRI_TABLE[case] if x >= 0]
            for e in row:
                a, b = M.EDGES[e]
                t = (0 - val[a]) / (val[b] - val[a])
                p = (cc + M.CORNERS[a] + t * (M.CORNERS[b] - M.CORNERS[a])) * 0.5 + o[v] - 0.5
                tris.append(p)
    tri = np.array(tris).reshape(-1, 3, 3)
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1).sum()
    assert abs(area - 4 * np.pi * R * R) < 0.02 * 4 * np.pi * R * R
    q = np.round(tri.reshape(-1, 3) * 4096).astype(np.int64)
    _, inv = np.unique((q[:, 0] << 42) + (q[:, 1] << 21) + q[:, 2], return_inverse=True)
    f = inv.reshape(-1, 3)
    cnt = Counter()
    for a, b, d in f:
        for e in ((a, b), (b, d), (d, a)):
            cnt[e] += 1
    assert all(n == 1 and cnt.get((b, a), 0) == 1 for (a, b), n in cnt.items())     # closed + consistently wound
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    assert (np.einsum("ij,ij->i", nrm, tri.mean(1) - c) > 0).all()                  # normals point to sdf > 0
