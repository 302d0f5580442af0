"""Marching-cubes case table for the per-voxel mesher (csrc/mesh.hip).

The reference meshes every active voxel's 3x3x3 SDF lattice with skimage's ``marching_cubes``
(sparse_volume.py:740-751).  scikit-image is not part of this environment, so the 256-case table is
GENERATED here rather than transcribed: for a sign configuration of the 8 cube corners

  1. every cube edge whose end points differ in sign carries one surface vertex;
  2. on each of the 6 faces the crossed edges are joined pairwise -- two crossed edges: one segment;
     four (the ambiguous face, diagonal corners alike): the segments that cut off the two INSIDE corners
     individually.  The rule looks only at the 4 corner signs of the face, so the two cubes sharing a
     face draw the same segments and the mesh has no cracks;
  3. the segments form closed loops over the crossed edges; each loop is fan-triangulated and wound so
     that its normal points from the inside (sdf < level) to the outside corners.

Conventions shared with the kernel: corner c = 4*dx + 2*dy + dz (the lattice's flatten order), edge e =
index into EDGES (pairs of corners differing in one bit, ascending).  ``TRI_TABLE[case]`` lists edge
triples, -1 terminated; case bit c is set when corner c is inside (sdf < level).
"""
import itertools

import numpy as np

CORNERS = np.array([[(c >> 2) & 1, (c >> 1) & 1, c & 1] for c in range(8)], dtype=np.int64)
EDGES = [(a, b) for a in range(8) for b in range(a + 1, 8) if (a ^ b) in (1, 2, 4)]
assert len(EDGES) == 12
_EDGE_ID = {e: i for i, e in enumerate(EDGES)}


def _edge(a, b):
    return _EDGE_ID[(min(a, b), max(a, b))]


def _faces():
    """The 6 faces as 4 corners in cyclic order."""
    out = []
    for axis in range(3):
        u, v = [a for a in range(3) if a != axis]
        for side in (0, 1):
            cyc = []
            for du, dv in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = [0, 0, 0]
                p[axis], p[u], p[v] = side, du, dv
                cyc.append(4 * p[0] + 2 * p[1] + p[2])
            out.append(cyc)
    return out


FACES = _faces()


def case_triangles(case):
    """-> list of (e0, e1, e2) edge triples for the 8-bit inside mask ``case``."""
    inside = [(case >> c) & 1 for c in range(8)]
    crossed = [i for i, (a, b) in enumerate(EDGES) if inside[a] != inside[b]]
    if not crossed:
        return []
    nbr = {e: [] for e in crossed}
    for f in FACES:
        fe = [_edge(f[k], f[(k + 1) % 4]) for k in range(4)]          # edge k joins corners k, k+1
        cr = [k for k in range(4) if inside[f[k]] != inside[f[(k + 1) % 4]]]
        if len(cr) == 2:
            pairs = [(fe[cr[0]], fe[cr[1]])]
        elif len(cr) == 4:
            # corners alternate; cut off each inside corner k with the segment (edge k-1, edge k)
            pairs = [(fe[(k - 1) % 4], fe[k]) for k in range(4) if inside[f[k]]]
        else:
            pairs = []
        for a, b in pairs:
            nbr[a].append(b)
            nbr[b].append(a)
    assert all(len(v) == 2 for v in nbr.values()), (case, nbr)
    mid = {e: (CORNERS[EDGES[e][0]] + CORNERS[EDGES[e][1]]) / 2.0 for e in crossed}
    tris, seen = [], set()
    for start in crossed:
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        seen.add(start)
        while True:
            nxt = [n for n in nbr[cur] if n != prev]
            n = nxt[0] if nxt else nbr[cur][0]
            if n == start:
                break
            if n in seen:            # two segments between the same pair of edges (cannot happen on a cube)
                raise AssertionError((case, loop))
            loop.append(n)
            seen.add(n)
            prev, cur = cur, n
        # orientation: Newell normal of the loop vs (outside end points - inside end points)
        pts = np.array([mid[e] for e in loop])
        nrm = np.zeros(3)
        for k in range(len(loop)):
            p, q = pts[k], pts[(k + 1) % len(loop)]
            nrm += np.cross(p, q)
        g = np.zeros(3)
        for e in loop:
            a, b = EDGES[e]
            g += (CORNERS[b] - CORNERS[a]) * (1 if inside[a] else -1)
        if np.dot(nrm, g) < 0:
            loop = loop[::-1]
        for k in range(1, len(loop) - 1):
            tris.append((loop[0], loop[k], loop[k + 1]))
    return tris


def build_tables():
    """-> (tri_table int8 [256, 3*T+1] with -1 padding, n_tri int32 [256])."""
    all_tris = [case_triangles(c) for c in range(256)]
    t_max = max(len(t) for t in all_tris)
    table = -np.ones((256, 3 * t_max + 1), dtype=np.int8)
    for c, tris in enumerate(all_tris):
        flat = list(itertools.chain.from_iterable(tris))
        table[c, : len(flat)] = flat
    return table, np.array([len(t) for t in all_tris], dtype=np.int32)


TRI_TABLE, N_TRI = build_tables()
MAX_TRI = int(N_TRI.max())
EDGE_CORNERS = np.array(EDGES, dtype=np.int32)          # [12, 2]
