"""A third statement of marching cubes for the mesh tests: plain numpy over the lattice, triangulating with the golden copy of the
reference's triangle table (tests/golden/mc_triangle_table.json, written by tests/golden/make_mc_fixture.py from
src/marching_cubes.cu:401-659). It shares nothing with rnb-neus2_amd/host/mesh.hpp or csrc/kernels_mesh.cuh: vertices are identified by
the lattice edge they sit on, triangles by three such edges, so two extractions are equal iff their triangle SETS are equal."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CORNER = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)])
EDGE = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]  # src/marching_cubes.cu:684-705


def triangle_table():
    with open(os.path.join(HERE, "golden", "mc_triangle_table.json")) as f:
        return json.load(f)["triangles"]


def edge_key(cell, e, shape):
    """Lattice edge of cell (x, y, z): (linear index of its lower corner) * 3 + axis."""
    a, b = CORNER[EDGE[e][0]], CORNER[EDGE[e][1]]
    lo = np.minimum(a, b)
    axis = int(np.nonzero(a != b)[0][0])
    rz, ry, rx = shape
    x, y, z = cell[0] + lo[0], cell[1] + lo[1], cell[2] + lo[2]
    return (x + y * rx + z * rx * ry) * 3 + axis


def triangles_by_edge(density, thresh=0.0):
    """Set of triangles, each a tuple of three edge keys rotated so that the smallest comes first (orientation kept)."""
    table = triangle_table()
    d = np.asarray(density, dtype=np.float32)
    rz, ry, rx = d.shape
    inside = d > np.float32(thresh)
    mask = np.zeros((rz - 1, ry - 1, rx - 1), dtype=np.int32)
    for c, (cx, cy, cz) in enumerate(CORNER):
        mask |= inside[cz:cz + rz - 1, cy:cy + ry - 1, cx:cx + rx - 1].astype(np.int32) << c
    out = set()
    for z, y, x in zip(*np.nonzero((mask != 0) & (mask != 255))):
        t = table[mask[z, y, x]]
        for k in range(0, len(t), 3):
            tri = [edge_key((x, y, z), e, d.shape) for e in t[k:k + 3]]
            r = tri.index(min(tri))
            out.add(tuple(tri[r:] + tri[:r]))
    return out


def edge_vertex(key, density, thresh=0.0, aabb_min=(0, 0, 0), aabb_max=(1, 1, 1)):
    """Position of the vertex on lattice edge `key` (gen_vertices, src/marching_cubes.cu:291-327), float32 arithmetic in its order."""
    d = np.asarray(density, dtype=np.float32)
    rz, ry, rx = d.shape
    idx, axis = divmod(int(key), 3)
    x, y, z = idx % rx, (idx // rx) % ry, idx // (rx * ry)
    f0 = d[z, y, x]
    f1 = d[z + (axis == 2), y + (axis == 1), x + (axis == 0)]
    dt = (np.float32(thresh) - f0) / (f1 - f0)
    p = np.array([x, y, z], dtype=np.float32)
    p[axis] += dt
    sc = (np.asarray(aabb_max, np.float32) - np.asarray(aabb_min, np.float32)) / np.array([rx, ry, rz], np.float32)
    return p * sc + np.asarray(aabb_min, np.float32)


def triangles_of_mesh(verts, idx, density, thresh=0.0, aabb_min=(0, 0, 0), aabb_max=(1, 1, 1)):
    """The same set for an extracted mesh (verts float32[n,3], idx uint32[3m]): every vertex is mapped back to its lattice edge (the
    one coordinate off the lattice names the axis), checked to be a sign change of the lattice and to sit where gen_vertices puts it."""
    d = np.asarray(density, dtype=np.float32)
    rz, ry, rx = d.shape
    mn, mx = np.asarray(aabb_min, np.float64), np.asarray(aabb_max, np.float64)
    q = (verts.astype(np.float64) - mn) / ((mx - mn) / np.array([rx, ry, rz]))
    frac = np.abs(q - np.rint(q))
    axis = np.argmax(frac, axis=1)
    base = np.rint(q).astype(np.int64)
    rows = np.arange(len(q))
    base[rows, axis] = np.floor(q[rows, axis] + 1e-9).astype(np.int64)
    off = np.sort(frac, axis=1)[:, :2]
    assert off.max() < 1e-3, "a vertex is off the lattice edges"
    keys = (base[:, 0] + base[:, 1] * rx + base[:, 2] * rx * ry) * 3 + axis
    assert len(np.unique(keys)) == len(keys), "two vertices on one lattice edge"
    for k in np.random.default_rng(0).choice(len(keys), size=min(200, len(keys)), replace=False):
        np.testing.assert_allclose(verts[k], edge_vertex(keys[k], d, thresh, aabb_min, aabb_max), rtol=0, atol=2e-6)
    out = set()
    for a, b, c in keys[idx.reshape(-1, 3).astype(np.int64)]:
        tri = [int(a), int(b), int(c)]
        r = tri.index(min(tri))
        out.add(tuple(tri[r:] + tri[:r]))
    assert len(out) == len(idx) // 3, "duplicate triangles"
    return out
