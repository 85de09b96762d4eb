"""Shared checks for iso-surface tests (CPU oracle twin and HIP library)."""
import numpy as np


def host_marching_cubes(density, thresh=0.0, aabb_min=(0, 0, 0), aabb_max=(1, 1, 1)):
    """The checker's own marching cubes (oracle/orc_mesh.h, vertices and triangles numbered by prefix sums in lattice order) through the oracle
    library's entry point: the ordering every device implementation must reproduce bit for bit."""
    from tests import oracle_lib
    cpu = oracle_lib.context(target_batch_size=128, max_rays_per_batch=128, initial_rays_per_batch=128, n_levels=2, log2_hashmap_size=12)
    try:
        d = np.ascontiguousarray(density, dtype=np.float32)
        rz, ry, rx = d.shape
        ptr = cpu.upload(d)
        v, i = cpu.marching_cubes(ptr, (rx, ry, rz), aabb_min, aabb_max, thresh)
        cpu.device_free(ptr)
        return v, i
    finally:
        cpu.close()


def assert_closed_oriented(verts, idx, inside_point=None):
    """Every edge is shared by exactly two triangles with opposite directions (closed, consistently oriented surface)."""
    t = idx.reshape(-1, 3).astype(np.int64)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    key = e[:, 0] * (len(verts) + 1) + e[:, 1]
    rev = e[:, 1] * (len(verts) + 1) + e[:, 0]
    assert len(np.unique(key)) == len(key), "a directed edge is used twice"
    assert np.array_equal(np.sort(key), np.sort(rev)), "an edge lacks its opposite: the surface is open or inconsistently oriented"
    a, b, c = verts[t[:, 0]].astype(np.float64), verts[t[:, 1]].astype(np.float64), verts[t[:, 2]].astype(np.float64)
    return float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0)  # signed volume
