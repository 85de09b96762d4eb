"""Iso-surface extraction on the MI355X (rnb_sdf_lattice, rnb_marching_cubes: src/testbed_nerf.cu:4218-4269, src/marching_cubes.cu:276-430,
794-822) against the host loop, and the size BASELINE.json's config 5 asks for (mesh resolution 1024)."""
import time

import numpy as np
import pytest

from tests import mesh_checks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import rnb_neus2_amd as rnb
    c = rnb.Context(target_batch_size=1 << 14, max_rays_per_batch=1 << 12, initial_rays_per_batch=1 << 12)
    c.init_params()
    yield c
    c.close()


@pytest.mark.parametrize("shape", [(40, 50, 70), (33, 33, 33), (2, 2, 2), (1, 5, 5)])
def test_marching_cubes_equals_host_loop(gpu, shape):
    """Vertices and indices bit-identical to the host loop (prefix-sum numbering in lattice order), on a field with many
    ambiguous cells (sum of sines), ragged lattice sizes and degenerate ones."""
    rz, ry, rx = shape
    rng = np.random.default_rng(rx * 1000 + ry)
    z, y, x = np.meshgrid(np.arange(rz), np.arange(ry), np.arange(rx), indexing="ij")
    density = (np.sin(0.9 * x + 0.3) + np.sin(0.7 * y) * np.cos(0.5 * z + 1.0) + 0.3 * rng.standard_normal(shape)).astype(np.float32)
    mn, mx = (-1.0, 0.0, 2.0), (1.0, 3.0, 2.5)
    want_v, want_i = mesh_checks.host_marching_cubes(density, 0.1, mn, mx)
    ptr = gpu.upload(density)
    got_v, got_i = gpu.marching_cubes(ptr, (rx, ry, rz), mn, mx, 0.1)
    gpu.device_free(ptr)
    assert got_v.shape == want_v.shape and got_i.shape == want_i.shape
    assert np.array_equal(got_i, want_i)
    assert np.array_equal(got_v.view(np.uint32), want_v.view(np.uint32))
    if min(shape) > 10:
        assert len(got_v) > 100


@pytest.mark.parametrize("shape", [(19, 23, 31), (12, 12, 12)])
def test_marching_cubes_triangle_set_equals_the_numpy_statement(gpu, shape):
    """Parity with the reference's triangulation, pinned independently: the device extraction against tests/mc_numpy.py -- plain numpy
    over the same lattice with the golden copy of the reference's triangle table (src/marching_cubes.cu:401-659), no code shared with
    host/mesh.hpp or kernels_mesh.cuh. Equal triangle SETS (each triangle = three lattice edges, orientation kept), and vertices where
    gen_vertices (src/marching_cubes.cu:291-327) puts them. Random fields: every one of the 254 non-trivial cases occurs."""
    from tests import mc_numpy
    rz, ry, rx = shape
    rng = np.random.default_rng(rx + 7 * ry)
    density = rng.standard_normal(shape).astype(np.float32)
    mn, mx = (0.0, -1.0, 0.5), (2.0, 1.0, 1.5)
    ptr = gpu.upload(density)
    v, i = gpu.marching_cubes(ptr, (rx, ry, rz), mn, mx, 0.05)
    gpu.device_free(ptr)
    want = mc_numpy.triangles_by_edge(density, 0.05)
    got = mc_numpy.triangles_of_mesh(v, i, density, 0.05, mn, mx)
    assert len(want) > 1000 and got == want, (len(got), len(want), len(got ^ want))


def test_sdf_lattice_matches_point_queries_and_oracle(gpu):
    from tests import oracle_lib
    res = (24, 20, 16)
    ptr = gpu.sdf_lattice(res, 0.0, 1.0, inference=False)
    lat = gpu.download(ptr, res[0] * res[1] * res[2], np.float32).reshape(res[2], res[1], res[0])
    gpu.device_free(ptr)
    zz, yy, xx = np.meshgrid(np.arange(res[2]), np.arange(res[1]), np.arange(res[0]), indexing="ij")
    pts = np.stack([xx / np.float32(res[0]), yy / np.float32(res[1]), zz / np.float32(res[2])], axis=-1).astype(np.float32).reshape(-1, 3)
    assert np.array_equal(lat.ravel(), gpu.sdf(pts, inference=False).astype(np.float32))  # same kernel, same positions
    cpu = oracle_lib.context(target_batch_size=1 << 14, max_rays_per_batch=1 << 12, initial_rays_per_batch=1 << 12)
    try:
        cpu.init_params()
        p2 = cpu.sdf_lattice(res, 0.0, 1.0, inference=False)
        want = cpu.download(p2, lat.size, np.float32)
        cpu.device_free(p2)
    finally:
        cpu.close()
    ok = np.abs(lat.ravel() - want) <= 2e-4 + 2e-3 * np.abs(want)
    assert ok.mean() > 0.999


def test_mesh_resolution_1024(gpu):
    """Config 5's `--resolution 1024`: 2^30 lattice points through the network (EMA weights = the geometric initialisation's
    sphere here), marching cubes on the device. Records time and the memory the call needs (4 + 12 bytes per lattice point)."""
    import torch
    res = 1024
    gpu.set_params(gpu.get("PARAMS_FP32"))  # EMA copy = the initial weights
    free0, _ = torch.cuda.mem_get_info()
    t0 = time.time()
    ptr = gpu.sdf_lattice(res, 0.0, 1.0, inference=True)
    t1 = time.time()
    verts, idx = gpu.marching_cubes(ptr, res)
    t2 = time.time()
    gpu.device_free(ptr)
    print("\n1024^3 lattice: SDF %.2f s, marching cubes %.2f s (incl. download of %d vertices / %d triangles); scratch %.1f GB lattice + %.1f GB edge grid"
          % (t1 - t0, t2 - t1, len(verts), len(idx) // 3, res ** 3 * 4 / 1e9, res ** 3 * 12 / 1e9))
    assert len(verts) > 100_000 and len(idx) % 3 == 0
    c = verts.mean(axis=0)
    r = np.linalg.norm(verts - c, axis=1)
    print("centroid", c, "radius %.4f +- %.4f" % (r.mean(), r.std()))
    assert np.all(np.abs(c - 0.5) < 0.05) and r.std() < 0.1 * r.mean()  # the geometric initialisation is a (lumpy) sphere around the cube's centre
    vol = mesh_checks.assert_closed_oriented(verts, idx)
    assert abs(abs(vol) - 4 / 3 * np.pi * r.mean() ** 3) < 0.15 * abs(vol)
    assert t2 - t0 < 120
