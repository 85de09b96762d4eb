"""Python side of the drop-in boundary (SURVEY §8b "Python side", §8f-3/4): pipeline orchestration, data preparation,
scene scaling, albedo scaling, mesh post-processing.

Pinned against the reference where it can run in the build container: tests/golden/pipeline_argv.json (command lines
rnb_neus2/pipeline.py issues, recorded with a stub testbed) and tests/golden/scaling_vectors.json (rnb_neus2/scaling.py
outputs) — both produced by tests/golden/make_python_fixtures.py. The cv2 / trimesh dependent parts of the reference
cannot run there; they are checked against analytic ground truth instead."""
import json
import os
import stat
import sys

import numpy as np
import pytest

from rnb_neus2_amd import albedo_scaling, dataloaders, hostlib, image_io, meshproc, pipeline, prepare, scaling, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

STUB = r'''#!/usr/bin/env python3
import json, os, sys
a = sys.argv[1:]
scene = a[a.index("--scene") + 1].rstrip("/")
it = a[a.index("--maxiter") + 1]
with open(os.environ["STUB_LOG"], "a") as f:
    f.write(json.dumps(a) + "\n")
os.makedirs(os.path.join(scene, "output"), exist_ok=True)
if "--save-snapshot" in a:
    open(os.path.join(scene, "output", "snapshot_%s.msgpack" % it), "wb").close()
if "--save-mesh" in a:
    open(os.path.join(scene, "output", "mesh_%s.obj" % it), "w").close()
print("iteration=100 loss=0.5")
'''


class Log:
    def __init__(self):
        self.lines = []

    def info(self, m):
        self.lines.append(str(m))

    warning = error = info


# ------------------------------------------------------------------------------- orchestration vs the reference's argv
@pytest.fixture()
def recorder(tmp_path, monkeypatch):
    root = str(tmp_path)
    tb = os.path.join(root, "testbed")
    with open(tb, "w") as f:
        f.write(STUB)
    os.chmod(tb, os.stat(tb).st_mode | stat.S_IEXEC)
    monkeypatch.setenv("STUB_LOG", os.path.join(root, "argv.jsonl"))
    calls = []

    def ratios(**kw):
        calls.append(["compute_albedo_scale_ratios", {k: ("<logger>" if hasattr(v, "info") else v) for k, v in kw.items()}])
        return np.ones((1, 3))

    def scale(**kw):
        calls.append(["scale_and_save_albedos", {k: ("<ratios>" if k == "scale_ratios" else "<logger>" if hasattr(v, "info") else v) for k, v in kw.items()}])
        os.makedirs(kw["output_albedo_path"], exist_ok=True)

    def prep(data, out, logger, **kw):
        calls.append(["prepare_testbed_data", out, kw])
        os.makedirs(os.path.join(out, "albedos"), exist_ok=True)

    monkeypatch.setattr(albedo_scaling, "compute_albedo_scale_ratios", ratios)
    monkeypatch.setattr(albedo_scaling, "scale_and_save_albedos", scale)
    monkeypatch.setattr(dataloaders, "load_data", lambda p, **kw: calls.append(["load_data", p, {k: v for k, v in kw.items() if k != "logger"}]) or {"views": []})
    monkeypatch.setattr(prepare, "prepare_testbed_data", prep)
    monkeypatch.setattr(pipeline, "postprocess_mesh", lambda d, o, logger=None: calls.append(["postprocess_mesh", d, o]))

    def result():
        log = os.path.join(root, "argv.jsonl")
        argv = [[s.replace(root, "<ROOT>") for s in json.loads(l)] for l in open(log)] if os.path.exists(log) else []
        return argv, json.loads(json.dumps(calls, default=str).replace(root, "<ROOT>"))

    def scene():
        d = os.path.join(root, "scene")
        os.makedirs(os.path.join(d, "albedos"), exist_ok=True)
        return d

    return dict(root=root, testbed=tb, result=result, scene=scene)


CASES = {
    "two_stage_default": lambda r: pipeline.run_two_stage(r["testbed"], r["scene"](), 10000, ["--mask-weight", "1.0"], logger=Log()),
    "two_stage_no_albedo_res512_extra": lambda r: pipeline.run_two_stage(r["testbed"], r["scene"](), 1000, ["--mask-weight", "0.5", "--lone"], resolution=512, no_albedo=True,
                                                                         extra_flags=["--bce"], logger=Log()),
    "two_stage_odd_steps": lambda r: pipeline.run_two_stage(r["testbed"], r["scene"](), 100, [], no_albedo=True, logger=Log()),
    "albedo_scaling_default": lambda r: pipeline.run_with_albedo_scaling(r["testbed"], r["scene"](), 20000, ["--mask-weight", "1.0"], logger=Log()),
    "albedo_scaling_short": lambda r: pipeline.run_with_albedo_scaling(r["testbed"], r["scene"](), 3000, ["--mask-weight", "2.0", "--supernormal"], resolution=256, warmup_ratio=0.5,
                                                                       n_samples=50, logger=Log()),
    "full_default": lambda r: pipeline.run_full_pipeline(os.path.join(r["root"], "in"), r["testbed"], os.path.join(r["root"], "out"), logger=Log()),
    "full_flags": lambda r: pipeline.run_full_pipeline(os.path.join(r["root"], "in.sfm"), r["testbed"], os.path.join(r["root"], "out"), max_steps=3000, mesh_resolution=256,
                                                       scaling_mode="cameras", sphere_scale=0.8, margin_px=5, mask_weight=0.25, super_normal=True, use_l1=True,
                                                       use_rgb_plus=False, albedo_sfm_path="a.sfm", mask_sfm_path="m.sfm", mask_folder_path="masks", logger=Log()),
    "full_albedo": lambda r: pipeline.run_full_pipeline(os.path.join(r["root"], "in"), r["testbed"], os.path.join(r["root"], "out"), max_steps=6000, has_albedo=True,
                                                        warmup_ratio=0.25, n_samples=77, logger=Log()),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_pipeline_issues_the_reference_command_lines(name, recorder):
    with open(os.path.join(GOLDEN, "pipeline_argv.json")) as f:
        golden = {c["name"]: c for c in json.load(f)["cases"]}
    assert set(golden) == set(CASES)
    CASES[name](recorder)
    argv, other = recorder["result"]()
    assert argv == golden[name]["testbed_argv"]
    assert other == golden[name]["other_calls"]


def test_pipeline_error_paths(recorder, tmp_path):
    bad = tmp_path / "failing_testbed"
    bad.write_text("#!/bin/sh\necho oops >&2\nexit 3\n")
    bad.chmod(0o755)
    log = Log()
    with pytest.raises(RuntimeError, match="Stage 1 failed with code 3"):
        pipeline.run_two_stage(str(bad), recorder["scene"](), 30, [], logger=log)
    assert any("oops" in l for l in log.lines)
    silent = tmp_path / "no_snapshot_testbed"
    silent.write_text("#!/bin/sh\nexit 0\n")
    silent.chmod(0o755)
    with pytest.raises(RuntimeError, match="Snapshot not found after 20 iterations"):
        pipeline.run_two_stage(str(silent), recorder["scene"](), 30, [], logger=Log())
    with pytest.raises(RuntimeError, match="Phase 1 mesh not found"):
        pipeline.run_with_albedo_scaling(str(silent), recorder["scene"](), 30, [], logger=Log())


def test_snapshot_fallback_location(recorder, tmp_path):
    """A snapshot next to the scene (old layout) is used when <scene>/output has none. (pipeline.py:77-85)"""
    tb = tmp_path / "legacy_testbed"
    tb.write_text("#!/bin/sh\necho \"$@\" >> %s/legacy.log\nexit 0\n" % tmp_path)
    tb.chmod(0o755)
    scene = recorder["scene"]()
    open(os.path.join(scene, "snapshot_20.msgpack"), "wb").close()
    pipeline.run_two_stage(str(tb), scene, 30, [], logger=Log())
    assert "--snapshot %s" % os.path.join(scene, "snapshot_20.msgpack") in open(tmp_path / "legacy.log").read()


def test_run_pipeline_cli_surface():
    """run_pipeline.py:27-92 — option names, defaults and their mapping onto run_full_pipeline's keywords."""
    sys.path.insert(0, ROOT)
    import run_pipeline
    p = run_pipeline.build_parser()
    a = p.parse_args(["-i", "in", "-t", "tb"])
    assert run_pipeline.pipeline_kwargs(a) == dict(input_path="in", testbed_path="tb", output_dir="output", max_steps=10000, mesh_resolution=1024, scaling_mode="auto",
                                                   sphere_scale=1.0, margin_px=20, warmup_ratio=0.1, mask_weight=1.0, super_normal=False, use_l1=False, use_rgb_plus=True,
                                                   has_albedo=False, albedo_sfm_path="", mask_sfm_path="", mask_folder_path="", n_samples=2000)
    assert a.seed == 0
    a = p.parse_args("--input x --testbed y --output o --max-steps 5 --mesh-resolution 64 --scaling-mode cameras --sphere-scale 0.5 --margin-px 3 --warmup-ratio 0.2 "
                     "--mask-weight 0.1 --has-albedo --albedo-sfm a --mask-sfm m --mask-folder f --supernormal --l1 --no-rgbplus --n-samples 9 --seed 4".split())
    kw = run_pipeline.pipeline_kwargs(a)
    assert kw["super_normal"] and kw["use_l1"] and not kw["use_rgb_plus"] and kw["has_albedo"] and kw["n_samples"] == 9 and kw["scaling_mode"] == "cameras"
    with pytest.raises(SystemExit):
        p.parse_args(["-i", "in"])  # --testbed is required
    with pytest.raises(SystemExit):
        p.parse_args(["-i", "in", "-t", "tb", "--scaling-mode", "bogus"])


# ------------------------------------------------------------------------------- scaling vs the reference's outputs
@pytest.fixture(scope="module")
def scaling_vectors():
    with open(os.path.join(GOLDEN, "scaling_vectors.json")) as f:
        return json.load(f)


def test_unit_sphere_scaling_matches_reference(scaling_vectors):
    for case in scaling_vectors["unit_sphere"]:
        c, f, m = scaling.compute_unit_sphere_scaling(np.array(case["points"]), case["sphere_scale"])
        np.testing.assert_allclose(c, case["center"], rtol=1e-12, atol=1e-12)
        assert abs(f - case["scale_factor"]) <= 1e-12 * abs(f)
        np.testing.assert_allclose(m, np.array(case["scale_matrix"], np.float32), rtol=1e-6, atol=1e-7)
        assert m.dtype == np.float32


def test_silhouette_scaling_matches_reference(scaling_vectors):
    for case in scaling_vectors["silhouettes"]:
        res = case["res"]
        cams = [dict(fx=c["fx"], fy=c["fy"], cx=c["cx"], cy=c["cy"], R_cam2world=np.array(c["R_cam2world"]), center=np.array(c["center"])) for c in case["cameras"]]
        masks = [np.unpackbits(np.array(p, np.uint8))[:res * res].reshape(res, res).astype(np.float32) for p in case["masks_packed"]]
        c, f = scaling.compute_scaling_from_silhouettes(cams, masks)
        np.testing.assert_allclose(c, case["default"]["center"], rtol=1e-9, atol=1e-10)
        assert abs(f - case["default"]["scale_factor"]) <= 1e-9 * f
        c, f = scaling.compute_scaling_from_silhouettes(cams, masks, sphere_scale=case["alt"]["sphere_scale"], fg_area_ratio=case["alt"]["fg_area_ratio"])
        assert abs(f - case["alt"]["scale_factor"]) <= 1e-9 * f
        np.testing.assert_allclose(scaling._triangulate_scene_center(cams, masks), case["triangulated"], rtol=1e-9, atol=1e-10)


def _ring_cameras(n, res, centre, dist=4.0, f_rel=1.2):
    cams = []
    for k in range(n):
        ang = 2 * np.pi * k / n
        eye = np.asarray(centre) + dist * np.array([np.cos(ang), 0.3 * np.sin(2 * ang), np.sin(ang)])
        fwd = np.asarray(centre) - eye
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, [0, 1, 0])
        right /= np.linalg.norm(right)
        R = np.stack([right, np.cross(fwd, right), fwd], axis=1)
        cams.append(dict(fx=f_rel * res, fy=f_rel * res, cx=res / 2, cy=res / 2, R_cam2world=R, center=eye))
    return cams


def _sphere_masks(cams, res, centre, radius):
    masks = []
    yy, xx = np.mgrid[0:res, 0:res]
    for c in cams:
        d = np.stack([(xx - c["cx"]) / c["fx"], (yy - c["cy"]) / c["fy"], np.ones_like(xx, float)], -1) @ c["R_cam2world"].T
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        oc = np.asarray(centre) - c["center"]
        b = d @ oc
        masks.append(((b * b - oc @ oc + radius * radius) >= 0).astype(np.float32))
    return masks


def test_silhouettes_v2_encloses_the_object():
    """Minimum enclosing sphere: for a sphere of radius r seen by a ring of cameras, centre is recovered and the fitted
    radius is r plus the pixel margin (margin_px * Z / f)."""
    centre, r, res = np.array([0.3, -0.2, 0.1]), 0.6, 160
    cams = _ring_cameras(8, res, centre)
    masks = _sphere_masks(cams, res, centre, r)
    c, f = scaling.compute_scaling_from_silhouettes_v2(cams, masks, sphere_scale=1.0, margin_px=4, percentile=100)
    assert np.linalg.norm(c - centre) < 0.03
    fitted = 1.0 / f
    margin = 4 * 4.0 / (1.2 * res)
    assert r + 0.5 * margin < fitted < r * 1.06 + 1.5 * margin
    assert c.dtype == np.float32 and isinstance(f, float)
    c0, f0 = scaling.compute_scaling_from_silhouettes_v2(cams, [np.zeros((res, res), np.float32)] * 8, sphere_scale=2.0)
    assert f0 == 2.0  # no contours anywhere -> untouched scale, triangulated/fallback centre


def test_outer_contour_matches_border_following_on_shapes():
    m = np.zeros((12, 12), np.float32)
    m[2:9, 3:10] = 1
    m[4:6, 5:7] = 0  # a hole: external contours ignore it
    pts = scaling._outer_contour_points(m)
    assert len(pts) == 2 * 7 + 2 * 5
    assert pts[:, 0].min() == 3 and pts[:, 0].max() == 9 and pts[:, 1].min() == 2 and pts[:, 1].max() == 8
    hull = scaling._convex_hull(pts)
    assert sorted(map(tuple, hull)) == [(3.0, 2.0), (3.0, 8.0), (9.0, 2.0), (9.0, 8.0)]


# ------------------------------------------------------------------------------- PNG codec / image helpers
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
@pytest.mark.parametrize("channels", [1, 2, 3, 4])
def test_png_roundtrip(tmp_path, dtype, channels):
    rng = np.random.default_rng(channels)
    shape = (13, 7) if channels == 1 else (13, 7, channels)
    img = rng.integers(0, np.iinfo(dtype).max + 1, size=shape).astype(dtype)
    path = str(tmp_path / "x.png")
    hostlib.png_write(path, img, level=6)
    assert hostlib.png_info(path) == (7, 13, channels, 8 * img.dtype.itemsize)
    back = hostlib.png_read(path)
    assert back.dtype == dtype and np.array_equal(back, img)
    from PIL import Image
    if dtype == np.uint8 or channels == 1:  # PIL decodes these layouts faithfully: an independent decoder agrees with the writer
        pil = np.array(Image.open(path))
        assert np.array_equal(pil.astype(np.int64), img.astype(np.int64))


def test_png_reads_foreign_encoders(tmp_path):
    """Files written by another encoder (PIL: adaptive filters, palette, interlace-free) decode to the same samples."""
    from PIL import Image
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, size=(31, 17, 3)).astype(np.uint8)
    rgb[:, :, 1] = np.linspace(0, 255, 17).astype(np.uint8)[None, :]  # smooth channel so that Sub/Up/Paeth filters get chosen
    Image.fromarray(rgb).save(tmp_path / "rgb.png", optimize=True)
    assert np.array_equal(hostlib.png_read(tmp_path / "rgb.png"), rgb)
    g16 = rng.integers(0, 65536, size=(9, 21)).astype(np.uint16)
    Image.fromarray(g16).save(tmp_path / "g16.png")
    back = hostlib.png_read(tmp_path / "g16.png")
    assert back.dtype == np.uint16 and np.array_equal(back, g16)
    pal = Image.fromarray(rgb).quantize(16)
    pal.save(tmp_path / "pal.png")
    assert np.array_equal(hostlib.png_read(tmp_path / "pal.png"), np.array(pal.convert("RGB")))
    rgba16 = hostlib.png_read_rgba16(tmp_path / "rgb.png")
    assert np.array_equal(rgba16[:, :, :3], rgb.astype(np.uint16) * 257) and np.all(rgba16[:, :, 3] == 65535)  # stbi_load_16 widening
    with pytest.raises(RuntimeError):
        hostlib.png_info(tmp_path / "missing.png")


def test_image_io_helpers(tmp_path):
    img = np.array([[[0.0, 0.5, 1.0, 1.0], [np.nan, -1.0, 2.0, 0.25]]], np.float32)
    image_io.save_image(img, tmp_path / "a.png", bit_depth=16)
    raw = hostlib.png_read(tmp_path / "a.png")
    assert raw.dtype == np.uint16 and raw.tolist() == [[[0, 32767, 65535, 65535], [0, 0, 65535, 16383]]]
    np.testing.assert_allclose(image_io.load_image(tmp_path / "a.png"), raw / 65535.0, rtol=1e-6)
    image_io.save_image(img, tmp_path / "b.png", bit_depth=8)
    assert hostlib.png_read(tmp_path / "b.png").tolist() == [[[0, 127, 255, 255], [0, 0, 255, 63]]]
    n = np.array([[[0.0, 0.0, 1.0], [-1.0, 1.0, 0.5]]], np.float32)
    image_io.save_normal_16bit(n, tmp_path / "n.png")
    np.testing.assert_allclose(image_io.load_normal(tmp_path / "n.png"), n, atol=2e-5)
    with pytest.raises(FileNotFoundError):
        image_io.load_image(tmp_path / "nope.png")
    with pytest.raises(FileNotFoundError):
        image_io.load_image(tmp_path / "x.exr")


@pytest.mark.parametrize("compression", ["none", "zips", "zip"])
@pytest.mark.parametrize("pixel_type", ["float", "half"])
def test_exr_roundtrip(tmp_path, compression, pixel_type):
    """The scanline EXR codec (rnb-neus2_amd/exr.py; the reference reads EXR through cv2): all supported compressions and pixel
    types, sizes that are not multiples of the 16-line ZIP block, 1 / 3 / 4 channels, RGB order in and out."""
    from rnb_neus2_amd import exr
    rng = np.random.default_rng(3)
    for shape in ((37, 23, 3), (16, 5, 4), (1, 1, 3), (40, 31)):
        img = (rng.standard_normal(shape) * 2).astype(np.float32)
        if pixel_type == "half":
            img = img.astype(np.float16).astype(np.float32)
        img[..., 0] = np.linspace(-1, 1, img[..., 0].size, dtype=np.float32).reshape(img[..., 0].shape) if img.ndim == 3 else img[..., 0]
        if pixel_type == "half":
            img = img.astype(np.float16).astype(np.float32)
        path = tmp_path / ("%s_%s_%d.exr" % (compression, pixel_type, len(shape) * 100 + shape[0]))
        exr.write_exr(path, img, compression=compression, pixel_type=pixel_type)
        back = exr.read_exr(path)
        assert back.dtype == np.float32 and back.shape == img.shape
        assert np.array_equal(back, img)
    # smooth data really is deflated (the predictor + byte split is in place, not just a stored block)
    smooth = np.tile(np.linspace(0, 1, 256, dtype=np.float32)[None, :, None], (64, 1, 3))
    exr.write_exr(tmp_path / "s.exr", smooth, compression=compression, pixel_type=pixel_type)
    size = (tmp_path / "s.exr").stat().st_size
    assert (size < 0.7 * smooth.nbytes) == (compression != "none" or pixel_type == "half")
    with open(tmp_path / "bad.exr", "wb") as f:
        f.write(b"not an exr file at all")
    assert image_io.read_unchanged(tmp_path / "bad.exr") is None


def test_exr_normals_albedos_masks_through_prepare(tmp_path):
    """EXR inputs take the reference's conversions (prepare.py:23-42, 161-190): normals in [-1, 1] -> (n + 1) / 2 * 65535, albedos
    clipped to [0, 1] * 65535, float masks thresholded at 0.5; load_normal returns EXR values as stored (image_io.py:91-110)."""
    from rnb_neus2_amd import exr
    h = w = 24
    n = np.zeros((h, w, 3), np.float32)
    n[..., 0], n[..., 1], n[..., 2] = -1.0, 0.25, 1.0
    a = np.zeros((h, w, 3), np.float32)
    a[..., 0], a[..., 1], a[..., 2] = 1.5, 0.5, -0.2
    m = np.zeros((h, w), np.float32)
    m[4:20, 6:18] = 0.9
    m[0, 0] = 0.4
    image_io.save_normal_exr(n, tmp_path / "n.exr")
    image_io.save_exr(a, tmp_path / "a.exr")
    exr.write_exr(tmp_path / "m.exr", m, compression="zips")
    np.testing.assert_array_equal(image_io.load_normal(tmp_path / "n.exr"), n)
    c2w = np.eye(4, dtype=np.float32)
    data = dict(views=[dict(c2w=c2w, K=np.eye(3, dtype=np.float32), normal_path=str(tmp_path / "n.exr"), albedo_path=str(tmp_path / "a.exr"), mask_path=str(tmp_path / "m.exr"))],
                image_width=w, image_height=h, landmarks=None)

    class Log:
        def info(self, m):
            pass
        warning = error = info

    prepare.prepare_testbed_data(data, str(tmp_path / "p"), Log(), scaling_mode="none")
    nm = hostlib.png_read(tmp_path / "p" / "normals" / "00000.png")
    al = hostlib.png_read(tmp_path / "p" / "albedos" / "00000.png")
    assert nm.dtype == np.uint16 and al.dtype == np.uint16 and nm.shape == (h, w, 4)
    assert nm[5, 7].tolist() == [0, int(np.float32(0.625) * 65535), 65535, 65535]
    assert al[5, 7].tolist() == [65535, 32767, 0, 65535]
    assert nm[0, 0, 3] == 0 and nm[3, 7, 3] == 0 and al[19, 17, 3] == 65535 and al[20, 17, 3] == 0


# ------------------------------------------------------------------------------- loaders + prepare
def _write_rnb_dir(root, n=5, res=32, centre=(0.2, 0.1, -0.3), radius=0.5, with_albedo=True, gains=None):
    """cameras.npz scene of a sphere: world_mat = K [R|t] (4x4), scale_mat = identity."""
    cams = _ring_cameras(n, res, centre, dist=3.0)
    masks = _sphere_masks(cams, res, centre, radius)
    for sub in ("normal", "mask") + (("albedo",) if with_albedo else ()):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    mats = {}
    for i, (c, m) in enumerate(zip(cams, masks)):
        K = np.array([[c["fx"], 0, c["cx"]], [0, c["fy"], c["cy"]], [0, 0, 1.0]])
        R_w2c = c["R_cam2world"].T
        P = np.eye(4)
        P[:3, :4] = K @ np.concatenate([R_w2c, (-R_w2c @ c["center"])[:, None]], axis=1)
        mats["world_mat_%d" % i] = P
        mats["scale_mat_%d" % i] = np.eye(4)
        nm = np.full((res, res, 3), 32768, np.uint16)
        hostlib.png_write(os.path.join(root, "normal", "%03d.png" % i), nm)
        hostlib.png_write(os.path.join(root, "mask", "%03d.png" % i), (m * 255).astype(np.uint8))
        if with_albedo:
            g = 1.0 if gains is None else gains[i]
            hostlib.png_write(os.path.join(root, "albedo", "%03d.png" % i), np.full((res, res, 3), int(20000 * g), np.uint16))
    np.savez(os.path.join(root, "cameras.npz"), **mats)
    return cams, masks


def test_projection_decomposition():
    rng = np.random.default_rng(0)
    for _ in range(20):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        K = np.array([[rng.uniform(300, 900), rng.uniform(-2, 2), rng.uniform(100, 400)], [0, rng.uniform(300, 900), rng.uniform(100, 400)], [0, 0, 1.0]])
        centre = rng.normal(size=3) * 3
        P = (K @ np.concatenate([q, (-q @ centre)[:, None]], axis=1)) * rng.uniform(0.5, 2.0)
        intr, pose = dataloaders.load_K_Rt_from_P(P)
        np.testing.assert_allclose(intr[:3, :3], K, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(pose[:3, :3], q.T, atol=1e-6)
        np.testing.assert_allclose(pose[:3, 3], centre, atol=1e-5)
        assert intr.shape == (4, 4) and pose.dtype == np.float32


def test_rnb_loader_and_factory(tmp_path):
    cams, _ = _write_rnb_dir(str(tmp_path / "s"), n=4, res=24)
    for src in (str(tmp_path / "s"), str(tmp_path / "s" / "cameras.npz")):
        data = dataloaders.load_data(src)
        assert data["image_width"] == 24 and data["image_height"] == 24 and data["landmarks"] is None and len(data["views"]) == 4
        for v, c in zip(data["views"], cams):
            np.testing.assert_allclose(v["c2w"][:3, :3], c["R_cam2world"], atol=1e-5)
            np.testing.assert_allclose(v["c2w"][:3, 3], c["center"], atol=1e-4)
            assert abs(v["K"][0, 0] - c["fx"]) < 1e-3 and v["K"].dtype == np.float32
            assert v["normal_path"].endswith(".png") and os.path.exists(v["mask_path"]) and os.path.exists(v["albedo_path"])
    os.makedirs(tmp_path / "empty")
    with pytest.raises(FileNotFoundError):
        dataloaders.create_loader(str(tmp_path / "empty"))
    with pytest.raises(ValueError):
        dataloaders.create_loader(str(tmp_path / "scene.xyz"))


def test_sfm_json_loader(tmp_path):
    """AliceVision SfMData: string-valued fields, pose = cam2world rotation (row-major) + centre, y/z world flip,
    principal point stored as an offset from the image centre. (sfm_json_loader.py:26-117)"""
    sfm = {
        "views": [{"viewId": "10", "poseId": "10", "intrinsicId": "1", "path": "n/10.png"}, {"viewId": "11", "poseId": "11", "intrinsicId": "1", "path": "/abs/11.png"},
                  {"viewId": "12", "poseId": "99", "intrinsicId": "1", "path": "n/12.png"}],
        "intrinsics": [{"intrinsicId": "1", "width": "640", "height": "480", "focalLength": "18", "sensorWidth": "36", "principalPoint": ["4.5", "-2"]}],
        "poses": [{"poseId": "10", "pose": {"transform": {"rotation": ["1", "0", "0", "0", "1", "0", "0", "0", "1"], "center": ["1", "2", "3"]}}},
                  {"poseId": "11", "pose": {"transform": {"rotation": ["0", "-1", "0", "1", "0", "0", "0", "0", "1"], "center": ["0", "0", "-5"]}}}],
        "structure": [{"X": ["1", "1", "1"]}, {"X": ["2", "-3", "4"]}],
    }
    with open(tmp_path / "normals.sfm", "w") as f:
        json.dump(sfm, f)
    alb = dict(sfm, views=[{"viewId": "20", "poseId": "10", "intrinsicId": "1", "path": "a/10.png"}])
    with open(tmp_path / "albedos.sfm", "w") as f:
        json.dump(alb, f)
    os.makedirs(tmp_path / "masks")
    open(tmp_path / "masks" / "11.png", "wb").close()
    data = dataloaders.load_data(str(tmp_path / "normals.sfm"), albedo_sfm_path=str(tmp_path / "albedos.sfm"), mask_folder_path=str(tmp_path / "masks"))
    assert len(data["views"]) == 2 and data["image_width"] == 640 and data["image_height"] == 480  # the view with a missing pose is dropped
    v0, v1 = data["views"]
    assert v0["K"][0, 0] == 320.0 and v0["K"][0, 2] == 324.5 and v0["K"][1, 2] == 238.0
    np.testing.assert_array_equal(v0["c2w"], [[1, 0, 0, 1], [0, -1, 0, -2], [0, 0, -1, -3], [0, 0, 0, 1]])
    np.testing.assert_array_equal(v1["c2w"][:3, :3], [[0, -1, 0], [-1, 0, 0], [0, 0, -1]])
    assert v0["normal_path"] == str(tmp_path / "n" / "10.png") and v1["normal_path"] == "/abs/11.png"
    assert v0["albedo_path"] == str(tmp_path / "a" / "10.png") and v1["albedo_path"] is None
    assert v0["mask_path"] is None and v1["mask_path"] == str(tmp_path / "masks" / "11.png")
    np.testing.assert_array_equal(data["landmarks"], [[1, -1, -1], [2, 3, -4]])
    sfm["intrinsics"][0].pop("focalLength")
    sfm["intrinsics"][0]["pxFocalLength"] = ["700", "710"]
    cams, _ = dataloaders.parse_sfm_json(sfm)
    assert (cams[0]["fx"], cams[0]["fy"]) == (700.0, 710.0)


def test_prepare_albedo_alpha_stays_opaque(tmp_path):
    """The reference's own test (tests/test_prepare_albedo_alpha.py): 8-bit normal + 16-bit albedo -> the albedo's alpha
    must be full-scale at ITS bit depth."""
    h = w = 16
    hostlib.png_write(tmp_path / "n0.png", np.full((h, w, 3), 128, np.uint8))
    hostlib.png_write(tmp_path / "a0.png", np.full((h, w, 3), 30000, np.uint16))
    c2w = np.eye(4, dtype=np.float32)
    c2w[2, 3] = 3.0
    data = dict(views=[dict(c2w=c2w, K=np.eye(3, dtype=np.float32), normal_path=str(tmp_path / "n0.png"), albedo_path=str(tmp_path / "a0.png"), mask_path=None)],
                landmarks=None, image_width=w, image_height=h)
    out = prepare.prepare_testbed_data(data, str(tmp_path / "prepared"), Log(), scaling_mode="cameras")
    alb = hostlib.png_read(tmp_path / "prepared" / "albedos" / "00000.png")
    nrm = hostlib.png_read(tmp_path / "prepared" / "normals" / "00000.png")
    assert alb.shape == (h, w, 4) and alb.dtype == np.uint16 and int(alb[:, :, 3].max()) == 65535 and np.all(alb[:, :, :3] == 30000)
    assert nrm.shape == (h, w, 4) and nrm.dtype == np.uint8 and int(nrm[:, :, 3].min()) == 255
    assert out["n_frames"] == 1


def test_prepare_writes_the_scene_format(tmp_path):
    """transform.json keys / constants, n2w = inverse normalisation, scaled camera centres, mask -> alpha, white albedo
    when none is given, unreadable frames skipped. (prepare.py:116-257)"""
    centre, radius = (0.2, 0.1, -0.3), 0.5
    _write_rnb_dir(str(tmp_path / "s"), n=6, res=48, centre=centre, radius=radius, with_albedo=False)
    data = dataloaders.load_data(str(tmp_path / "s"))
    data["views"][3]["normal_path"] = str(tmp_path / "s" / "normal" / "gone.png")
    log = Log()
    out = prepare.prepare_testbed_data(data, str(tmp_path / "p"), log, scaling_mode="silhouettes", sphere_scale=1.0)
    assert out["n_frames"] == 5 and any("Normal not found" in l for l in log.lines)
    np.testing.assert_allclose(out["scene_center"], centre, atol=0.03)
    assert 1.0 / out["scale_factor"] == pytest.approx(radius * np.sqrt(1.5), rel=0.08)  # fg_area_ratio 1.5 => r_fit = r * sqrt(1.5)
    with open(tmp_path / "p" / "transform.json") as f:
        t = json.load(f)
    assert (t["w"], t["h"], t["aabb_scale"], t["scale"], t["offset"], t["from_na"]) == (48, 48, 1.0, 0.5, [0.5, 0.5, 0.5], True)
    np.testing.assert_allclose(np.array(t["n2w"]) @ out["scale_matrix"], np.eye(4), atol=1e-5)
    assert [f["normal_path"] for f in t["frames"]] == ["normals/%05d.png" % i for i in (0, 1, 2, 4, 5)]
    fr = t["frames"][0]
    np.testing.assert_allclose(np.array(fr["transform_matrix"])[:3, 3], out["scale_factor"] * (data["views"][0]["c2w"][:3, 3] - out["scene_center"]), rtol=1e-5)
    nrm = hostlib.png_read(tmp_path / "p" / fr["normal_path"])
    alb = hostlib.png_read(tmp_path / "p" / fr["albedo_path"])
    mask = hostlib.png_read(tmp_path / "s" / "mask" / "000.png")
    assert np.array_equal(nrm[:, :, 3] == 65535, mask > 125) and np.array_equal(alb[:, :, 3], nrm[:, :, 3])
    assert np.all(alb[:, :, :3] == 65535) and np.all(nrm[:, :, :3] == 32768)
    with pytest.raises(RuntimeError, match="No data for scaling"):
        prepare._compute_scaling(dict(views=[], landmarks=None), "pcd", 1.0, 20, Log())
    c, f, m = prepare._compute_scaling(data, "none", 1.0, 20, Log())
    assert f == 1.0 and np.array_equal(m, np.eye(4))


# ------------------------------------------------------------------------------- ray casting / meshes / albedo scaling
def _lattice_sphere(centre, radius, n):
    th = np.linspace(0, np.pi, n + 1)[1:-1]
    ph = np.linspace(0, 2 * np.pi, 2 * n, endpoint=False)
    v = [[0, 0, 1.0]] + [[np.sin(t) * np.cos(p), np.sin(t) * np.sin(p), np.cos(t)] for t in th for p in ph] + [[0, 0, -1.0]]
    m = 2 * n
    f = []
    for j in range(m):
        f.append([0, 1 + j, 1 + (j + 1) % m])
        f.append([len(v) - 1, 1 + (n - 2) * m + (j + 1) % m, 1 + (n - 2) * m + j])
    for i in range(n - 2):
        for j in range(m):
            a, b = 1 + i * m + j, 1 + i * m + (j + 1) % m
            c, d = a + m, b + m
            f += [[a, c, b], [b, c, d]]
    return np.asarray(centre) + radius * np.array(v), np.array(f)


def test_raycaster_against_brute_force():
    rng = np.random.default_rng(5)
    v, f = _lattice_sphere((0.1, 0.2, 0.3), 0.8, 12)
    v = v.astype(np.float32).astype(np.float64)  # the caster keeps float32 vertices
    rc = hostlib.MeshRayCaster(v, f)
    o = rng.normal(size=(400, 3)) * 2.5
    d = np.array([0.1, 0.2, 0.3]) + rng.normal(size=(400, 3)) * 0.7 - o  # aimed near the sphere: a mix of hits and misses
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t, tri = rc.first_hit(o, d)
    tv = v[f]
    e1, e2 = tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0]
    best = np.full(len(o), np.inf)
    for i in range(len(o)):
        p = np.cross(d[i], e2)
        det = np.einsum("ij,ij->i", e1, p)
        s = o[i] - tv[:, 0]
        u = np.einsum("ij,ij->i", s, p) / det
        q = np.cross(s, e1)
        w = (q @ d[i]) / det
        tt = np.einsum("ij,ij->i", e2, q) / det
        ok = (u >= 0) & (w >= 0) & (u + w <= 1) & (tt > 0)
        if ok.any():
            best[i] = tt[ok].min()
    assert np.array_equal(np.isfinite(best), tri >= 0) and 50 < (tri >= 0).sum() < 400
    np.testing.assert_allclose(t[tri >= 0], best[tri >= 0], rtol=1e-9)
    occl = rc.occluded(o, d, np.where(np.isfinite(best), best * 0.5, 10.0))
    assert not occl.any()
    assert np.array_equal(rc.occluded(o, d, np.full(len(o), 100.0)), tri >= 0)
    with pytest.raises(RuntimeError):
        hostlib.MeshRayCaster(v, f + len(v))


def test_mesh_split_and_orientation(tmp_path):
    v1, f1 = _lattice_sphere((0, 0, 0), 1.0, 10)
    v2, f2 = _lattice_sphere((5, 0, 0), 0.3, 6)
    rng = np.random.default_rng(1)
    f1 = f1.copy()
    flip = rng.random(len(f1)) < 0.4
    f1[flip] = f1[flip][:, ::-1]  # scrambled winding
    m = meshproc.Mesh(np.concatenate([v1, v2]), np.concatenate([f1, f2[:, ::-1] + len(v1)]))
    parts = m.split()
    assert sorted(len(p.vertices) for p in parts) == [len(v2), len(v1)]
    big = max(parts, key=lambda p: p.area)
    big.fix_normals()
    assert big.signed_volume == pytest.approx(4 / 3 * np.pi, rel=0.05)
    n = big.vertex_normals()
    assert np.all(np.einsum("ij,ij->i", n, big.vertices) > 0.9)
    m.fix_normals()
    assert m.signed_volume == pytest.approx(4 / 3 * np.pi * (1 + 0.027), rel=0.06)  # the inward-wound small sphere is flipped too
    meshproc.save_obj(str(tmp_path / "m.obj"), big)
    back = meshproc.load_obj(str(tmp_path / "m.obj"))
    np.testing.assert_allclose(back.vertices, big.vertices, atol=1e-7)
    assert np.array_equal(back.faces, big.faces)


def test_postprocess_mesh_keeps_largest_component(tmp_path):
    v1, f1 = _lattice_sphere((0, 0, 0), 1.0, 8)
    v2, f2 = _lattice_sphere((3, 0, 0), 0.2, 4)
    os.makedirs(tmp_path / "d" / "output")
    with open(tmp_path / "d" / "output" / "mesh_100.obj", "w") as f:  # the testbed's layout: v with colours, vn, f a//a
        for p in np.concatenate([v1, v2]):
            f.write("v %0.5f %0.5f %0.5f 0.500 0.250 1.000\n" % tuple(p))
        for p in np.concatenate([v1, v2]):
            f.write("vn 0 0 1\n")
        for a, b, c in np.concatenate([f1[:, ::-1], f2 + len(v1)]) + 1:
            f.write("f %d//%d %d//%d %d//%d\n" % (a, a, b, b, c, c))
    open(tmp_path / "d" / "output" / "mesh_100.json", "w").close()
    log = Log()
    pipeline.postprocess_mesh(str(tmp_path / "d"), str(tmp_path / "out" / "mesh.obj"), log)
    assert not os.path.exists(tmp_path / "d" / "output")
    m = meshproc.load_obj(str(tmp_path / "out" / "mesh.obj"))
    assert len(m.vertices) == len(v1) and m.signed_volume > 3.5 and m.colors is not None and np.allclose(m.colors, [0.5, 0.25, 1.0])
    assert any("Kept largest component" in l for l in log.lines)
    with pytest.raises(RuntimeError, match="No mesh files"):
        pipeline.postprocess_mesh(str(tmp_path / "d"), str(tmp_path / "out" / "mesh2.obj"), Log())


def test_albedo_scaling_recovers_per_view_gains(tmp_path):
    """Uniform-albedo sphere photographed with per-view gains g_i: the recovered factors are (1/g_i) up to the common
    normalisation (mean 1), and applying them equalises the views."""
    centre, radius, res, n = np.array([0.1, -0.2, 0.3]), 0.6, 64, 6
    gains = np.array([1.0, 1.3, 0.7, 1.1, 0.9, 1.6])
    cams = _ring_cameras(n, res, centre, dist=3.0)
    masks = _sphere_masks(cams, res, centre, radius * 0.98)
    os.makedirs(tmp_path / "albedos")
    frames = []
    for i, (c, m) in enumerate(zip(cams, masks)):
        rgba = np.zeros((res, res, 4), np.uint16)
        rgba[:, :, :3] = (np.array([20000, 15000, 10000]) * gains[i]).astype(np.uint16)
        rgba[:, :, 3] = (m * 65535).astype(np.uint16)
        hostlib.png_write(tmp_path / "albedos" / ("%05d.png" % i), rgba)
        c2w = np.eye(4)
        c2w[:3, :3], c2w[:3, 3] = c["R_cam2world"], c["center"]
        frames.append(dict(albedo_path="albedos/%05d.png" % i, normal_path="normals/%05d.png" % i, transform_matrix=c2w.tolist(),
                           intrinsic_matrix=[[c["fx"], 0, c["cx"]], [0, c["fy"], c["cy"]], [0, 0, 1]]))
    # cameras are stored in normalised space; the mesh lives in world space = n2w * normalised
    n2w = np.array([[2.0, 0, 0, 1], [0, 2.0, 0, -1], [0, 0, 2.0, 0.5], [0, 0, 0, 1]])
    with open(tmp_path / "transform.json", "w") as f:
        json.dump(dict(w=res, h=res, n2w=n2w.tolist(), frames=frames), f)
    v, fa = _lattice_sphere(n2w[:3, :3] @ centre + n2w[:3, 3], 2.0 * radius, 40)
    meshproc.save_obj(str(tmp_path / "mesh.obj"), meshproc.Mesh(v, fa))
    np.random.seed(0)
    ratios = albedo_scaling.compute_albedo_scale_ratios(str(tmp_path / "albedos"), str(tmp_path / "transform.json"), str(tmp_path / "mesh.obj"), n_samples=300, logger=Log())
    assert ratios.shape == (n, 3)
    np.testing.assert_allclose(ratios.mean(axis=0), 1.0, atol=1e-12)
    want = (1.0 / gains) / (1.0 / gains).mean()
    np.testing.assert_allclose(ratios, np.repeat(want[:, None], 3, axis=1), rtol=2e-3)
    albedo_scaling.scale_and_save_albedos(str(tmp_path / "albedos"), str(tmp_path / "scaled"), ratios, logger=Log())
    means = []
    for i in range(n):
        img = hostlib.png_read(tmp_path / "scaled" / ("%05d.png" % i))
        assert img.dtype == np.uint16 and np.array_equal(img[:, :, 3], (masks[i] * 65535).astype(np.uint16))
        means.append(img[res // 2, res // 2, :3].astype(float))
    means = np.array(means)
    assert np.all(np.abs(means / means.mean(axis=0) - 1) < 3e-3)
    k, r, c = albedo_scaling.load_cameras(str(tmp_path / "transform.json"), ["%05d.png" % i for i in range(n)])
    np.testing.assert_allclose(c[2][:, 0], n2w[:3, :3] @ cams[2]["center"] + n2w[:3, 3], rtol=1e-5)
    with pytest.raises(RuntimeError, match="No frame for albedo image"):
        albedo_scaling.load_cameras(str(tmp_path / "transform.json"), ["zzz.png"])


# ------------------------------------------------------------------------------- the whole pipeline on the CPU checker
def test_full_pipeline_end_to_end_cpu(small_install, tmp_path):
    """cameras.npz directory -> prepare -> testbed stage 1 -> snapshot -> testbed stage 2 (resume, opti-lights, mesh) ->
    post-processed mesh.obj, with the real testbed source running on the CPU checker and a small network."""
    _write_rnb_dir(str(tmp_path / "in"), n=4, res=40, centre=(0.0, 0.0, 0.0), radius=0.5, with_albedo=False)
    log = Log()
    mesh_path = pipeline.run_full_pipeline(str(tmp_path / "in"), str(small_install / "build" / "testbed"), str(tmp_path / "out"), max_steps=6, mesh_resolution=32,
                                           scaling_mode="silhouettes", logger=log)
    assert mesh_path == str(tmp_path / "out" / "mesh.obj") and os.path.exists(mesh_path)
    assert os.path.exists(tmp_path / "out" / "prepared_data" / "transform.json") and not os.path.exists(tmp_path / "out" / "prepared_data" / "output")
    cmds = [l for l in log.lines if " command: " in l]
    assert len(cmds) == 2 and "--maxiter 4 " in cmds[0] and "--maxiter 6 " in cmds[1] and "snapshot_4.msgpack" in cmds[1]
    assert any("Loaded snapshot succeed" in l for l in log.lines)
    m = meshproc.load_obj(mesh_path)
    assert len(m.faces) > 100 and m.signed_volume > 0


def test_sfm_json_loader_against_the_references_outputs(tmp_path):
    """tests/golden/sfm_loader_vectors.json: the REFERENCE's `parse_sfm_json` and `SfmJsonDataLoader.load` (rnb_neus2/dataloaders/sfm_json_loader.py) run on four seeded
    SfMData documents (pixel / millimetre focal lengths, the 36 mm default, a view without a pose, absolute and relative image paths, landmarks, an albedo document and a
    mask folder) -- every camera field, matrix element and resolved path equal (the matrices are float32 casts of float64 products: bit for bit)."""
    with open(os.path.join(ROOT, "tests", "golden", "sfm_loader_vectors.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) == 4
    for ci, case in enumerate(cases):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cams, landmarks = dataloaders.parse_sfm_json(case["document"], case["sfm_dir"])
        want = case["parse"]["cameras"]
        assert len(cams) == len(want), ci
        for c, w in zip(cams, want):
            assert set(c) == set(w), (ci, set(c) ^ set(w))
            for k, v in w.items():
                if isinstance(v, list):
                    np.testing.assert_array_equal(np.asarray(c[k], dtype=np.float64), np.asarray(v, dtype=np.float64), err_msg="%d %s" % (ci, k))
                else:
                    assert c[k] == v, (ci, k, c[k], v)
        if case["parse"]["landmarks"] is None:
            assert landmarks is None
        else:
            np.testing.assert_array_equal(np.asarray(landmarks, dtype=np.float64), np.asarray(case["parse"]["landmarks"]))
        d = tmp_path / ("case%d" % ci)
        os.makedirs(d / "masks")
        with open(d / "normals.sfm", "w") as f:
            json.dump(case["document"], f)
        with open(d / "albedos.sfm", "w") as f:
            json.dump(case["load"]["albedo_document"], f)
        for name in case["load"]["mask_files"]:
            open(d / "masks" / name, "wb").close()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = dataloaders.SfmJsonDataLoader(str(d / "normals.sfm"), albedo_sfm_path=str(d / "albedos.sfm"), mask_folder_path=str(d / "masks")).load()
        L = case["load"]
        assert out["image_width"] == L["image_width"] and out["image_height"] == L["image_height"] and out["scale_mat"] is None and L["scale_mat"] is None
        assert len(out["views"]) == len(L["views"])
        for v, w in zip(out["views"], L["views"]):
            assert str(v["c2w"].dtype) == w["c2w_dtype"] == "float32"
            np.testing.assert_array_equal(v["c2w"].astype(np.float64), np.asarray(w["c2w"]))
            np.testing.assert_array_equal(np.asarray(v["K"], dtype=np.float64), np.asarray(w["K"]))
            for k in ("normal_path", "albedo_path", "mask_path"):
                assert v[k] == (None if w[k] is None else w[k].replace("<DIR>", str(d))), (ci, k, v[k], w[k])
            assert v["pose_id"] == w["pose_id"]
        if L["landmarks"] is None:
            assert out["landmarks"] is None
        else:
            np.testing.assert_array_equal(np.asarray(out["landmarks"], dtype=np.float64), np.asarray(L["landmarks"]))


def test_albedo_scaling_camera_loader_against_the_references_outputs(tmp_path):
    """tests/golden/albedo_camera_vectors.json: the REFERENCE's `load_cameras` (rnb_neus2/albedo_scaling.py:128-211) on six seeded transform.json documents -- frames matched
    to the albedo files by stem and returned in THEIR order, `intrinsic_matrix` vs per-frame vs global vs default focal lengths and principal points (with the reference's `or`
    fall-backs), `n2w` back to world space -- K, R and centres equal bit for bit, same dtypes, same exceptions."""
    from rnb_neus2_amd import albedo_scaling
    with open(os.path.join(ROOT, "tests", "golden", "albedo_camera_vectors.json")) as f:
        fix = json.load(f)
    assert len(fix["cases"]) == 6
    for ci, case in enumerate(fix["cases"]):
        p = tmp_path / ("transform%d.json" % ci)
        p.write_text(json.dumps(case["document"]))
        K, R, C = albedo_scaling.load_cameras(str(p), case["albedo_images"])
        assert [str(K.dtype), str(R.dtype), str(C.dtype)] == case["dtypes"], ci
        np.testing.assert_array_equal(K.astype(np.float64), np.asarray(case["K"]), err_msg="K %d" % ci)
        np.testing.assert_array_equal(R.astype(np.float64), np.asarray(case["R_c2w"]), err_msg="R %d" % ci)
        np.testing.assert_array_equal(C.astype(np.float64), np.asarray(case["centers"]), err_msg="C %d" % ci)
    p = tmp_path / "t.json"
    p.write_text(json.dumps(fix["cases"][0]["document"]))
    assert fix["missing_frame_raises"] == "RuntimeError" and fix["unknown_suffix_raises"] == "ValueError"
    with pytest.raises(RuntimeError):
        albedo_scaling.load_cameras(str(p), ["/x/00099.png"])
    with pytest.raises(ValueError):
        albedo_scaling.load_cameras("/x/cameras.bin", ["a.png"])


def test_prepare_scaling_mode_cascade_against_the_references_outputs():
    """tests/golden/prepare_scaling_vectors.json: the REFERENCE's `_compute_scaling` (rnb_neus2/prepare.py:44-113) on thirteen seeded loader dicts without mask files -- "none",
    landmarks, camera centres, "auto" falling through the silhouette branch to landmarks / to the cameras (no or empty landmarks), and every mode that finds no data (or is
    unknown) raising RuntimeError with the reference's message: centre, factor and matrix equal to float32 rounding, same dtypes, the same info lines in the same order."""
    from rnb_neus2_amd import prepare

    class Log:
        def __init__(self):
            self.lines = []

        def info(self, msg):
            self.lines.append(str(msg))

        warning = info

    with open(os.path.join(ROOT, "tests", "golden", "prepare_scaling_vectors.json")) as f:
        fix = json.load(f)
    assert len(fix["cases"]) == 13 and sum(c["raises"] is not None for c in fix["cases"]) == 6
    for ci, case in enumerate(fix["cases"]):
        data = {"views": [{"c2w": np.asarray(m, np.float32), "K": np.array([[900.0, 0, 320.0], [0, 905.0, 240.0], [0, 0, 1]], np.float32), "mask_path": "",
                           "normal_path": "", "albedo_path": ""} for m in case["views_c2w"]]}
        if case["landmarks"] is not None:
            data["landmarks"] = np.asarray(case["landmarks"], np.float32).reshape(-1, 3)
        log = Log()
        if case["raises"] is not None:
            assert case["raises"] == "RuntimeError"
            with pytest.raises(RuntimeError) as ei:
                prepare._compute_scaling(data, case["mode"], case["sphere_scale"], case["margin_px"], log)
            assert str(ei.value) == case["message"], ci
            assert log.lines == case["log"], ci
            continue
        center, factor, matrix = prepare._compute_scaling(data, case["mode"], case["sphere_scale"], case["margin_px"], log)
        assert [str(np.asarray(center).dtype), str(np.asarray(matrix).dtype)] == case["dtypes"], (ci, case["mode"])
        np.testing.assert_allclose(np.asarray(center, np.float64), np.asarray(case["center"]), rtol=2e-6, atol=1e-6, err_msg="center %d" % ci)
        assert abs(float(factor) - case["factor"]) <= 2e-6 * abs(case["factor"]), (ci, factor, case["factor"])
        np.testing.assert_allclose(np.asarray(matrix, np.float64), np.asarray(case["matrix"]), rtol=2e-6, atol=1e-6, err_msg="matrix %d" % ci)
        assert len(log.lines) == len(case["log"]), (ci, log.lines, case["log"])
        for mine, ref in zip(log.lines, case["log"]):
            if ref.startswith(("Scene center:", "Scale factor:")):  # numbers: compared above; the line's shape here
                assert mine.split(":")[0] == ref.split(":")[0], (ci, mine, ref)
            else:
                assert mine == ref, (ci, mine, ref)
