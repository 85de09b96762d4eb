"""The `testbed` command line (rnb-neus2_amd/host/testbed_main.cpp, mirror of the reference's src/main.cu) on CPU:
the same source is built here against the oracle library (`-include oracle/orc_prefix.h`), so the flag parser, the scene
loader, the training loop, the snapshot writer/reader and the mesh export are exercised end to end without a GPU.
The GPU build of the same file is covered by tests/test_gpu_cli.py::test_testbed_cli_gpu."""
import json
import os
import subprocess

import msgpack
import numpy as np
import pytest

from rnb_neus2_amd import synthetic
from tests import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "rnb-neus2_amd", "host")

from tests.conftest import SMALL_CFG  # noqa: E402

# what SMALL_CFG means for rnb_config (per_level_scale as Testbed::reset_network derives it, src/testbed.cu:2296-2305)
SMALL_KW = dict(n_levels=4, log2_hashmap_size=12, base_resolution=16, per_level_scale=float(np.exp(np.float32(np.log(np.float32(64.0 / 16.0))) / np.float32(3))),
                target_batch_size=4096, mask_loss_weight=1.0, apply_no_albedo=1)


def run(install, *args, **kw):
    return subprocess.run([str(install / "build" / "testbed"), *map(str, args)], capture_output=True, text=True, timeout=600, **kw)


def dump(install, path):
    out = subprocess.run([str(install / "build" / "dump_dataset"), str(path)], capture_output=True, text=True, check=True).stdout
    return json.loads(out.strip().splitlines()[-1])


# ---------------------------------------------------------------- flags / exit codes (src/main.cu:73-258, 300-347)
def test_version_and_help(install):
    r = run(install, "--version")
    assert r.returncode == 0 and "version" in r.stdout
    r = run(install, "-h")
    assert r.returncode == 0
    for flag in ("--scene", "--maxiter", "--mask-weight", "--save-mesh", "--save-snapshot", "--resolution", "--snapshot", "--opti-lights", "--no-albedo",
                 "--free-memory", "--lone", "--supernormal", "--no-rgbplus", "--relu", "--bce", "--disable-snap-to-center", "--no-gui", "--no-train",
                 "--save-each", "--fractional-training", "--width", "--height", "--config", "--network"):
        assert flag in r.stdout, flag


def test_every_flag_of_the_references_command_line(install):
    """tests/golden/cli_flags.json: the `args` declarations of the reference's main (src/main.cu:73-258), parsed from its text by make_cli_fixture.py -- 25 of them. `testbed -h`
    lists every one in the reference's order with its short and long names, its placeholder and (the scene flag's wording aside) its help text; a value flag refuses to go
    without its value, a plain flag refuses one; the only flags the reference does not have are --accumulate and --deterministic."""
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cli_flags.json")) as f:
        flags = json.load(f)["flags"]
    assert len(flags) == 25
    text = run(install, "-h").stdout
    pos = -1
    for fl in flags:
        if fl["kind"] == "ValueFlag":
            head = ", ".join(["-%s[%s]" % (c, fl["placeholder"]) for c in fl["short"]] + ["--%s=[%s]" % (n, fl["placeholder"]) for n in fl["long"]])
        else:
            head = ", ".join(["-%s" % c for c in fl["short"]] + ["--%s" % n for n in fl["long"]])
        at = text.find("      " + head + "\n")
        assert at > pos, (head, at, pos)  # present, in the reference's order
        pos = at
        if fl["long"] != ["scene"]:  # (the reference's text lists dataset kinds of its other modes)
            assert fl["help"] in text[at:at + 400], (head, fl["help"])
    listed = [l.strip() for l in text.splitlines() if l.startswith("      -")]
    assert len(listed) == 27 and listed[-2].startswith("--accumulate=") and listed[-1].startswith("--deterministic")
    for fl in flags:
        name = "--" + fl["long"][0]
        if fl["kind"] == "ValueFlag":
            r = run(install, name)
            assert r.returncode == 255 and "OPTIONS" in r.stderr, name  # missing value: the usage text, `return -1`
        elif fl["kind"] == "Flag" and name != "--version":
            r = run(install, name + "=1")
            assert r.returncode == 255, name


def test_parse_errors_exit_minus_one(install):
    for bad in (["--does-not-exist"], ["--maxiter"], ["--maxiter", "many"], ["--no-gui=1"], ["stray"]):
        r = run(install, *bad)
        assert r.returncode == 255, (bad, r.returncode)  # `return -1`
        assert "OPTIONS" in r.stderr


def test_missing_paths_exit_one(install, tmp_path):
    r = run(install, "--scene", tmp_path / "nope", "--no-gui")
    assert r.returncode == 1 and "does not exist" in r.stderr
    views, normals, albedos = synthetic.make_scene(2, 16, 28.0)
    synthetic.write_scene(str(tmp_path / "s"), views, normals, albedos)
    r = run(install, "--scene", tmp_path / "s", "--no-gui", "--snapshot", tmp_path / "missing.msgpack")
    assert r.returncode == 1 and "Snapshot path" in r.stderr
    r = run(install, "--scene", tmp_path / "s", "--no-gui", "--config", "missing.json")
    assert r.returncode == 1 and "Network config path" in r.stderr
    r = run(install, "--scene", tmp_path / "s", "--no-gui", "--config", "small.json", "--maxiter", "5", "--fractional-training", "9")
    assert r.returncode == 1 and "lower than max-iter" in r.stderr
    r = run(install, "--scene", tmp_path / "s", "--no-gui", "--config", "small.json", "--fractional-training", "9")
    assert r.returncode == 1 and "works with max-iter" in r.stderr


# ---------------------------------------------------------------- scene loader (src/nerf_loader.cu, nerf_loader.h:180-201)
def test_loader_roundtrips_written_scene(install, tmp_path):
    views, normals, albedos = synthetic.make_scene(3, 24, 42.0)
    synthetic.write_scene(str(tmp_path / "s"), views, normals, albedos, scale=0.5, offset=(0.5, 0.25, 0.125), n2w=[[2, 0, 0, 1], [0, 2, 0, 2], [0, 0, 2, 3], [0, 0, 0, 1]])
    d = dump(install, tmp_path / "s")
    assert d["from_na"] == 1 and d["scale"] == 0.5 and d["offset"] == [0.5, 0.25, 0.125]
    assert d["n2w_s"] == 2 and d["n2w_t"] == [1, 2, 3]

    def fnv(a):
        h = 1469598103934665603
        for p in np.asarray(a, np.uint16).ravel().tolist():
            h = ((h ^ p) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return f"{h:016x}"

    for v, nm, al, got in zip(views, normals, albedos, d["views"]):
        assert (got["width"], got["height"]) == (24, 24)
        assert got["focal_length"] == [42.0, 42.0] and got["principal_point"] == [0.5, 0.5]
        np.testing.assert_allclose(np.array(got["xform"]).reshape(3, 4), np.asarray(v["xform"]).reshape(3, 4), atol=2e-6)
        assert got["normal_fnv"] == fnv(nm) and got["albedo_fnv"] == fnv(al)  # PNG16 decode is lossless


def _write_json_scene(path, meta, n=1, res=4):
    os.makedirs(path, exist_ok=True)
    px = np.full((res, res, 4), 65535, np.uint16)
    synthetic.write_png16(os.path.join(path, "n.png"), px)
    synthetic.write_png16(os.path.join(path, "a.png"), px)
    with open(os.path.join(path, "transform.json"), "w") as f:
        json.dump(meta, f)


def test_loader_axis_conventions(install, tmp_path):
    """nerf_matrix_to_ngp: flip y/z columns, scale+offset the position, then from_na un-flips / default cycles rows yzx."""
    M = [[1, 2, 3, 4], [5, 6, 7, 8], [9, 10, 11, 12], [0, 0, 0, 1]]
    K = [[10, 0, 2, 0], [0, 20, 1, 0], [0, 0, 1, 0], [0, 0, 0, 1]]
    frame = dict(normal_path="n", albedo_path="a.png", transform_matrix=M, intrinsic_matrix=K)  # extension-less path -> ".png"
    _write_json_scene(str(tmp_path / "na"), dict(from_na=False, w=4, h=4, scale=2.0, offset=[0.5, 1.0, 1.5], frames=[frame]))
    d = dump(install, tmp_path / "na")  # the key's presence selects from_na, whatever its value
    assert d["from_na"] == 1
    np.testing.assert_array_equal(np.array(d["views"][0]["xform"]).reshape(3, 4), [[1, 2, 3, 8.5], [5, 6, 7, 17], [9, 10, 11, 25.5]])
    assert d["views"][0]["focal_length"] == [10, 20] and d["views"][0]["principal_point"] == [0.5, 0.25]
    _write_json_scene(str(tmp_path / "ngp"), dict(w=4, h=4, frames=[frame]))
    d = dump(install, tmp_path / "ngp")  # defaults: scale 0.33, offset 0.5; rows cycled xyz <- yzx
    f32 = np.float32
    want = np.array([[5, -6, -7, f32(8) * f32(0.33) + f32(0.5)], [9, -10, -11, f32(12) * f32(0.33) + f32(0.5)], [1, -2, -3, f32(4) * f32(0.33) + f32(0.5)]], np.float32)
    np.testing.assert_array_equal(np.array(d["views"][0]["xform"], np.float32).reshape(3, 4), want)
    assert d["from_na"] == 0 and abs(d["scale"] - 0.33) < 1e-7
    _write_json_scene(str(tmp_path / "aabb"), dict(from_na=True, w=4, h=4, aabb=[[-1, -2, -3], [3, 0, -1]], offset=0.25, frames=[frame]))
    d = dump(install, tmp_path / "aabb")  # "aabb" overrides scale/offset: scale = 1/longest side, centre -> 0.5
    assert d["scale"] == 0.25 and d["offset"] == [0.25, 0.75, 1.0]


def test_loader_axis_conventions_against_the_references_own_function(install, tmp_path):
    """tests/golden/float_fixtures.json `axes_*`: the REFERENCE's NerfDataset::nerf_matrix_to_ngp (nerf_loader.h:180-201, compiled from its file by
    make_float_fixtures.py) on 36 seeded camera matrices -- default (rows cycled yzx), from_na, Mitsuba; scale and offset of the position. The same matrices as frames of
    a transform.json through the scene loader (host/dataset.hpp): every xform bit for bit."""
    from tests import float_fixture_cases
    v = np.array(float_fixture_cases.load()["axes_mode_scale_offset3_matrix12_ngp12"], dtype=np.uint32).reshape(-1, 29)
    assert len(v) == 36 and sorted(set(v[:, 0].tolist())) == [0, 1, 2]
    K = [[10, 0, 2, 0], [0, 20, 1, 0], [0, 0, 1, 0], [0, 0, 0, 1]]
    for k, row in enumerate(v):
        mode = int(row[0])
        fl = row[1:].view(np.float32).astype(np.float64)
        scale, offset, m, want = float(fl[0]), fl[1:4].tolist(), fl[4:16].reshape(3, 4), row[17:29].view(np.float32).reshape(3, 4)
        frame = dict(normal_path="n", albedo_path="a.png", transform_matrix=m.tolist() + [[0, 0, 0, 1]], intrinsic_matrix=K)
        meta = dict(w=4, h=4, frames=[frame])
        if mode == 1:
            meta.update(from_na=True, scale=scale, offset=offset)
        elif mode == 2:
            meta.update(normal_mts_args="-", frames=[frame])  # a Mitsuba scene: scale 0.66, offset 0.25 * scale (nerf_loader.cu:387-402) -- the fixture's values
            assert abs(scale - 0.66) < 1e-7
        else:
            meta.update(scale=scale, offset=offset)
        _write_json_scene(str(tmp_path / ("s%d" % k)), meta)
        d = dump(install, tmp_path / ("s%d" % k))
        assert (d["from_na"], d["from_mitsuba"]) == (int(mode == 1), int(mode == 2)), k
        got = np.array(d["views"][0]["xform"], np.float32).reshape(3, 4)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, mode, got, want)


def test_loader_errors(install, tmp_path):
    os.makedirs(tmp_path / "empty")
    r = subprocess.run([str(install / "build" / "dump_dataset"), str(tmp_path / "empty")], capture_output=True, text=True)
    assert r.returncode == 1
    _write_json_scene(str(tmp_path / "noframes"), dict(from_na=True, w=4, h=4, frames=[]))
    r = subprocess.run([str(install / "build" / "dump_dataset"), str(tmp_path / "noframes")], capture_output=True, text=True)
    assert r.returncode == 1 and "No training images" in r.stderr


# ---------------------------------------------------------------- training loop, snapshot, mesh
@pytest.fixture(scope="session")
def trained(install, tmp_path_factory):
    scene = tmp_path_factory.mktemp("scene")
    views, normals, albedos = synthetic.make_scene(4, 48, 84.0)
    synthetic.write_scene(str(scene), views, normals, albedos)  # scale 1, offset 0: the loader reproduces the poses exactly
    r = run(install, "--scene", str(scene) + "/", "--maxiter", 4, "--no-gui", "--mask-weight", 1.0, "--config", "small.json", "--no-albedo",
            "--save-snapshot", "--save-mesh", "--resolution", 40)
    assert r.returncode == 0, r.stderr
    return dict(scene=scene, out=r.stdout, data=(views, normals, albedos))


def test_outputs_and_layout(trained):
    out = trained["scene"] / "output"
    assert (out / "log.txt").exists() and (out / "mesh").is_dir() and (out / "images").is_dir()
    assert (out / "mesh_4.obj").exists() and (out / "snapshot_4.msgpack").exists()
    assert "Number of iterations : 4" in trained["out"] and "Saving Snapshot !" in trained["out"]


def test_snapshot_matches_library_state(trained):
    """The CLI's loader + loop + snapshot writer against the same 4 steps driven through the Python binding: bit-exact."""
    views, normals, albedos = trained["data"]
    ctx = oracle_lib.context(**SMALL_KW)
    ctx.init_params()
    ctx.set_dataset(views, normals, albedos)
    for _ in range(4):
        st = ctx.train_step()
    with open(trained["scene"] / "output" / "snapshot_4.msgpack", "rb") as f:
        root = msgpack.unpackb(f.read(), raw=False)
    snap = root["snapshot"]
    assert snap["training_step"] == 4 and snap["density_grid_size"] == 128 and snap["nerf"]["aabb_scale"] == 1
    assert snap["n_params"] == ctx.n_params
    ema = np.frombuffer(snap["params_binary"], np.uint16)
    np.testing.assert_array_equal(ema, ctx.get("PARAMS_EMA").view(np.uint16))
    grid = np.frombuffer(snap["density_grid_binary"], np.float16)
    np.testing.assert_array_equal(grid, ctx.get("DENSITY_GRID").astype(np.float16))
    assert snap["nerf"]["rgb"]["rays_per_batch"] == st.next_rays_per_batch
    assert snap["nerf"]["rgb"]["measured_batch_size"] == st.measured_batch_size
    assert snap["nerf"]["rgb"]["measured_batch_size_before_compaction"] == st.measured_batch_size_before_compaction
    assert abs(snap["loss"] - st.loss) <= 1e-6 * abs(st.loss)
    assert root["encoding"]["n_levels"] == 4 and root["encoding"]["log2_hashmap_size"] == 12


def test_resume_from_snapshot(install, trained):
    """--snapshot restores weights (master = EMA), occupancy grid, step and the ray controller, then trains on
    (Testbed::load_snapshot, src/testbed.cu:3333-3390)."""
    scene = trained["scene"]
    r = run(install, "--scene", scene, "--maxiter", 6, "--no-gui", "--mask-weight", 1.0, "--no-albedo", "--save-snapshot",
            "--snapshot", scene / "output" / "snapshot_4.msgpack")
    assert r.returncode == 0, r.stderr
    assert "Loaded snapshot succeed" in r.stdout
    with open(scene / "output" / "snapshot_6.msgpack", "rb") as f:
        snap2 = msgpack.unpackb(f.read(), raw=False)["snapshot"]
    with open(scene / "output" / "snapshot_4.msgpack", "rb") as f:
        snap1 = msgpack.unpackb(f.read(), raw=False)["snapshot"]
    assert snap2["training_step"] == 6
    views, normals, albedos = trained["data"]
    ctx = oracle_lib.context(**SMALL_KW)
    ctx.init_params()
    ctx.set_dataset(views, normals, albedos)
    ctx.set_params(np.frombuffer(snap1["params_binary"], np.float16).astype(np.float32))
    ctx.put("DENSITY_GRID", np.frombuffer(snap1["density_grid_binary"], np.float16).astype(np.float32))
    ctx.update_density_bitfield()
    ctx.set_controller(4, snap1["nerf"]["rgb"]["rays_per_batch"], snap1["nerf"]["rgb"]["measured_batch_size_before_compaction"], 0)
    for _ in range(2):
        ctx.train_step()
    np.testing.assert_array_equal(np.frombuffer(snap2["params_binary"], np.uint16), ctx.get("PARAMS_EMA").view(np.uint16))


def test_mesh_obj(trained):
    """OBJ layout of save_mesh (src/marching_cubes.cu:922-981): `v x y z r g b`, `vn`, `f a//a b//b c//c`; after 4 steps the
    surface is still the geometric-initialisation sphere around the scene centre."""
    v, vn, f = [], [], []
    with open(trained["scene"] / "output" / "mesh_4.obj") as fh:
        for line in fh:
            t = line.split()
            if t[0] == "v":
                assert len(t) == 7
                v.append([float(x) for x in t[1:]])
            elif t[0] == "vn":
                vn.append([float(x) for x in t[1:]])
            elif t[0] == "f":
                idx = [tuple(int(q) for q in c.split("//")) for c in t[1:]]
                assert all(a == b for a, b in idx)
                f.append([a for a, _ in idx])
    v, vn, f = np.array(v), np.array(vn), np.array(f)
    assert len(v) > 100 and len(vn) == len(v) and f.min() == 1 and f.max() == len(v)
    assert np.all((v[:, 3:] >= 0) & (v[:, 3:] <= 1))
    r = np.linalg.norm(v[:, :3] - 0.5, axis=1)
    assert r.std() < 0.05 and 0.1 < r.mean() < 0.6
    outward = np.sum(vn * (v[:, :3] - 0.5), axis=1)
    # the reference's vn records are (pb - pa) x (pa - pc) summed over the 1-ring (src/marching_cubes.cu:354-356) and written as they are
    # (:933-936): with its table and "corner bit = sdf > 0" they point down the SDF gradient, into the object
    assert (outward < 0).mean() > 0.95
    # from_na datasets (invert_normals, src/testbed.cu:376-379; marching_cubes.cu:966-975): faces counter-clockwise seen from outside
    tri = v[np.array(f) - 1][:, :, :3]
    geo = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    assert (np.sum(geo * (tri.mean(axis=1) - 0.5), axis=1) > 0).mean() > 0.95
    edges = {}
    for a, b, c in f:
        for e in ((a, b), (b, c), (c, a)):
            edges[tuple(sorted(e))] = edges.get(tuple(sorted(e)), 0) + 1
    assert all(n == 2 for n in edges.values())  # closed surface


def test_progress_line_and_fractional_training(install, tmp_path):
    """`iteration=<step> loss=<float>` every 100 steps (src/main.cu:444-451); --fractional-training switches the colour
    branch on at the given step (src/testbed.cu:1886-1895): the colour MLP stays at its initial values before it."""
    views, normals, albedos = synthetic.make_scene(2, 16, 28.0)
    synthetic.write_scene(str(tmp_path / "s"), views, normals, albedos)
    cfg = json.loads(json.dumps(SMALL_CFG))
    cfg["hyperparams"]["batch_size"] = 256
    with open(install / "configs" / "nerf" / "tiny.json", "w") as f:
        json.dump(cfg, f)
    r = run(install, "--scene", tmp_path / "s", "--maxiter", 101, "--no-gui", "--config", "tiny.json", "--fractional-training", 60, "--save-snapshot")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("iteration=")]
    assert [l.split()[0] for l in lines] == ["iteration=100"]  # (the 200-step run: test_config_1_single_256x256_view_200_steps)
    assert all(np.isfinite(float(l.split("loss=")[1])) for l in lines)


def test_snapshot_carries_the_whole_network_config(install, tmp_path):
    """The reference serialises m_network_config with the snapshot (src/testbed.cu:3280-3313) and resets the network from it on
    load (:3352-3357): a run that was configured with non-default optimizer hyper-parameters must resume with THEM, not with
    the library defaults. Stage 1 here uses lr = 3e-3, beta2 = 0.95, EMA decay 0.9, l2_reg 1e-5; the snapshot must hold them,
    and two more steps from the snapshot must equal the same two steps driven through the binding with those values
    (and differ from a resume with the defaults)."""
    import copy
    cfg = copy.deepcopy(SMALL_CFG)
    cfg["optimizer"]["decay"] = 0.9
    cfg["optimizer"]["nested"]["decay_start"] = 3
    cfg["optimizer"]["nested"]["decay_interval"] = 2
    cfg["optimizer"]["nested"]["decay_base"] = 0.5
    cfg["optimizer"]["nested"]["nested"].update(learning_rate=3e-3, beta2=0.95, l2_reg=1e-5)
    cfg["loss"] = {"otype": "Huber"}  # blocks the hot path does not read travel too
    with open(install / "configs" / "nerf" / "custom_opt.json", "w") as f:
        json.dump(cfg, f)
    scene = tmp_path / "s"
    views, normals, albedos = synthetic.make_scene(4, 48, 84.0)
    synthetic.write_scene(str(scene), views, normals, albedos)
    r = run(install, "--scene", scene, "--maxiter", 4, "--no-gui", "--mask-weight", 1.0, "--config", "custom_opt.json", "--no-albedo", "--save-snapshot")
    assert r.returncode == 0, r.stderr
    with open(scene / "output" / "snapshot_4.msgpack", "rb") as f:
        root = msgpack.unpackb(f.read(), raw=False)
    adam = root["optimizer"]["nested"]["nested"]
    assert root["optimizer"]["decay"] == pytest.approx(0.9) and root["optimizer"]["nested"]["decay_base"] == pytest.approx(0.5)
    assert adam["learning_rate"] == pytest.approx(3e-3) and adam["beta2"] == pytest.approx(0.95) and adam["l2_reg"] == pytest.approx(1e-5)
    assert root["loss"]["otype"] == "Huber" and root["hyperparams"]["batch_size"] == 4096
    r = run(install, "--scene", scene, "--maxiter", 6, "--no-gui", "--mask-weight", 1.0, "--no-albedo", "--save-snapshot", "--snapshot", scene / "output" / "snapshot_4.msgpack")
    assert r.returncode == 0, r.stderr
    with open(scene / "output" / "snapshot_6.msgpack", "rb") as f:
        root2 = msgpack.unpackb(f.read(), raw=False)
    assert root2["optimizer"]["nested"]["nested"]["learning_rate"] == pytest.approx(3e-3)  # and travels on
    snap1, snap2 = root["snapshot"], root2["snapshot"]

    def two_steps(**opt):
        ctx = oracle_lib.context(**dict(SMALL_KW, **opt))
        ctx.init_params()
        ctx.set_dataset(views, normals, albedos)
        ctx.set_params(np.frombuffer(snap1["params_binary"], np.float16).astype(np.float32))
        ctx.put("DENSITY_GRID", np.frombuffer(snap1["density_grid_binary"], np.float16).astype(np.float32))
        ctx.update_density_bitfield()
        ctx.set_controller(4, snap1["nerf"]["rgb"]["rays_per_batch"], snap1["nerf"]["rgb"]["measured_batch_size_before_compaction"], 0)
        for _ in range(2):
            ctx.train_step()
        out = ctx.get("PARAMS_EMA").view(np.uint16).copy()
        ctx.close()
        return out

    got = np.frombuffer(snap2["params_binary"], np.uint16)
    want = two_steps(learning_rate=3e-3, beta2=0.95, l2_reg=1e-5, ema_decay=0.9, lr_decay_start=3, lr_decay_interval=2, lr_decay_base=0.5)
    np.testing.assert_array_equal(got, want)
    assert np.any(got != two_steps())  # the library defaults give another result: the test would catch the old behaviour


# ---------------------------------------------------------------- BASELINE.json configs[0], at its exact shape
@pytest.mark.timeout(900)
def test_config_1_single_256x256_view_200_steps(install, tmp_path):
    """BASELINE.json configs[0]: "Single 256x256 view, normals+mask only, 200 steps on CPU reference path (plumbing, no GPU)" --
    SURVEY.md section 8d's generator with 1 view, 256 x 256, fx = 448, through the reference's command line (stage 1 flags of the
    pipeline) on the CPU checker. The network is tests/conftest.py's SMALL_CFG (4 levels, 2^12 samples per step: the full network
    costs the checker ~1.4 s per step on 256 cores, this one 200 steps in well under a minute on 8); everything else is the
    product's host code. A single view cannot pin the geometry, so the test checks the plumbing: 200 steps, the progress lines,
    a finite and falling loss, the snapshot and a closed mesh."""
    scene = tmp_path / "one_view"
    views, normals, albedos = synthetic.make_scene(1, 256, 448.0)
    assert len(views) == 1 and views[0]["width"] == views[0]["height"] == 256 and views[0]["focal_length"][0] == 448.0
    synthetic.write_scene(str(scene), views, normals, albedos, scale=0.5, offset=(0.5, 0.5, 0.5))
    r = run(install, "--scene", str(scene) + "/", "--maxiter", 200, "--no-gui", "--mask-weight", 1.0, "--no-albedo", "--config", "small.json",
            "--save-snapshot", "--save-mesh", "--resolution", 48)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Number of iterations : 200" in r.stdout
    losses = [float(l.split("loss=")[1]) for l in r.stdout.splitlines() if l.startswith("iteration=")]
    assert [l.split()[0] for l in r.stdout.splitlines() if l.startswith("iteration=")] == ["iteration=100"]  # every 100 steps, not at maxiter (src/main.cu:444-451)
    assert all(np.isfinite(losses)) and losses[0] > 0
    out = scene / "output"
    assert (out / "snapshot_200.msgpack").exists() and (out / "mesh_200.obj").exists() and (out / "log.txt").exists()
    import msgpack
    snap = msgpack.unpackb((out / "snapshot_200.msgpack").read_bytes(), raw=False)["snapshot"]
    assert snap["training_step"] == 200 and np.isfinite(snap["loss"])
    n_v = sum(1 for l in open(out / "mesh_200.obj") if l.startswith("v "))
    assert n_v > 50


# ---------------------------------------------------------------- snapshot interoperability with the reference (missing #3)
def test_snapshot_carries_every_key_the_reference_loader_reads(trained):
    """Testbed::load_snapshot (src/testbed.cu:3333-3390), Trainer::deserialize (trainer.h:292-301) and load_global/local_movement
    (nerf_network.h:1017-1081) read exactly these entries of the msgpack file; binaries are plain msgpack `bin` (nlohmann binary_t without
    a subtype, gpu_memory_json.h:36-56). The parameter block is [density MLP | rgb MLP | hash grid | variance] (nerf_network.h:539-583)."""
    with open(trained["scene"] / "output" / "snapshot_4.msgpack", "rb") as f:
        root = msgpack.unpackb(f.read(), raw=False)
    snap = root["snapshot"]
    assert snap["density_grid_size"] == 128                                                   # testbed.cu:3347
    for k in ("rays_per_batch", "measured_batch_size", "measured_batch_size_before_compaction"):
        assert isinstance(snap["nerf"]["rgb"][k], int)                                        # testbed.cu:3351-3353
    assert snap["nerf"]["aabb_scale"] == 1                                                    # testbed.cu:3361-3363
    assert isinstance(snap["density_grid_binary"], bytes) and len(snap["density_grid_binary"]) == 128 ** 3 * 2   # testbed.cu:3366
    assert isinstance(snap["training_step"], int) and isinstance(snap["loss"], float)         # testbed.cu:3383-3384
    assert isinstance(snap["params_binary"], bytes) and len(snap["params_binary"]) == 2 * snap["n_params"]      # trainer.h:293-294
    one = np.float16(1.0).tobytes()
    z = np.float16(0.0).tobytes()
    assert snap["rotation"] == (one + z * 3) * 2 + one + z * 3 and snap["transition"] == z * 4   # nerf_network.h:1020-1038 (identity, 852-905)
    assert snap["local_rotation"] == one + z * 3 + one + z * 3 and snap["local_transition"] == z * 4   # nerf_network.h:1062-1080
    for block in ("encoding", "network", "optimizer"):                                        # reset_network reads the config blocks of the same file (whatever the run's config held)
        assert block in root, block
    # tests/golden/snapshot_keys.json: every key path Testbed::save_snapshot and Trainer::serialize assign, parsed from their text (make_snapshot_fixture.py)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "snapshot_keys.json")) as f:
        keys = json.load(f)
    assert len(keys["written_by_save_snapshot"]) == 9 and len(keys["written_by_trainer_serialize"]) == 3
    for path in keys["written_by_save_snapshot"] + keys["written_by_trainer_serialize"]:
        if path == ["snapshot", "optimizer"]:
            continue  # only with include_optimizer_state, which main.cu does not ask for (src/main.cu: save_snapshot(path, false))
        if path == ["snapshot", "nerf", "dataset"]:
            continue  # the reference's serialised NerfDataset (image paths, transforms): not read back when --scene is given; not written here
        node = root
        for k in path:
            assert isinstance(node, dict) and k in node, path
            node = node[k]


def test_resume_from_a_reference_style_snapshot(install, trained, tmp_path):
    """A file as the reference writes it: nlohmann's to_msgpack encodings (float32 where exact, str8, map16), the extra `nerf.dataset`
    block and optimizer state this build does not read, keys in another order -- `--snapshot` restores it and trains on."""
    scene = trained["scene"]
    with open(scene / "output" / "snapshot_4.msgpack", "rb") as f:
        root = msgpack.unpackb(f.read(), raw=False)
    snap = root["snapshot"]
    ref = {k: root[k] for k in sorted(root, reverse=True) if k != "snapshot"}
    ref["snapshot"] = {k: snap[k] for k in sorted(snap, reverse=True)}
    ref["snapshot"]["nerf"]["dataset"] = {"n_images": 4, "aabb_scale": 1, "scale": 1.0, "offset": [0.0, 0.0, 0.0], "paths": ["a" * 300]}   # testbed.cu:3313, ignored with --scene
    ref["snapshot"]["optimizer"] = {"nested": {"nested": {"current_step": 4, "base_learning_rate": np.float32(1e-3).item()}}}                # trainer.h:296-298: not written by main.cu (include_optimizer_state = false)
    path = tmp_path / "reference_style.msgpack"
    with open(path, "wb") as f:
        f.write(msgpack.packb(ref, use_single_float=True, use_bin_type=True))
    r = run(install, "--scene", scene, "--maxiter", 5, "--no-gui", "--mask-weight", 1.0, "--no-albedo", "--save-snapshot", "--snapshot", path)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Loaded snapshot succeed" in r.stdout
    with open(scene / "output" / "snapshot_5.msgpack", "rb") as f:
        after = msgpack.unpackb(f.read(), raw=False)["snapshot"]
    assert after["training_step"] == 5 and after["params_binary"] != snap["params_binary"]


# ---------------------------------------------------------------- the reference's own network config through --config
def reference_style_config():
    """A config of the shape (every key, every value) of the reference's configs/nerf/base.json, rebuilt from tests/golden/reference_config_keys.json."""
    with open(os.path.join(ROOT, "tests", "golden", "reference_config_keys.json")) as f:
        leaves = json.load(f)["leaves"]
    root = {}
    for path, value in leaves:
        node = root
        for k, nxt in zip(path[:-1], path[1:]):
            if isinstance(node, list):
                while len(node) <= k:
                    node.append(None)
                if node[k] is None:
                    node[k] = [] if isinstance(nxt, int) else {}
                node = node[k]
            else:
                node = node.setdefault(k, [] if isinstance(nxt, int) else {})
        if isinstance(node, list):
            while len(node) <= path[-1]:
                node.append(None)
        node[path[-1]] = value
    return root


def test_the_references_own_config_file_parses_and_is_kept_whole(install, tmp_path):
    """`--config <the reference's base.json>`: keys this path never reads (loss, globalmove, dir_encoding's Composite, predict_global_movement, anneal_end,
    optimize_*_params ...) must parse cleanly, give the rnb_config the shipped configs/nerf/base.json gives, and travel whole in the snapshot (m_network_config,
    src/testbed.cu:3282-3313). Training with it runs on the GPU (tests/test_gpu_cli.py: the full network is minutes per step on the CPU checker)."""
    cfg = reference_style_config()
    assert set(cfg) == {"loss", "optimizer", "encoding", "network", "dir_encoding", "rgb_network", "hyperparams", "globalmove"}
    assert cfg["dir_encoding"]["nested"][0]["otype"] == "SphericalHarmonics" and cfg["hyperparams"]["anneal_end"] == 0
    path = tmp_path / "reference_base.json"
    path.write_text(json.dumps(cfg, indent=4))
    views, normals, albedos = synthetic.make_scene(2, 16, 28.0)
    snaps = {}
    for name, args in (("reference", ["--config", str(path)]), ("shipped", [])):
        scene = tmp_path / name
        synthetic.write_scene(str(scene), views, normals, albedos)
        r = run(install, "--scene", str(scene) + "/", "--maxiter", 0, "--no-gui", "--no-train", "--mask-weight", 1.0, "--no-albedo", "--save-snapshot", *args)
        assert r.returncode == 0, r.stderr[-2000:]
        with open(scene / "output" / "snapshot_0.msgpack", "rb") as f:
            snaps[name] = msgpack.unpackb(f.read(), raw=False)
    a, b = snaps["reference"], snaps["shipped"]
    assert a["snapshot"]["n_params"] == b["snapshot"]["n_params"] and a["snapshot"]["params_binary"] == b["snapshot"]["params_binary"]
    for blk in ("encoding", "network", "rgb_network", "hyperparams"):
        for k, v in b[blk].items():
            assert a[blk][k] == v, (blk, k)
    assert a["optimizer"]["nested"]["nested"]["learning_rate"] == b["optimizer"]["nested"]["nested"]["learning_rate"] == float(np.float32(0.001))  # (overlaid with the value in effect, a float)
    # the keys this build does not read are still there for a reader of the snapshot
    assert a["loss"]["otype"] == "Huber" and a["globalmove"]["optimizer"]["nested"]["decay_interval"] == 25
    assert a["dir_encoding"]["otype"] == "Composite" and a["hyperparams"]["predict_global_movement"] is True
    assert a["optimizer"]["nested"]["nested"]["optimize_params_components"] == {"rgb_network": True, "density_network": True}


def test_unsupported_network_configs_are_refused_by_name(install, tmp_path):
    """A width-48 density network (n_levels 15..22: the reference then loads utils/mlp_weights.txt, nerf_network.h:595-600), other layer sizes or feature counts:
    refused with the key's name, exit code 1 -- not silently trained with the fixed architecture."""
    views, normals, albedos = synthetic.make_scene(2, 16, 28.0)
    synthetic.write_scene(str(tmp_path / "s"), views, normals, albedos)
    for patch, needle in (({"encoding": {"n_levels": 16}}, "width 48"), ({"network": {"n_neurons": 128}}, "network.n_neurons"), ({"rgb_network": {"n_hidden_layers": 3}}, "rgb_network.n_hidden_layers"),
                          ({"encoding": {"n_features_per_level": 4}}, "encoding.n_features_per_level"), ({"network": {"activation": "Sine"}}, "network.activation")):
        cfg = reference_style_config()
        for blk, kv in patch.items():
            cfg[blk].update(kv)
        p = tmp_path / "bad.json"
        p.write_text(json.dumps(cfg))
        r = run(install, "--scene", str(tmp_path / "s") + "/", "--maxiter", 1, "--no-gui", "--config", str(p))
        assert r.returncode == 1 and needle in r.stderr, (patch, r.returncode, r.stderr[-500:])


def test_snapshot_records_the_accumulate_mode_and_a_resume_continues_in_it(install, tmp_path):
    """`--accumulate half` (this build's flag; include/rnb_neus2.h rnb_config::accumulate): the snapshot's hyperparams carry the mode, a resumed run without the flag continues
    in it (equal to an uninterrupted half-mode run on the checker), and the flag overrides the record."""
    views, normals, albedos = synthetic.make_scene(3, 32, 56.0)
    scene = tmp_path / "s"
    synthetic.write_scene(str(scene), views, normals, albedos)
    common = ["--scene", str(scene) + "/", "--no-gui", "--mask-weight", 1.0, "--no-albedo", "--save-snapshot"]
    r = run(install, *common, "--maxiter", 3, "--config", "small.json", "--accumulate", "half")
    assert r.returncode == 0, r.stderr[-1500:]
    snap3 = msgpack.unpackb((scene / "output" / "snapshot_3.msgpack").read_bytes(), raw=False)
    assert snap3["hyperparams"]["accumulate"] == "half"
    r = run(install, *common, "--maxiter", 5, "--snapshot", scene / "output" / "snapshot_3.msgpack")
    assert r.returncode == 0, r.stderr[-1500:]
    snap5 = msgpack.unpackb((scene / "output" / "snapshot_5.msgpack").read_bytes(), raw=False)
    assert snap5["hyperparams"]["accumulate"] == "half"
    # the same two steps on the checker in the half mode, from the snapshot's state
    ctx = oracle_lib.context(accumulate=1, **SMALL_KW)
    try:
        ctx.init_params()
        ctx.set_dataset(views, normals, albedos)
        s3 = snap3["snapshot"]
        ctx.set_params(np.frombuffer(s3["params_binary"], np.float16).astype(np.float32))
        ctx.put("DENSITY_GRID", np.frombuffer(s3["density_grid_binary"], np.float16).astype(np.float32))
        ctx.update_density_bitfield()
        ctx.set_controller(3, s3["nerf"]["rgb"]["rays_per_batch"], s3["nerf"]["rgb"]["measured_batch_size_before_compaction"], 0)
        for _ in range(2):
            ctx.train_step()
        np.testing.assert_array_equal(np.frombuffer(snap5["snapshot"]["params_binary"], np.uint16), ctx.get("PARAMS_EMA").view(np.uint16))
    finally:
        ctx.close()
    r = run(install, *common, "--maxiter", 4, "--snapshot", scene / "output" / "snapshot_3.msgpack", "--accumulate", "fp32")
    assert r.returncode == 0, r.stderr[-1500:]
    assert msgpack.unpackb((scene / "output" / "snapshot_4.msgpack").read_bytes(), raw=False)["hyperparams"]["accumulate"] == "fp32"
    r = run(install, *common, "--maxiter", 1, "--config", "small.json", "--accumulate", "double")
    assert r.returncode == 255 and "--accumulate takes fp32 or half" in r.stderr
