"""Generates tests/golden/pipeline_argv.json and tests/golden/scaling_vectors.json by importing the REFERENCE's Python
modules from /root/reference (possible only in the build container; the fixtures are what travels).

 * pipeline_argv.json — the testbed command lines `rnb_neus2.pipeline` issues (run_two_stage, run_with_albedo_scaling,
   run_full_pipeline) for a set of argument combinations, recorded by a stub testbed executable that writes its argv and
   creates the snapshot / mesh files the pipeline looks for. Modules that need cv2 / trimesh (absent here) are replaced
   by recorders in sys.modules so that pipeline.py itself runs unmodified.
 * scaling_vectors.json — inputs and outputs of rnb_neus2.scaling.{compute_unit_sphere_scaling,
   compute_scaling_from_silhouettes, _triangulate_scene_center} on seeded inputs.

Usage:  python tests/golden/make_python_fixtures.py
"""
import json
import os
import stat
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

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


class Quiet:
    def info(self, m): pass
    def warning(self, m): pass
    def error(self, m): pass


def normalise(argv, root):
    return [s.replace(root, "<ROOT>") for s in argv]


def pipeline_fixture():
    sys.path.insert(0, REF)
    calls = []
    fake_alb = types.ModuleType("rnb_neus2.albedo_scaling")

    def compute_albedo_scale_ratios(**kw):
        calls.append(["compute_albedo_scale_ratios", {k: (v if not hasattr(v, "info") else "<logger>") for k, v in kw.items()}])
        return np.ones((1, 3))

    def scale_and_save_albedos(**kw):
        calls.append(["scale_and_save_albedos", {k: ("<ratios>" if k == "scale_ratios" else v if not hasattr(v, "info") else "<logger>") for k, v in kw.items()}])
        os.makedirs(kw["output_albedo_path"], exist_ok=True)

    fake_alb.compute_albedo_scale_ratios = compute_albedo_scale_ratios
    fake_alb.scale_and_save_albedos = scale_and_save_albedos
    sys.modules["rnb_neus2.albedo_scaling"] = fake_alb
    fake_dl = types.ModuleType("rnb_neus2.dataloaders")
    fake_dl.load_data = lambda input_path, **kw: calls.append(["load_data", input_path, {k: v for k, v in kw.items() if k != "logger"}]) or {"views": []}
    sys.modules["rnb_neus2.dataloaders"] = fake_dl
    fake_prep = types.ModuleType("rnb_neus2.prepare")

    def prepare_testbed_data(data, out, logger, **kw):
        calls.append(["prepare_testbed_data", out, kw])
        os.makedirs(os.path.join(out, "albedos"), exist_ok=True)

    fake_prep.prepare_testbed_data = prepare_testbed_data
    sys.modules["rnb_neus2.prepare"] = fake_prep
    from rnb_neus2 import pipeline

    pipeline.postprocess_mesh = lambda data_dir, out, logger=None: calls.append(["postprocess_mesh", data_dir, out])
    cases = []

    def record(name, fn):
        root = tempfile.mkdtemp(prefix="pipefix_")
        tb = os.path.join(root, "testbed")
        with open(tb, "w") as f:
            f.write(STUB)
        os.chmod(tb, os.stat(tb).st_mode | stat.S_IEXEC)
        log = os.path.join(root, "argv.jsonl")
        os.environ["STUB_LOG"] = log
        del calls[:]
        fn(root, tb)
        argvs = [normalise(json.loads(l), root) for l in open(log)] if os.path.exists(log) else []
        other = json.loads(json.dumps(calls, default=str).replace(root, "<ROOT>"))
        cases.append(dict(name=name, testbed_argv=argvs, other_calls=other))

    def scene(root):
        d = os.path.join(root, "scene")
        os.makedirs(os.path.join(d, "albedos"), exist_ok=True)
        return d

    record("two_stage_default", lambda r, tb: pipeline.run_two_stage(tb, scene(r), 10000, ["--mask-weight", "1.0"], logger=Quiet()))
    record("two_stage_no_albedo_res512_extra", lambda r, tb: pipeline.run_two_stage(tb, scene(r), 1000, ["--mask-weight", "0.5", "--lone"], resolution=512, no_albedo=True, extra_flags=["--bce"], logger=Quiet()))
    record("two_stage_odd_steps", lambda r, tb: pipeline.run_two_stage(tb, scene(r), 100, [], no_albedo=True, logger=Quiet()))
    record("albedo_scaling_default", lambda r, tb: pipeline.run_with_albedo_scaling(tb, scene(r), 20000, ["--mask-weight", "1.0"], logger=Quiet()))
    record("albedo_scaling_short", lambda r, tb: pipeline.run_with_albedo_scaling(tb, scene(r), 3000, ["--mask-weight", "2.0", "--supernormal"], resolution=256, warmup_ratio=0.5, n_samples=50, logger=Quiet()))
    record("full_default", lambda r, tb: pipeline.run_full_pipeline(os.path.join(r, "in"), tb, os.path.join(r, "out"), logger=Quiet()))
    record("full_flags", lambda r, tb: pipeline.run_full_pipeline(os.path.join(r, "in.sfm"), tb, os.path.join(r, "out"), max_steps=3000, mesh_resolution=256, scaling_mode="cameras",
                                                                  sphere_scale=0.8, margin_px=5, mask_weight=0.25, super_normal=True, use_l1=True, use_rgb_plus=False,
                                                                  albedo_sfm_path="a.sfm", mask_sfm_path="m.sfm", mask_folder_path="masks", logger=Quiet()))
    record("full_albedo", lambda r, tb: pipeline.run_full_pipeline(os.path.join(r, "in"), tb, os.path.join(r, "out"), max_steps=6000, has_albedo=True, warmup_ratio=0.25, n_samples=77, logger=Quiet()))
    with open(os.path.join(HERE, "pipeline_argv.json"), "w") as f:
        json.dump(dict(_generator="tests/golden/make_python_fixtures.py (reference rnb_neus2/pipeline.py with a stub testbed)", cases=cases), f, indent=1)


def scaling_fixture():
    sys.path.insert(0, REF)
    from rnb_neus2 import scaling
    rng = np.random.default_rng(7)
    out = dict(_generator="tests/golden/make_python_fixtures.py (reference rnb_neus2/scaling.py)", unit_sphere=[], silhouettes=[])
    for n, s in ((50, 1.0), (400, 0.8), (1000, 2.0)):
        pts = rng.normal(size=(n, 3)) * rng.uniform(0.5, 3.0, size=3) + rng.uniform(-2, 2, size=3)
        pts[:3] *= 25.0  # outliers
        c, f, m = scaling.compute_unit_sphere_scaling(pts, s)
        out["unit_sphere"].append(dict(points=pts.tolist(), sphere_scale=s, center=np.asarray(c).tolist(), scale_factor=float(f), scale_matrix=np.asarray(m).tolist()))
    for n_views, res, obj_r, centre in ((6, 64, 0.7, (0.2, -0.1, 0.3)), (9, 48, 0.4, (0.0, 0.0, 0.0))):
        cams, masks = [], []
        for k in range(n_views):
            ang = 2 * np.pi * k / n_views
            eye = np.array([4 * np.cos(ang), 0.8 * np.sin(3 * ang), 4 * np.sin(ang)])
            fwd = np.asarray(centre) + rng.normal(size=3) * 0.05 - eye
            fwd /= np.linalg.norm(fwd)
            right = np.cross(fwd, [0, 1, 0]); right /= np.linalg.norm(right)
            down = np.cross(fwd, right)
            R = np.stack([right, down, fwd], axis=1)
            fx = fy = 1.2 * res
            cx = cy = res / 2
            pc = R.T @ (np.asarray(centre) - eye)
            u0, v0 = fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy
            yy, xx = np.mgrid[0:res, 0:res]
            mask = (((xx - u0) ** 2 + (yy - v0) ** 2) <= (fx * obj_r / pc[2]) ** 2).astype(np.float32)
            cams.append(dict(fx=fx, fy=fy, cx=cx, cy=cy, R_cam2world=R, center=eye))
            masks.append(mask)
        c1, f1 = scaling.compute_scaling_from_silhouettes(cams, masks, sphere_scale=1.0)
        c2, f2 = scaling.compute_scaling_from_silhouettes(cams, masks, sphere_scale=0.5, fg_area_ratio=2.0)
        tri = scaling._triangulate_scene_center(cams, masks)
        out["silhouettes"].append(dict(
            cameras=[dict(fx=c["fx"], fy=c["fy"], cx=c["cx"], cy=c["cy"], R_cam2world=c["R_cam2world"].tolist(), center=c["center"].tolist()) for c in cams],
            mask_circles=None, masks_packed=[np.packbits(m.astype(np.uint8)).tolist() for m in masks], res=res,
            default=dict(center=np.asarray(c1).tolist(), scale_factor=f1), alt=dict(sphere_scale=0.5, fg_area_ratio=2.0, center=np.asarray(c2).tolist(), scale_factor=f2),
            triangulated=np.asarray(tri).tolist()))
    with open(os.path.join(HERE, "scaling_vectors.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    assert os.path.isdir(REF), "fixtures can only be regenerated where the reference checkout is mounted"
    pipeline_fixture()
    scaling_fixture()
    print("wrote fixtures to", HERE)
