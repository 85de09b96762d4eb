"""Generates tests/golden/sfm_loader_vectors.json by running the REFERENCE's SfMData JSON loader (rnb_neus2/dataloaders/sfm_json_loader.py: parse_sfm_json and
SfmJsonDataLoader.load) on seeded synthetic SfMData documents. The module and its base class are loaded from their files under /root/reference (build container only)
into a bare package object, so that the package's __init__ -- which imports the cv2-based RNb loader -- is not executed: nothing is stubbed, the loader itself needs numpy
only. The fixture holds every input document and the loader's outputs (paths relative to the scene directory); the product's dataloaders.py is tested against it.

Usage:  python tests/golden/make_loader_fixtures.py
"""
import importlib.util
import json
import os
import sys
import tempfile
import types
import warnings

import numpy as np

REF = "/root/reference/rnb_neus2/dataloaders"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_loader():
    pkg = types.ModuleType("ref_dataloaders")
    pkg.__path__ = [REF]
    sys.modules["ref_dataloaders"] = pkg
    for name in ("base", "sfm_json_loader"):
        spec = importlib.util.spec_from_file_location("ref_dataloaders." + name, os.path.join(REF, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
    return sys.modules["ref_dataloaders.sfm_json_loader"]


def random_rotation(rng):
    q, r = np.linalg.qr(rng.standard_normal((3, 3)))
    q *= np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def make_document(rng, n_views, focal_kind, with_structure, drop_pose=None):
    """An AliceVision SfMData document: every number a string, as the files are."""
    intr = {"intrinsicId": "900", "width": str(int(rng.integers(320, 1300))), "height": str(int(rng.integers(240, 1000))),
            "principalPoint": ["%.6f" % rng.uniform(-8, 8), "%.6f" % rng.uniform(-8, 8)]}
    if focal_kind == "px_pair":
        intr["pxFocalLength"] = ["%.5f" % rng.uniform(500, 2000), "%.5f" % rng.uniform(500, 2000)]
    elif focal_kind == "px_scalar":
        intr["pxFocalLength"] = "%.5f" % rng.uniform(500, 2000)
    elif focal_kind == "mm":
        intr["focalLength"] = "%.4f" % rng.uniform(12, 85)
        intr["sensorWidth"] = "%.3f" % rng.uniform(20, 40)
    else:  # mm without a sensor width: the 36 mm default (with a warning)
        intr["focalLength"] = "%.4f" % rng.uniform(12, 85)
    views, poses = [], []
    for k in range(n_views):
        pid = str(1000 + 7 * k)
        views.append({"viewId": str(50 + k), "poseId": pid, "intrinsicId": "900", "path": ("normals/%05d.png" % k) if k % 3 else ("/data/abs/%05d.png" % k)})
        if drop_pose is not None and k == drop_pose:
            continue
        R, c = random_rotation(rng), rng.uniform(-3, 3, 3)
        poses.append({"poseId": pid, "pose": {"transform": {"rotation": ["%.17g" % v for v in R.ravel()], "center": ["%.17g" % v for v in c]}}})
    doc = {"version": ["1", "2", "4"], "views": views, "intrinsics": [intr], "poses": poses}
    if with_structure:
        doc["structure"] = [{"landmarkId": str(i), "X": ["%.17g" % v for v in rng.uniform(-2, 2, 3)]} for i in range(int(rng.integers(3, 9)))]
    return doc


def main():
    ref = load_reference_loader()
    rng = np.random.default_rng(20260930)
    cases = []
    specs = [(4, "px_pair", True, None), (3, "px_scalar", False, None), (5, "mm", True, 2), (2, "mm_default", False, None)]
    for ci, (n_views, focal, structure, drop) in enumerate(specs):
        doc = make_document(rng, n_views, focal, structure, drop)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cams, landmarks = ref.parse_sfm_json(doc, "/scene")
        case = {"document": doc, "sfm_dir": "/scene",
                "parse": {"cameras": [{k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in c.items()} for c in cams],
                          "landmarks": None if landmarks is None else landmarks.tolist()}}
        # the loader class on files: normals document + an albedo document (every second view) + a mask folder (one pose)
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "normals.sfm"), "w") as f:
                json.dump(doc, f)
            alb = dict(doc, views=[dict(v, viewId=str(700 + i), path="albedos/%s.png" % v["poseId"]) for i, v in enumerate(doc["views"]) if i % 2 == 0])
            with open(os.path.join(d, "albedos.sfm"), "w") as f:
                json.dump(alb, f)
            os.makedirs(os.path.join(d, "masks"))
            mask_pose = doc["views"][-1]["poseId"]
            open(os.path.join(d, "masks", mask_pose + ".jpg"), "wb").close()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out = ref.SfmJsonDataLoader(os.path.join(d, "normals.sfm"), albedo_sfm_path=os.path.join(d, "albedos.sfm"), mask_folder_path=os.path.join(d, "masks")).load()

            def rel(p):
                return None if p is None else p.replace(d, "<DIR>")
            case["load"] = {"albedo_document": alb, "mask_files": [mask_pose + ".jpg"], "image_width": out["image_width"], "image_height": out["image_height"],
                            "scale_mat": out["scale_mat"], "landmarks": None if out["landmarks"] is None else out["landmarks"].tolist(),
                            "views": [{"c2w": v["c2w"].astype(np.float64).tolist(), "K": v["K"].astype(np.float64).tolist(), "c2w_dtype": str(v["c2w"].dtype),
                                       "normal_path": rel(v["normal_path"]), "albedo_path": rel(v["albedo_path"]), "mask_path": rel(v["mask_path"]), "pose_id": v["pose_id"]}
                                      for v in out["views"]]}
        cases.append(case)
    with open(os.path.join(HERE, "sfm_loader_vectors.json"), "w") as f:
        json.dump({"source": "rnb_neus2/dataloaders/sfm_json_loader.py of RobinBruneau/RNb-NeuS2 (parse_sfm_json, SfmJsonDataLoader.load), run by tests/golden/make_loader_fixtures.py", "cases": cases}, f)
    print("%d cases, %d bytes" % (len(cases), os.path.getsize(os.path.join(HERE, "sfm_loader_vectors.json"))))


if __name__ == "__main__":
    main()
