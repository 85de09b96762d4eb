"""Generates tests/golden/albedo_camera_vectors.json by running the REFERENCE's camera loader of the albedo-scaling stage
(rnb_neus2/albedo_scaling.py: load_cameras / load_cameras_from_transform_json) on seeded transform.json documents. The module imports cv2 and trimesh at its top
(absent in this image); as in make_python_fixtures.py they are replaced by EMPTY modules in sys.modules so that the file imports -- the functions run here never touch
them (the projection-matrix loader and the ray casting, which do, are not part of this fixture). /root/reference is read at generation time only.

Usage:  python tests/golden/make_albedo_fixtures.py
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def documents(rng):
    def frame(k, mode):
        c2w = np.eye(4)
        q, r = np.linalg.qr(rng.standard_normal((3, 3)))
        c2w[:3, :3] = q * np.sign(np.diag(r))
        c2w[:3, 3] = rng.uniform(-2, 2, 3)
        f = {"albedo_path": "albedos/%05d.png" % k, "normal_path": "normals/%05d.png" % k, "transform_matrix": c2w.tolist()}
        if mode == "matrix":
            K = np.eye(4)
            K[0, 0], K[1, 1], K[0, 2], K[1, 2] = rng.uniform(400, 1500), rng.uniform(400, 1500), rng.uniform(200, 700), rng.uniform(200, 700)
            f["intrinsic_matrix"] = K.tolist()
        elif mode == "per_frame":
            f.update(fl_x=float(rng.uniform(400, 1500)), fl_y=float(rng.uniform(400, 1500)), cx=float(rng.uniform(200, 700)), cy=float(rng.uniform(200, 700)))
        elif mode == "fx_only":
            f.update(fl_x=float(rng.uniform(400, 1500)))
        return f
    n2w = np.eye(4)
    n2w[:3, :3] *= 2.5
    n2w[:3, 3] = [0.25, -1.0, 3.0]
    return [
        {"doc": {"frames": [frame(k, "matrix") for k in range(4)], "w": 640, "h": 480}, "order": [2, 0, 3, 1]},
        {"doc": {"frames": [frame(k, "matrix") for k in range(3)], "n2w": n2w.tolist()}, "order": [0, 1, 2]},
        {"doc": {"frames": [frame(k, "per_frame") for k in range(3)], "w": 800, "h": 600}, "order": [1, 2, 0]},
        {"doc": {"frames": [frame(k, "global") for k in range(3)], "fl_x": 1111.5, "cx": 401.25, "cy": 299.5, "w": 800, "h": 600}, "order": [0, 2]},
        {"doc": {"frames": [frame(k, "global") for k in range(2)], "w": 1000, "h": 700}, "order": [1, 0]},                     # no focal length anywhere: the 500 default, w / 2, h / 2
        {"doc": {"frames": [frame(k, "fx_only") for k in range(2)], "fl_y": 900.0, "w": 1000, "h": 700, "n2w": n2w.tolist()}, "order": [0, 1]},
    ]


def main():
    for name in ("cv2", "trimesh"):
        sys.modules.setdefault(name, types.ModuleType(name))  # imported at the top of the module, not used by what runs here
    sys.path.insert(0, REF)
    from rnb_neus2 import albedo_scaling as ref
    rng = np.random.default_rng(77)
    cases = []
    for d in documents(rng):
        names = ["/some/where/%05d.png" % k for k in d["order"]]  # matched to the frames by file stem
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "transform.json")
            with open(path, "w") as f:
                json.dump(d["doc"], f)
            K, R, C = ref.load_cameras(path, names)
        cases.append({"document": d["doc"], "albedo_images": names, "K": K.astype(np.float64).tolist(), "R_c2w": R.astype(np.float64).tolist(), "centers": C.astype(np.float64).tolist(),
                      "dtypes": [str(K.dtype), str(R.dtype), str(C.dtype)]})
    # a frame that is missing: the reference raises RuntimeError
    try:
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "transform.json")
            with open(path, "w") as f:
                json.dump(cases[0]["document"], f)
            ref.load_cameras(path, ["/x/00099.png"])
        missing = None
    except Exception as e:
        missing = type(e).__name__
    try:
        ref.load_cameras("/x/cameras.bin", ["a.png"])
        bad = None
    except Exception as e:
        bad = type(e).__name__
    out = {"source": "rnb_neus2/albedo_scaling.py of RobinBruneau/RNb-NeuS2 (load_cameras -> load_cameras_from_transform_json), run by tests/golden/make_albedo_fixtures.py",
           "cases": cases, "missing_frame_raises": missing, "unknown_suffix_raises": bad}
    with open(os.path.join(HERE, "albedo_camera_vectors.json"), "w") as f:
        json.dump(out, f)
    print(len(cases), "cases;", missing, bad, os.path.getsize(os.path.join(HERE, "albedo_camera_vectors.json")), "bytes")


if __name__ == "__main__":
    main()
