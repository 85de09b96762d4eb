"""Generates tests/golden/prepare_scaling_vectors.json by running the REFERENCE's scaling-mode cascade (rnb_neus2/prepare.py: _compute_scaling, which calls
scaling.py: extract_cameras_for_scaling / compute_unit_sphere_scaling) on seeded loader dicts WITHOUT mask files: the module imports cv2 at its top (absent in
this image); as in make_python_fixtures.py it is replaced by an EMPTY module in sys.modules so that the file imports -- a view without a mask file never
reaches cv2 (scaling.py:287-291), so what runs here is the reference's own dispatch: "none", landmarks ("pcd"), camera centres, "auto" falling through the
silhouette branch, the RuntimeError when a mode finds no data, and the info lines it logs. (The silhouette branches need cv2.imread / findContours; the v1 silhouette
arithmetic itself is pinned by scaling_vectors.json.) /root/reference is read at generation time only.

Usage:  python tests/golden/make_prepare_fixtures.py
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class Log:
    def __init__(self):
        self.lines = []

    def info(self, msg):
        self.lines.append(str(msg))

    warning = info


def loader_dict(rng, n_views, n_landmarks):
    views = []
    for k in range(n_views):
        c2w = np.eye(4, dtype=np.float32)
        q, r = np.linalg.qr(rng.standard_normal((3, 3)))
        c2w[:3, :3] = (q * np.sign(np.diag(r))).astype(np.float32)
        c2w[:3, 3] = rng.uniform(-3, 3, 3).astype(np.float32)
        K = np.array([[900.0, 0, 320.0], [0, 905.0, 240.0], [0, 0, 1]], dtype=np.float32)
        views.append({"c2w": c2w, "K": K, "mask_path": "", "normal_path": "", "albedo_path": ""})
    d = {"views": views}
    if n_landmarks is not None:
        d["landmarks"] = (rng.standard_normal((n_landmarks, 3)) * rng.uniform(0.5, 4) + rng.uniform(-2, 2, 3)).astype(np.float32)
    return d


def main():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))  # imported at the top of the modules, not used by what runs here
    sys.path.insert(0, REF)
    from rnb_neus2 import prepare as ref
    rng = np.random.default_rng(2024)
    cases = []
    plan = [("none", 3, 50, 1.0), ("auto", 4, 200, 1.0), ("auto", 5, None, 1.0), ("auto", 3, 0, 0.8), ("pcd", 4, 120, 0.9), ("pcd", 4, None, 1.0), ("pcd", 3, 0, 1.0),
            ("cameras", 6, 300, 1.0), ("cameras", 2, None, 0.5), ("silhouettes", 3, 100, 1.0), ("silhouettes_v2", 3, 100, 1.0), ("cameras", 0, None, 1.0), ("bogus", 3, 10, 1.0)]
    for mode, n_views, n_lm, sphere in plan:
        d = loader_dict(rng, n_views, n_lm)
        log = Log()
        case = {"mode": mode, "sphere_scale": sphere, "margin_px": 20,
                "views_c2w": [v["c2w"].astype(np.float64).tolist() for v in d["views"]],
                "landmarks": None if "landmarks" not in d else d["landmarks"].astype(np.float64).tolist()}
        try:
            center, factor, matrix = ref._compute_scaling(d, mode, sphere, 20, log)
            case.update(center=np.asarray(center, np.float64).tolist(), factor=float(factor), matrix=np.asarray(matrix, np.float64).tolist(),
                        dtypes=[str(np.asarray(center).dtype), str(np.asarray(matrix).dtype)], raises=None)
        except Exception as e:  # noqa: BLE001 -- the fixture records which exception the reference raises
            case.update(raises=type(e).__name__, message=str(e))
        case["log"] = log.lines
        cases.append(case)
    out = {"source": "rnb_neus2/prepare.py:_compute_scaling of RobinBruneau/RNb-NeuS2, run by tests/golden/make_prepare_fixtures.py", "cases": cases}
    path = os.path.join(HERE, "prepare_scaling_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(len(cases), "cases;", [c["raises"] for c in cases], os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
