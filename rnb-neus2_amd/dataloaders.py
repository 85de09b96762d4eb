"""Input formats of the pipeline -> one standard dict (mirror of rnb_neus2/dataloaders/{__init__,base,rnb_loader,
sfm_json_loader}.py):

    {"views": [{"c2w": (4,4), "K": (4,4), "normal_path", "albedo_path"|None, "mask_path"|None, "pose_id"}, ...],
     "landmarks": (N,3)|None, "image_width", "image_height", "scale_mat": (4,4)|None}

Formats: a directory holding cameras.npz (+ normal/, albedo/, mask/), the .npz itself, or an AliceVision SfMData JSON
(.sfm / .json). The pyalicevision-backed loader of the reference needs a package that does not exist on the target
image; .sfm / .abc inputs go to the JSON parser exactly as the reference does when that import fails."""
import json
import os
import warnings

import numpy as np
from scipy import linalg

from . import hostlib

# AliceVision (y down, z forward) -> the y-up world the cameras.npz data uses
WORLD_CORRECTION = np.diag([1.0, -1.0, -1.0])


def load_K_Rt_from_P(P):
    """3x4 projection -> (4x4 intrinsics with K[2,2]=1, 4x4 camera-to-world). RQ factorisation with a positive
    intrinsic diagonal and det(R)=+1 — the decomposition cv2.decomposeProjectionMatrix returns for a camera in front
    of the scene (rnb_loader.py:20-36)."""
    P = np.asarray(P, np.float64)
    K, R = linalg.rq(P[:3, :3])
    S = np.diag(np.sign(np.diag(K)))
    K, R = K @ S, S @ R
    if np.linalg.det(R) < 0:
        R = -R
    center = -np.linalg.solve(P[:3, :3], P[:3, 3])
    intrinsics = np.eye(4)
    intrinsics[:3, :3] = K / K[2, 2]
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.T
    pose[:3, 3] = center
    return intrinsics, pose


class BaseDataLoader:
    def load(self):
        raise NotImplementedError


class RnbDataLoader(BaseDataLoader):
    """cameras.npz (world_mat_i, scale_mat_i) + normal/ [albedo/] [mask/] with zero-padded file names. (rnb_loader.py:39-110)"""

    def __init__(self, data_dir):
        self.data_dir = data_dir

    def load(self):
        npz_path = os.path.join(self.data_dir, "cameras.npz")
        if not os.path.exists(npz_path):
            raise FileNotFoundError("cameras.npz not found in {}".format(self.data_dir))
        cams = np.load(npz_path)
        n_images = 1 + max(int(k.rsplit("_", 1)[-1]) for k in cams.keys())
        normal_dir = os.path.join(self.data_dir, "normal")
        if not os.path.isdir(normal_dir):
            raise FileNotFoundError("normal/ folder not found in {}".format(self.data_dir))
        first = sorted(os.listdir(normal_dir))[0]
        digits = len(first.split(".")[0])
        width, height, _, _ = hostlib.png_info(os.path.join(normal_dir, first))
        albedo_dir = os.path.join(self.data_dir, "albedo")
        mask_dir = os.path.join(self.data_dir, "mask")
        views = []
        for i in range(n_images):
            P = (cams["world_mat_{}".format(i)].astype(np.float32) @ cams["scale_mat_{}".format(i)].astype(np.float32))[:3, :4]
            K, c2w = load_K_Rt_from_P(P)
            name = "{:0{n}d}.png".format(i, n=digits)
            mask_path = os.path.join(mask_dir, name)
            views.append(dict(c2w=c2w, K=K.astype(np.float32), normal_path=os.path.join(normal_dir, name),
                              albedo_path=os.path.join(albedo_dir, name) if os.path.isdir(albedo_dir) else None,
                              mask_path=mask_path if os.path.exists(mask_path) else None, pose_id=str(i)))
        return dict(views=views, landmarks=None, image_width=width, image_height=height, scale_mat=cams["scale_mat_0"].astype(np.float32))


def parse_sfm_json(data, sfm_dir=None):
    """SfMData dict -> (cameras, landmarks). Cameras carry view_id, pose_id, image_path, R_cam2world, center, fx, fy,
    cx, cy, width, height; views whose intrinsic or pose is missing are skipped. (sfm_json_loader.py:26-117)"""
    intrinsics = {i["intrinsicId"]: i for i in data.get("intrinsics", [])}
    poses = {p["poseId"]: p["pose"]["transform"] for p in data.get("poses", [])}
    cameras = []
    for view in data.get("views", []):
        intr, xf = intrinsics.get(view["intrinsicId"]), poses.get(view["poseId"])
        if intr is None or xf is None:
            continue
        width, height = int(intr["width"]), int(intr["height"])
        if "pxFocalLength" in intr:
            f = intr["pxFocalLength"]
            fx, fy = (float(f[0]), float(f[1])) if isinstance(f, list) else (float(f), float(f))
        else:
            if "sensorWidth" not in intr:
                warnings.warn("sensorWidth not found, using default 36.0mm")
            fx = fy = float(intr["focalLength"]) * width / float(intr.get("sensorWidth", 36.0))
        pp = intr.get("principalPoint", ["0", "0"])
        path = view.get("path", "")
        if path and sfm_dir is not None and not os.path.isabs(path):
            path = os.path.join(sfm_dir, path)
        cameras.append(dict(view_id=view["viewId"], pose_id=view["poseId"], image_path=path,
                            R_cam2world=WORLD_CORRECTION @ np.array([float(r) for r in xf["rotation"]]).reshape(3, 3),
                            center=WORLD_CORRECTION @ np.array([float(c) for c in xf["center"]]),
                            fx=fx, fy=fy, cx=width / 2.0 + float(pp[0]), cy=height / 2.0 + float(pp[1]), width=width, height=height))
    pts = [[float(x) for x in s["X"][:3]] for s in data.get("structure", []) if s.get("X") is not None]
    landmarks = (WORLD_CORRECTION @ np.array(pts).T).T if pts else None
    return cameras, landmarks


class SfmJsonDataLoader(BaseDataLoader):
    """Normal-map SfMData (+ optional albedo / mask SfMData matched by poseId, or a folder of <poseId>.<ext> masks).
    (sfm_json_loader.py:120-216)"""

    def __init__(self, sfm_path, normal_sfm_path=None, albedo_sfm_path="", mask_sfm_path="", mask_folder_path=""):
        self.sfm_path = sfm_path
        self.normal_sfm_path = normal_sfm_path or sfm_path
        self.albedo_sfm_path = albedo_sfm_path
        self.mask_sfm_path = mask_sfm_path
        self.mask_folder_path = mask_folder_path

    @staticmethod
    def _by_pose(path):
        if not (path and os.path.exists(path)):
            return {}
        with open(path) as f:
            cams, _ = parse_sfm_json(json.load(f), os.path.dirname(os.path.abspath(path)))
        return {c["pose_id"]: c["image_path"] for c in cams}

    def load(self):
        with open(self.normal_sfm_path) as f:
            cams, landmarks = parse_sfm_json(json.load(f), os.path.dirname(os.path.abspath(self.normal_sfm_path)))
        if not cams:
            raise RuntimeError("No valid views in {}".format(self.normal_sfm_path))
        albedos, masks = self._by_pose(self.albedo_sfm_path), self._by_pose(self.mask_sfm_path)
        views = []
        for cam in cams:
            c2w = np.eye(4, dtype=np.float32)
            c2w[:3, :3], c2w[:3, 3] = cam["R_cam2world"], cam["center"]
            K = np.eye(4, dtype=np.float32)
            K[0, 0], K[1, 1], K[0, 2], K[1, 2] = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
            pose_id = cam["pose_id"]
            mask_path = masks.get(pose_id)
            if mask_path is None and self.mask_folder_path and os.path.isdir(self.mask_folder_path):
                for ext in (".png", ".jpg", ".jpeg", ".exr"):
                    cand = os.path.join(self.mask_folder_path, "{}{}".format(pose_id, ext))
                    if os.path.exists(cand):
                        mask_path = cand
                        break
            views.append(dict(c2w=c2w, K=K, normal_path=cam["image_path"], albedo_path=albedos.get(pose_id), mask_path=mask_path, pose_id=pose_id))
        return dict(views=views, landmarks=landmarks, image_width=cams[0]["width"], image_height=cams[0]["height"], scale_mat=None)


def create_loader(input_path, **kwargs):
    """Pick the loader from the path: directory with cameras.npz / .npz / .sfm / .abc / .json. (dataloaders/__init__.py:13-70)"""
    if os.path.isdir(input_path):
        if os.path.exists(os.path.join(input_path, "cameras.npz")):
            return RnbDataLoader(input_path)
        raise FileNotFoundError("No cameras.npz found in {}. Provide a .sfm or .json file instead.".format(input_path))
    ext = os.path.splitext(input_path)[1].lower()
    if ext == ".npz":
        return RnbDataLoader(os.path.dirname(input_path))
    if ext in (".json", ".sfm"):
        return SfmJsonDataLoader(sfm_path=input_path, normal_sfm_path=input_path, albedo_sfm_path=kwargs.get("albedo_sfm_path", ""),
                                 mask_sfm_path=kwargs.get("mask_sfm_path", ""), mask_folder_path=kwargs.get("mask_folder_path", ""))
    raise ValueError("Unsupported input format: {}. Supported: directory with cameras.npz, .npz, .sfm, .abc, .json".format(ext))


def load_data(input_path, **kwargs):
    return create_loader(input_path, **kwargs).load()
