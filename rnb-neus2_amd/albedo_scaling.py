"""Per-view albedo gain equalisation between the two training phases (mirror of rnb_neus2/albedo_scaling.py:214-436).

For every view, sample foreground pixels, hit the phase-1 mesh, re-project the visible surface points into the two
neighbouring views, and take the median ratio of the albedos seen; chaining the medians around the ring and dividing
by the mean gives one RGB gain per view. Ray casting uses librnb_host.so instead of trimesh/embree."""
import json
import os
from pathlib import Path

import numpy as np

from . import hostlib
from .dataloaders import load_K_Rt_from_P
from .image_io import load_image, read_unchanged, save_image
from .meshproc import load_obj


def load_cameras_from_npz(npz_path, n_views, logger=None):
    cams = np.load(npz_path)
    K, R, C = [], [], []
    for k in range(n_views):
        intr, pose = load_K_Rt_from_P(cams["world_mat_{}".format(k)][:3, :])
        K.append(intr[:3, :3])
        R.append(pose[:3, :3])
        C.append(pose[:3, [3]])
    return np.array(K), np.array(R), np.array(C)


def load_cameras_from_transform_json(json_path, albedo_images, logger=None):
    """Frames are matched to the albedo files by stem; `n2w` (if present) takes the cameras back to world space, where
    the saved mesh lives. (albedo_scaling.py:136-210)"""
    with open(json_path) as f:
        data = json.load(f)
    n2w = np.array(data["n2w"], np.float64) if "n2w" in data else None
    by_stem = {}
    for fr in data["frames"]:
        by_stem.setdefault(Path(fr["albedo_path"]).stem, fr)
    K, R, C = [], [], []
    for name in albedo_images:
        fr = by_stem.get(Path(name).stem)
        if fr is None:
            raise RuntimeError("No frame for albedo image: {}".format(name))
        k = np.eye(3, dtype=np.float32)
        if "intrinsic_matrix" in fr:
            k[:3, :3] = np.array(fr["intrinsic_matrix"], np.float32)[:3, :3]
        else:
            gfx = data.get("fl_x")
            k[0, 0] = fr.get("fl_x", gfx or 500.0)
            k[1, 1] = fr.get("fl_y", data.get("fl_y", gfx) or k[0, 0])
            k[0, 2] = fr.get("cx", data.get("cx") or data.get("w", 512) / 2)
            k[1, 2] = fr.get("cy", data.get("cy") or data.get("h", 512) / 2)
        c2w = np.array(fr["transform_matrix"], np.float64)
        if n2w is not None:
            c2w = n2w @ c2w
        K.append(k)
        R.append(c2w[:3, :3].astype(np.float32))
        C.append(c2w[:3, [3]].astype(np.float32))
    return np.array(K), np.array(R), np.array(C)


def load_cameras(camera_source, albedo_images, logger=None):
    p = Path(camera_source)
    if p.suffix == ".npz":
        return load_cameras_from_npz(p, len(albedo_images), logger)
    if p.suffix == ".json":
        return load_cameras_from_transform_json(p, albedo_images, logger)
    if p.suffix == ".sfm":
        raise RuntimeError("sfmData cameras need pyalicevision, which this build does not bundle; pass transform.json or cameras.npz")
    raise ValueError("Unsupported camera format: {}".format(p.suffix))


def _bilinear(img, yx):
    """Linear interpolation of img[(H,W,3)] at fractional (row, col) positions inside the image."""
    y0 = np.clip(np.floor(yx[:, 0]).astype(int), 0, img.shape[0] - 2)
    x0 = np.clip(np.floor(yx[:, 1]).astype(int), 0, img.shape[1] - 2)
    fy, fx = (yx[:, 0] - y0)[:, None], (yx[:, 1] - x0)[:, None]
    return (img[y0, x0] * (1 - fy) * (1 - fx) + img[y0, x0 + 1] * (1 - fy) * fx + img[y0 + 1, x0] * fy * (1 - fx) + img[y0 + 1, x0 + 1] * fy * fx)


def compute_albedo_scale_ratios(albedo_path, camera_source, mesh_path, n_samples=2000, logger=None):
    """-> (n_views, 3) gains, mean 1 per channel. Pixel sampling uses numpy's global RNG (seeded by run_pipeline --seed)."""
    log = logger.info if logger else (lambda m: None)
    names = sorted(f for f in os.listdir(albedo_path) if f.lower().endswith((".png", ".exr")))
    n_views = len(names)
    log("Loading {} albedo images...".format(n_views))
    albedos, masks = [], []
    for name in names:
        img = load_image(os.path.join(albedo_path, name))
        masks.append(img[:, :, 3] if img.shape[2] == 4 else np.ones(img.shape[:2]))
        albedos.append(img[:, :, :3])
    albedos, masks = np.array(albedos), np.array(masks)
    h, w = albedos.shape[1:3]
    K, R, C = load_cameras(camera_source, names, logger)
    log("Loading mesh from {}...".format(mesh_path))
    mesh = load_obj(mesh_path)
    caster = hostlib.MeshRayCaster(mesh.vertices, mesh.faces)
    ratios = np.zeros((n_views, n_samples, 3, 2), np.float32)
    found = np.zeros((n_views, n_samples, 2), bool)
    log("Computing ratios between neighboring views...")
    for cam in range(n_views):
        log("Processing camera {}/{}...".format(cam, n_views))
        ys, xs = np.where(masks[cam].astype(bool))
        n_good = min(n_samples, len(xs))
        if n_good < n_samples:
            log("Warning: only {} valid pixels in image {}".format(n_good, cam))
        pick = np.random.choice(len(xs), n_good, replace=False)
        px = np.stack([xs[pick], ys[pick], np.ones(n_good)], axis=0).astype(np.float64)
        seen = albedos[cam, ys[pick], xs[pick], :]
        origin = C[cam].astype(np.float64).T
        d = (R[cam].astype(np.float64) @ (np.linalg.inv(K[cam].astype(np.float64)) @ px)).T
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        t, tri = caster.first_hit(np.repeat(origin, n_good, axis=0), d)
        ray = np.nonzero(tri >= 0)[0]
        pts = origin + t[ray, None] * d[ray]
        seen = seen[ray]
        for side, nb in enumerate(((cam + 1) % n_views, (cam - 1) % n_views)):
            to_nb = C[nb].astype(np.float64).T - pts
            dist = np.linalg.norm(to_nb, axis=1)
            to_nb /= dist[:, None]
            eps = np.maximum(dist * 1e-4, 1e-2)
            start = pts + eps[:, None] * to_nb
            vis = ~caster.occluded(start, to_nb, dist - eps) if len(pts) else np.zeros(0, bool)
            pc = R[nb].astype(np.float64).T @ (pts[vis].T - C[nb].astype(np.float64))
            uv = (K[nb].astype(np.float64) @ pc).T
            uv = uv[:, :2] / uv[:, 2:3]
            inside = (0 <= uv[:, 1]) & (uv[:, 1] < h - 1) & (0 <= uv[:, 0]) & (uv[:, 0] < w - 1)
            other = _bilinear(albedos[nb].astype(np.float32), uv[inside][:, ::-1])
            keep = ~np.any(other == 0, axis=1)
            slot = ray[vis][inside][keep]
            ratios[cam, slot, :, side] = seen[vis][inside][keep] / other[keep]
            found[cam, slot, side] = True
    log("Computing final scaling factors...")
    # ratio view i -> i+1 from both directions: i's "right" samples and (i+1)'s "left" samples inverted
    left = np.roll(ratios[:, :, :, 1], -1, axis=0)
    left_found = np.roll(found[:, :, 1], -1, axis=0)
    chain = np.ones((n_views, 3))
    for i in range(n_views - 1):
        both = np.concatenate([ratios[i, found[i, :, 0], :, 0], 1.0 / left[i, left_found[i]]], axis=0)
        chain[i + 1] = chain[i] * np.median(both, axis=0)
    gains = chain / chain.mean(axis=0)
    log("Scale ratios: {}".format(gains))
    return gains


def scale_and_save_albedos(albedo_path, output_albedo_path, scale_ratios, bit_depth=None, logger=None):
    """Multiply each view's RGB by its gain (alpha untouched) and write PNGs of the same depth. (albedo_scaling.py:399-436)"""
    log = logger.info if logger else (lambda m: None)
    os.makedirs(output_albedo_path, exist_ok=True)
    names = sorted(f for f in os.listdir(albedo_path) if f.lower().endswith((".png", ".exr")))
    if bit_depth is None:
        first = read_unchanged(os.path.join(albedo_path, names[0]))
        bit_depth = 8 if first.dtype == np.uint8 else 16
        log("Auto-detected bit depth: {}".format(bit_depth))
    log("Scaling {} albedos ({}bit)...".format(len(names), bit_depth))
    for i, name in enumerate(names):
        img = load_image(os.path.join(albedo_path, name))
        alpha = img[:, :, 3] if img.shape[2] == 4 else np.ones(img.shape[:2], np.float32)
        save_image(np.dstack([img[:, :, :3] * scale_ratios[i], alpha]), os.path.join(output_albedo_path, name), bit_depth=bit_depth)
        log("Saved {}/{}: {}".format(i + 1, len(names), name))
