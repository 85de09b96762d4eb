"""Loader dict -> the scene directory the testbed reads: transform.json + normals/NNNNN.png + albedos/NNNNN.png (RGBA,
alpha = mask). Mirror of rnb_neus2/prepare.py:23-257 (PNG inputs; EXR needs a codec the target image lacks)."""
import json
import os

import numpy as np

from .image_io import read_unchanged
from . import hostlib
from .scaling import (compute_scaling_from_silhouettes, compute_scaling_from_silhouettes_v2, compute_unit_sphere_scaling,
                      extract_cameras_for_scaling, _similarity)


def _load_mask_image(mask_path, img_shape, bit_depth):
    """Binary mask as 0 / full-scale samples of the requested depth; all-opaque when there is no mask file.
    Thresholds: >125 for 8-bit masks, >30000 for 16-bit ones, >0.5 for float (EXR) ones. (prepare.py:23-42)"""
    full = 65535 if bit_depth == 16 else 255
    dtype = np.uint16 if bit_depth == 16 else np.uint8
    img = read_unchanged(mask_path) if mask_path and os.path.exists(mask_path) else None
    if img is None:
        return np.full(img_shape, full, dtype)
    if img.ndim == 3:
        img = img[:, :, 2] if img.shape[2] >= 3 else img[:, :, 0]  # cv2's channel 0 is blue; arrays here are RGB(A)
    if img.dtype == np.float32:  # EXR masks
        return np.where(img > 0.5, full, 0).astype(dtype)
    return np.where(img > (125 if img.dtype == np.uint8 else 30000), full, 0).astype(dtype)


def _compute_scaling(data, scaling_mode, sphere_scale, margin_px, logger):
    """Mode cascade: silhouettes(_v2) -> landmarks ("pcd") -> camera centres; "auto" tries them in that order,
    "none" leaves the scene untouched. (prepare.py:45-113)"""
    if scaling_mode == "none":
        return np.zeros(3, np.float32), 1.0, np.eye(4, dtype=np.float32)
    result = None
    if scaling_mode in ("auto", "silhouettes", "silhouettes_v2"):
        cams, masks = extract_cameras_for_scaling(data)
        if cams and masks:
            if scaling_mode == "silhouettes":
                logger.info("Scaling from silhouettes: {} views".format(len(cams)))
                center, factor = compute_scaling_from_silhouettes(cams, masks, sphere_scale=sphere_scale)
            else:
                logger.info("Scaling from silhouettes_v2 (min enclosing sphere): {} views".format(len(cams)))
                center, factor = compute_scaling_from_silhouettes_v2(cams, masks, sphere_scale=sphere_scale, margin_px=margin_px)
            center = np.asarray(center, np.float32)
            result = (center, factor, _similarity(center, factor))
    if result is None and scaling_mode in ("auto", "pcd"):
        landmarks = data.get("landmarks")
        if landmarks is not None and len(landmarks) > 0:
            logger.info("Scaling from landmarks: {} points".format(len(landmarks)))
            result = compute_unit_sphere_scaling(landmarks, sphere_scale)
    if result is None and scaling_mode in ("auto", "cameras"):
        centers = np.array([v["c2w"][:3, 3] for v in data["views"]], np.float32)
        if len(centers):
            logger.info("Scaling from camera centers: {} cameras".format(len(centers)))
            result = compute_unit_sphere_scaling(centers, sphere_scale)
    if result is None:
        raise RuntimeError("No data for scaling. Use scaling_mode='none' to disable.")
    logger.info("Scene center: {}".format(np.asarray(result[0]).tolist()))
    logger.info("Scale factor: {:.6f}".format(result[1]))
    return result


def _rgb(img):
    """Drop alpha, replicate grey."""
    if img.ndim == 2:
        return np.repeat(img[:, :, None], 3, axis=2)
    return img[:, :, :3] if img.shape[2] >= 3 else np.repeat(img[:, :, :1], 3, axis=2)


def prepare_testbed_data(data, output_folder, logger, scaling_mode="auto", sphere_scale=1.0, margin_px=20):
    """Writes the scene; returns dict(scene_center, scale_factor, scale_matrix, n2w, n_frames). Camera centres become
    scale_factor * (c - scene_center); images keep their bit depth, get the mask as alpha (each at its own depth), and a
    missing albedo becomes all-white; unreadable frames are skipped. transform.json carries aabb_scale 1, scale 0.5,
    offset 0.5, from_na and n2w = inverse normalisation. (prepare.py:116-257)"""
    scene_center, scale_factor, scale_matrix = _compute_scaling(data, scaling_mode, sphere_scale, margin_px, logger)
    for sub in ("albedos", "normals"):
        os.makedirs(os.path.join(output_folder, sub), exist_ok=True)
    frames = []
    for idx, view in enumerate(data["views"]):
        c2w = np.array(view["c2w"], copy=True)
        c2w[:3, 3] = scale_factor * (c2w[:3, 3] - scene_center)
        if not os.path.exists(view["normal_path"]):
            logger.warning("Normal not found: {}, skipping".format(view["normal_path"]))
            continue
        normal = read_unchanged(view["normal_path"])
        if normal is None:
            logger.warning("Cannot read: {}".format(view["normal_path"]))
            continue
        if normal.dtype == np.float32:  # EXR normals in [-1, 1] -> 16-bit (prepare.py:166-170)
            normal = (np.clip((normal + 1.0) / 2.0, 0, 1) * 65535).astype(np.uint16)
        normal = _rgb(normal)
        depth = 16 if normal.dtype == np.uint16 else 8
        albedo_path = view.get("albedo_path")
        albedo = read_unchanged(albedo_path) if albedo_path and os.path.exists(albedo_path) else None
        if albedo is not None and albedo.dtype == np.float32:  # EXR albedos (prepare.py:185-188)
            albedo = (np.clip(albedo, 0, 1) * 65535).astype(np.uint16)
        albedo = np.full_like(normal, 65535 if depth == 16 else 255) if albedo is None else _rgb(albedo)
        normal_mask = _load_mask_image(view.get("mask_path"), normal.shape[:2], depth)
        albedo_depth = 16 if albedo.dtype == np.uint16 else 8
        albedo_mask = normal_mask if albedo_depth == depth else _load_mask_image(view.get("mask_path"), albedo.shape[:2], albedo_depth)
        name = "{:05d}.png".format(idx)
        hostlib.png_write(os.path.join(output_folder, "normals", name), np.dstack([normal, normal_mask]))
        hostlib.png_write(os.path.join(output_folder, "albedos", name), np.dstack([albedo, albedo_mask]))
        frames.append(dict(albedo_path="albedos/" + name, normal_path="normals/" + name, transform_matrix=c2w.tolist(), intrinsic_matrix=np.asarray(view["K"]).tolist()))
    if not frames:
        raise RuntimeError("No valid frames could be processed")
    logger.info("Processed {} frames".format(len(frames)))
    n2w = np.linalg.inv(scale_matrix)
    path = os.path.join(output_folder, "transform.json")
    with open(path, "w") as f:
        json.dump(dict(w=data["image_width"], h=data["image_height"], aabb_scale=1.0, scale=0.5, offset=[0.5, 0.5, 0.5], from_na=True, n2w=n2w.tolist(), frames=frames), f, indent=4)
    logger.info("Saved transform.json to {}".format(path))
    return dict(scene_center=scene_center, scale_factor=scale_factor, scale_matrix=scale_matrix, n2w=n2w, n_frames=len(frames))
