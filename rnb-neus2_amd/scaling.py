"""Scene normalisation: similarity transform that puts the object inside the unit sphere the network is trained in.
Mirror of rnb_neus2/scaling.py (same function names / arguments / return values); the silhouette-contour step that the
reference takes from cv2.findContours is done with scipy.ndimage here."""
import os

import numpy as np
from scipy import ndimage


def _similarity(center, factor):
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] *= factor
    m[:3, 3] = -np.asarray(center) * factor
    return m


def compute_unit_sphere_scaling(points_3d, sphere_scale=1.0):
    """(scene_center, scale_factor, scale_matrix) from a point cloud; the farthest 1 % (by distance to the centroid) is
    ignored. (scaling.py:9-37)"""
    points_3d = np.asarray(points_3d)
    dist = np.linalg.norm(points_3d - points_3d.mean(axis=0), axis=1)
    kept = points_3d[dist <= np.percentile(dist, 99)]
    scene_center = kept.mean(axis=0)
    scale_factor = sphere_scale / np.linalg.norm(kept - scene_center, axis=1).max()
    return scene_center, scale_factor, _similarity(scene_center, scale_factor)


def _centroid_ray(cam, mask):
    """World-space unit ray through the silhouette's centre of mass, or None for an empty mask."""
    com = ndimage.center_of_mass(np.asarray(mask, np.float64))
    if np.any(np.isnan(com)):
        return None
    K = np.array([[cam["fx"], 0, cam["cx"]], [0, cam["fy"], cam["cy"]], [0, 0, 1.0]])
    d = np.linalg.inv(K) @ np.array([com[1], com[0], 1.0])
    n = np.linalg.norm(d)
    if n < 1e-12:
        return None
    return np.asarray(cam["R_cam2world"]) @ (d / n)


def _triangulate_scene_center(cameras, masks):
    """Least-squares point closest to all centroid rays: sum_i (I - m m^T)(c - o_i) = 0. (scaling.py:108-146)"""
    A, b = np.zeros((3, 3)), np.zeros(3)
    for cam, mask in zip(cameras, masks):
        m = _centroid_ray(cam, mask)
        if m is None:
            continue
        P = np.eye(3) - np.outer(m, m)
        A += P
        b += P @ np.asarray(cam["center"])
    try:
        return np.linalg.lstsq(A, b, rcond=None)[0]
    except np.linalg.LinAlgError:
        return np.mean([cam["center"] for cam in cameras], axis=0)


def compute_scaling_from_silhouettes(cameras, masks, sphere_scale=1.0, fg_area_ratio=1.5):
    """Centre by centroid-ray triangulation; radius so that the sphere's summed projected area is `fg_area_ratio` times
    the summed foreground area: r = sqrt(ratio * sum(area) / (pi * sum((fx/Z)^2))). (scaling.py:40-105)"""
    scene_center = _triangulate_scene_center(cameras, masks)
    area, f_over_z2 = 0.0, 0.0
    for cam, mask in zip(cameras, masks):
        area += np.asarray(mask).sum()
        z = (np.asarray(cam["R_cam2world"]).T @ (scene_center - np.asarray(cam["center"])))[2]
        if abs(z) < 1e-8:
            z = 1e-8
        f_over_z2 += (cam["fx"] / z) ** 2
    radius = np.sqrt(fg_area_ratio * area / (np.pi * f_over_z2))
    if radius < 1e-8:
        radius = 1.0
    return scene_center, float(sphere_scale / radius)


def _outer_contour_points(mask):
    """(N, 2) x,y of the outer border pixels of every 8-connected blob (holes filled first) — the pixel set
    cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE) walks."""
    fg = ndimage.binary_fill_holes(np.asarray(mask) > 0.5)
    border = fg & ~ndimage.binary_erosion(fg, structure=ndimage.generate_binary_structure(2, 1), border_value=0)
    ys, xs = np.nonzero(border)
    return np.stack([xs, ys], axis=1).astype(np.float64)


def _convex_hull(points):
    """Andrew's monotone chain; returns the hull vertices."""
    pts = np.unique(points, axis=0)
    if len(pts) < 3:
        return pts

    def half(seq):
        out = []
        for p in seq:
            while len(out) >= 2 and (out[-1][0] - out[-2][0]) * (p[1] - out[-2][1]) - (out[-1][1] - out[-2][1]) * (p[0] - out[-2][0]) <= 0:
                out.pop()
            out.append(p)
        return out

    lower, upper = half(pts), half(pts[::-1])
    return np.array(lower[:-1] + upper[:-1])


def compute_scaling_from_silhouettes_v2(cameras, masks, sphere_scale=1.0, margin_px=20, percentile=99):
    """Smallest sphere whose projection encloses every silhouette contour plus `margin_px`: Nelder-Mead over the centre,
    radius = worst back-projected contour distance. (scaling.py:149-258)"""
    from scipy.optimize import minimize

    start = _triangulate_scene_center(cameras, masks)
    views, max_pts = [], 2000
    for cam, mask in zip(cameras, masks):
        pts = _outer_contour_points(mask)
        if len(pts) < 2:
            continue
        if percentile < 100:
            com = ndimage.center_of_mass(np.asarray(mask, np.float64))
            if not np.any(np.isnan(com)):
                d = np.linalg.norm(pts - np.array([com[1], com[0]]), axis=1)
                pts = pts[d <= np.percentile(d, percentile)]
                if len(pts) == 0:
                    continue
        if len(pts) > max_pts:  # extremal points (hull) + uniform subsample
            hull = _convex_hull(pts)
            pts = np.vstack([hull, pts[::max(1, len(pts) // max(1, max_pts - len(hull)))]])
        R_w2c = np.asarray(cam["R_cam2world"]).T
        views.append(dict(cam=cam, R=R_w2c, t=-R_w2c @ np.asarray(cam["center"]), pts=pts))
    if not views:
        return start, float(sphere_scale)

    def required_radius(c):
        worst = 0.0
        for v in views:
            p = v["R"] @ c + v["t"]
            if p[2] <= 1e-6:
                return 1e12
            cam = v["cam"]
            u0, v0 = cam["fx"] * p[0] / p[2] + cam["cx"], cam["fy"] * p[1] / p[2] + cam["cy"]
            r = np.hypot((v["pts"][:, 0] - u0) * p[2] / cam["fx"], (v["pts"][:, 1] - v0) * p[2] / cam["fy"]).max()
            worst = max(worst, r + margin_px * p[2] / (0.5 * (cam["fx"] + cam["fy"])))
        return worst

    res = minimize(required_radius, start, method="Nelder-Mead", options={"maxiter": 5000, "xatol": 1e-4, "fatol": 1e-6})
    return res.x.astype(np.float32), float(sphere_scale / required_radius(res.x))


def extract_cameras_for_scaling(data, mask_folder_path=""):
    """Scaling-ready (cameras, masks) from a loader dict; views without a readable mask are dropped. (scaling.py:261-305)"""
    from .image_io import read_unchanged

    cameras, masks = [], []
    for view in data["views"]:
        path = view["mask_path"]
        img = read_unchanged(path) if path and os.path.exists(path) else None
        if img is None:
            continue
        if img.ndim == 3:
            img = img[:, :, 0]
        K, c2w = view["K"], view["c2w"]
        cameras.append(dict(fx=float(K[0, 0]), fy=float(K[1, 1]), cx=float(K[0, 2]), cy=float(K[1, 2]),
                            R_cam2world=c2w[:3, :3].astype(np.float64), center=c2w[:3, 3].astype(np.float64)))
        masks.append((img > (125 if img.dtype == np.uint8 else 30000)).astype(np.float32))
    return cameras, masks
