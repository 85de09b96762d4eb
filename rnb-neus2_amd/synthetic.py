"""Synthetic normal+mask dataset of SURVEY.md §8d: an analytic sphere (optionally with a 3-lobe bump) seen by
pinhole cameras on a Fibonacci sphere, rendered straight into the reference's wire format (RGBA16 normal maps with
alpha = mask, white albedo maps; PREP:192-209) in NGP space (what nerf_loader.cu hands to the GPU after applying
``scale``/``offset``): unit-cube scene, object centre (0.5,0.5,0.5).
"""
import numpy as np


def fibonacci_sphere(n):
    i = np.arange(n, dtype=np.float64) + 0.5
    phi = np.arccos(1.0 - 2.0 * i / n)
    theta = np.pi * (1.0 + 5.0 ** 0.5) * i
    return np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)


def look_at_c2w(eye, target):
    """OpenCV-style camera-to-world: x right, y down, z forward (nerf_loader.h:180-188 leaves these axes untouched)."""
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, 0.0, 1.0])
    if abs(fwd @ up) > 0.99:
        up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    m = np.zeros((3, 4), dtype=np.float64)
    m[:, 0], m[:, 1], m[:, 2], m[:, 3] = right, down, fwd, eye
    return m


def _srgb_to_linear(s):
    return np.where(s <= 0.04045, s / 12.92, ((s + 0.055) / 1.055) ** 2.4)


def render_view(c2w, res, fx, radius=0.25, bump=0.0, center=(0.5, 0.5, 0.5)):
    """Returns (normal_rgba16, albedo_rgba16), each [res, res, 4] uint16."""
    h = w = res
    cx = cy = res * 0.5
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64) + 0.5, np.arange(w, dtype=np.float64) + 0.5, indexing="ij")
    dcam = np.stack([(xs - cx) / fx, (ys - cy) / fx, np.ones_like(xs)], axis=-1)
    R = c2w[:, :3]
    o = c2w[:, 3]
    d = dcam @ R.T
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    oc = o - np.asarray(center)
    b = d @ oc
    cterm = oc @ oc - radius * radius
    disc = b * b - cterm
    hit = disc > 0
    t = -b - np.sqrt(np.where(hit, disc, 0.0))
    p = o + t[..., None] * d
    n = (p - np.asarray(center)) / radius
    if bump:
        # r(theta,phi) = R + bump*sin(3 theta) sin(3 phi): one Newton-free correction of the normal is enough for a test scene
        th = np.arctan2(n[..., 1], n[..., 0])
        ph = np.arccos(np.clip(n[..., 2], -1, 1))
        g = bump / radius * np.stack([-3 * np.cos(3 * th) * np.sin(3 * ph) * np.sin(th), 3 * np.cos(3 * th) * np.sin(3 * ph) * np.cos(th), -3 * np.sin(3 * th) * np.cos(3 * ph) * np.sin(ph)], axis=-1)
        n = n - g
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
    ncam = n @ R  # R^T n
    m = np.stack([ncam[..., 0], -ncam[..., 1], -ncam[..., 2]], axis=-1)
    enc = np.clip(np.rint((m + 1.0) * 0.5 * 65535.0), 0, 65535)
    alpha = np.where(hit, 65535, 0)
    normal = np.zeros((h, w, 4), dtype=np.uint16)
    normal[..., :3] = np.where(hit[..., None], enc, 0).astype(np.uint16)
    normal[..., 3] = alpha
    albedo = np.zeros((h, w, 4), dtype=np.uint16)
    albedo[..., :3] = 65535
    albedo[..., 3] = alpha
    return normal, albedo


def make_scene(n_views=64, res=800, fx=None, cam_radius=1.5, radius=0.25, bump=0.0):
    """Config 4 of BASELINE.json by default (64 views, 800x800, fx=1400); config 1 = make_scene(1, 256, 448)."""
    if fx is None:
        fx = 1400.0 * res / 800.0
    center = np.array([0.5, 0.5, 0.5])
    dirs = fibonacci_sphere(n_views)
    views, normals, albedos = [], [], []
    for k in range(n_views):
        c2w = look_at_c2w(center + cam_radius * dirs[k], center)
        nm, al = render_view(c2w, res, fx, radius=radius, bump=bump, center=center)
        views.append(dict(width=res, height=res, focal_length=(fx, fx), principal_point=(0.5, 0.5), xform=c2w.astype(np.float32)))
        normals.append(nm)
        albedos.append(al)
    return views, normals, albedos


def write_png16(path, rgba16):
    """RGBA 16-bit PNG (colour type 6, depth 16, no interlace) — the format of the scenes' normal/albedo maps."""
    import struct
    import zlib

    h, w, c = rgba16.shape
    assert c == 4 and rgba16.dtype == np.uint16
    raw = np.empty((h, 1 + w * 8), np.uint8)
    raw[:, 0] = 0
    raw[:, 1:] = rgba16.astype(">u2").view(np.uint8).reshape(h, w * 8)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw.tobytes(), 1)) + chunk(b"IEND", b""))


def write_scene(path, views, normals, albedos, scale=1.0, offset=(0.0, 0.0, 0.0), n2w=None):
    """Write the on-disk scene format the loader reads (src/nerf_loader.cu:225-764): transform.json with `from_na`,
    per-frame transform_matrix / intrinsic_matrix / normal_path / albedo_path, plus the PNG16 maps. With `from_na` the
    loader keeps the rotation and maps the position to pos*scale+offset, so the inverse is applied here."""
    import json
    import os

    os.makedirs(os.path.join(path, "normal"), exist_ok=True)
    os.makedirs(os.path.join(path, "albedo"), exist_ok=True)
    frames = []
    for i, (v, nm, al) in enumerate(zip(views, normals, albedos)):
        m = np.eye(4)
        m[:3, :4] = np.asarray(v["xform"], np.float64).reshape(3, 4)
        m[:3, 3] = (m[:3, 3] - np.asarray(offset, np.float64)) / scale
        k = np.eye(4)
        k[0, 0], k[1, 1] = v["focal_length"]
        k[0, 2] = v["principal_point"][0] * v["width"]
        k[1, 2] = v["principal_point"][1] * v["height"]
        write_png16(os.path.join(path, "normal", f"{i:03d}.png"), np.ascontiguousarray(nm).reshape(v["height"], v["width"], 4))
        write_png16(os.path.join(path, "albedo", f"{i:03d}.png"), np.ascontiguousarray(al).reshape(v["height"], v["width"], 4))
        frames.append(dict(normal_path=f"normal/{i:03d}.png", albedo_path=f"albedo/{i:03d}.png", transform_matrix=m.tolist(), intrinsic_matrix=k.tolist()))
    meta = dict(from_na=True, w=views[0]["width"], h=views[0]["height"], aabb_scale=1.0, scale=scale, offset=list(offset), frames=frames)
    if n2w is not None:
        meta["n2w"] = np.asarray(n2w).tolist()
    with open(os.path.join(path, "transform.json"), "w") as f:
        json.dump(meta, f)
