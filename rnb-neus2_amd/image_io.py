"""Image helpers of the preparation stages (mirror of rnb_neus2/image_io.py:15-125, PNG only: the target image has
neither OpenCV nor OpenEXR, decoding goes through librnb_host.so). Arrays are RGB(A) everywhere."""
import os

import numpy as np

from . import hostlib


def _no_exr(path):
    raise NotImplementedError("EXR I/O is not available in this build (no OpenEXR codec on the target image): {}".format(path))


def read_unchanged(path):
    """What `cv2.imread(path, IMREAD_UNCHANGED)` yields for a PNG, but in RGB(A) order; None when unreadable."""
    if str(path).lower().endswith(".exr"):
        _no_exr(path)
    try:
        return hostlib.png_read(path)
    except RuntimeError:
        return None


def load_image(path):
    """PNG 8/16-bit -> float32 in [0, 1], (H, W[, C]) RGB(A). (rnb_neus2/image_io.py:15-45)"""
    image = read_unchanged(path)
    if image is None:
        raise FileNotFoundError("Cannot read image: {}".format(path))
    return image.astype(np.float32) / np.float32(255.0 if image.dtype == np.uint8 else 65535.0)


def save_image(image, path, bit_depth=16):
    """float image in [0,1] -> PNG of the given depth; NaN -> 0, clipped, truncating conversion (image_io.py:48-73)."""
    image = np.clip(np.nan_to_num(np.asarray(image), nan=0.0), 0.0, 1.0) * float(2 ** bit_depth - 1)
    hostlib.png_write(path, image.astype(np.uint8 if bit_depth == 8 else np.uint16), level=0)


def save_exr(image, path):
    _no_exr(path)


def load_normal(path):
    """Normal map as float32 in [-1, 1] (PNG stores (n+1)/2). (image_io.py:91-110)"""
    if os.path.splitext(str(path))[1].lower() == ".exr":
        _no_exr(path)
    image = load_image(path)
    if image.ndim == 3 and image.shape[2] > 3:
        image = image[:, :, :3]
    return image * 2.0 - 1.0


def save_normal_16bit(normal, path):
    """[-1,1] -> 16-bit PNG. (image_io.py:113-119)"""
    hostlib.png_write(path, np.clip(0.5 * (1.0 + np.asarray(normal)) * 65535.0, 0, 65535).astype(np.uint16), level=0)


def save_normal_exr(normal, path):
    _no_exr(path)
