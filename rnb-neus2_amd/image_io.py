"""Image helpers of the preparation stages (mirror of rnb_neus2/image_io.py:15-125; the target image has neither OpenCV nor
OpenEXR: PNG goes through librnb_host.so, EXR through the scanline codec of exr.py). Arrays are RGB(A) everywhere."""
import os
import zlib

from struct import error as struct_error

import numpy as np

from . import exr, hostlib


def read_unchanged(path):
    """What `cv2.imread(path, IMREAD_UNCHANGED)` yields -- uint8 / uint16 for PNG, float32 for EXR -- but in RGB(A) order; None when
    unreadable. (An EXR in a compression this build does not decode raises: silently skipping the frame would hide it.)"""
    if str(path).lower().endswith(".exr"):
        try:
            return exr.read_exr(path)
        except (OSError, ValueError, KeyError, IndexError, struct_error, zlib.error):  # truncated / corrupt file: cv2 would return None
            return None
    try:
        return hostlib.png_read(path)
    except RuntimeError:
        return None


def load_image(path):
    """PNG 8/16-bit -> float32 in [0, 1]; EXR float32 as stored (values may leave [0, 1]); (H, W[, C]) RGB(A).
    (rnb_neus2/image_io.py:15-45)"""
    image = read_unchanged(path)
    if image is None:
        raise FileNotFoundError("Cannot read image: {}".format(path))
    if image.dtype == np.float32:
        return image
    return image.astype(np.float32) / np.float32(255.0 if image.dtype == np.uint8 else 65535.0)


def save_image(image, path, bit_depth=16):
    """float image in [0,1] -> PNG of the given depth; NaN -> 0, clipped, truncating conversion (image_io.py:48-73)."""
    image = np.clip(np.nan_to_num(np.asarray(image), nan=0.0), 0.0, 1.0) * float(2 ** bit_depth - 1)
    hostlib.png_write(path, image.astype(np.uint8 if bit_depth == 8 else np.uint16), level=0)


def save_exr(image, path):
    """float32 RGB image as EXR (FLOAT channels). (image_io.py:73-88)"""
    exr.write_exr(path, np.asarray(image, np.float32))


def load_normal(path):
    """Normal map as float32 in [-1, 1] (PNG stores (n+1)/2). (image_io.py:91-110)"""
    image = load_image(path)
    if image.ndim == 3 and image.shape[2] > 3:
        image = image[:, :, :3]
    if os.path.splitext(str(path))[1].lower() == ".exr":
        return image  # already in [-1, 1]
    return image * 2.0 - 1.0


def save_normal_16bit(normal, path):
    """[-1,1] -> 16-bit PNG. (image_io.py:113-119)"""
    hostlib.png_write(path, np.clip(0.5 * (1.0 + np.asarray(normal)) * 65535.0, 0, 65535).astype(np.uint16), level=0)


def save_normal_exr(normal, path):
    """Raw [-1, 1] values. (image_io.py:123-125)"""
    save_exr(np.asarray(normal, np.float32), path)
