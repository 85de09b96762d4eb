"""ctypes binding of librnb_host.so (include/rnb_host.h): PNG I/O and triangle-mesh ray casting for the data-preparation
and albedo-scaling stages — the jobs the reference gives to cv2 and trimesh/embree."""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def library_path():
    return os.path.join(_PKG, "librnb_host.so")


def load():
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError("librnb_host.so is missing (run `python -m rnb_neus2_amd.build`)")
        lib = C.CDLL(path)
        lib.rnb_host_last_error.restype = C.c_char_p
        lib.rnb_png_info.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.rnb_png_read_rgba16.argtypes = [C.c_char_p, C.c_void_p]
        lib.rnb_png_write.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int32]
        lib.rnb_bvh_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        lib.rnb_bvh_destroy.argtypes = [C.c_void_p]
        lib.rnb_bvh_first_hit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.rnb_bvh_occluded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _LIB = lib
    return _LIB


def _check(rc):
    if rc != 0:
        raise RuntimeError(load().rnb_host_last_error().decode())


def png_info(path):
    w, h, c, d = C.c_uint32(), C.c_uint32(), C.c_int32(), C.c_int32()
    _check(load().rnb_png_info(os.fsencode(str(path)), C.byref(w), C.byref(h), C.byref(c), C.byref(d)))
    return w.value, h.value, c.value, d.value


def png_read_rgba16(path):
    """(H, W, 4) uint16 with the loader's widening rules (8-bit v -> v*257, grey replicated, alpha 65535 if absent)."""
    w, h, _, _ = png_info(path)
    out = np.empty((h, w, 4), np.uint16)
    _check(load().rnb_png_read_rgba16(os.fsencode(str(path)), out.ctypes.data))
    return out


def png_read(path):
    """Decode keeping the file's own layout: uint8 or uint16; (H, W) for grey, (H, W, C) otherwise; RGB(A) order."""
    _, _, channels, depth = png_info(path)
    rgba = png_read_rgba16(path)
    if depth == 8:
        rgba = (rgba // 257).astype(np.uint8)
    if channels == 1:
        return np.ascontiguousarray(rgba[:, :, 0])
    if channels == 2:
        return np.ascontiguousarray(rgba[:, :, [0, 3]])
    return np.ascontiguousarray(rgba[:, :, :channels])


def png_write(path, image, level=1):
    image = np.ascontiguousarray(image)
    if image.dtype not in (np.uint8, np.uint16):
        raise ValueError("png_write needs uint8 or uint16 samples, got {}".format(image.dtype))
    h, w = image.shape[:2]
    channels = 1 if image.ndim == 2 else image.shape[2]
    _check(load().rnb_png_write(os.fsencode(str(path)), image.ctypes.data, w, h, channels, 8 * image.dtype.itemsize, level))


class MeshRayCaster:
    """Nearest-hit / occlusion queries against a triangle mesh."""

    def __init__(self, vertices, triangles):
        self._v = np.ascontiguousarray(vertices, np.float32)
        self._t = np.ascontiguousarray(triangles, np.uint32)
        self._h = C.c_void_p()
        _check(load().rnb_bvh_create(self._v.ctypes.data, len(self._v), self._t.ctypes.data, len(self._t), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            load().rnb_bvh_destroy(self._h)
            self._h = None

    def first_hit(self, origins, directions):
        """-> (t, triangle): t = +inf and triangle = -1 where the ray misses."""
        o = np.ascontiguousarray(origins, np.float64)
        d = np.ascontiguousarray(directions, np.float64)
        t = np.empty(len(o), np.float64)
        tri = np.empty(len(o), np.int32)
        _check(load().rnb_bvh_first_hit(self._h, o.ctypes.data, d.ctypes.data, len(o), t.ctypes.data, tri.ctypes.data))
        return t, tri

    def occluded(self, origins, directions, t_max):
        o = np.ascontiguousarray(origins, np.float64)
        d = np.ascontiguousarray(directions, np.float64)
        tm = np.ascontiguousarray(np.broadcast_to(t_max, (len(o),)), np.float64)
        out = np.empty(len(o), np.uint8)
        _check(load().rnb_bvh_occluded(self._h, o.ctypes.data, d.ctypes.data, tm.ctypes.data, len(o), out.ctypes.data))
        return out.astype(bool)
