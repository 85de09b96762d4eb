"""Minimal OpenEXR scanline codec for the preparation stages (the reference reads EXR normals / albedos / masks through
cv2 with OPENCV_IO_ENABLE_OPENEXR, rnb_neus2/image_io.py:8-45, prepare.py:161-190; the target image has neither OpenCV nor
OpenEXR). Single-part scanline files, channels HALF / FLOAT / UINT, compression NONE / ZIPS / ZIP (what cv2, Blender and
Meshroom write by default); PIZ, PXR24, B44, DWA and tiled / multi-part files are refused with a clear message.
Arrays are (H, W[, C]) float32 in RGB(A) order (cv2 hands out BGR(A); everything in this package is RGB)."""
import struct
import zlib

import numpy as np

_MAGIC = 20000630
_PIXEL = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}
_LINES = {0: 1, 2: 1, 3: 16}  # NONE, ZIPS, ZIP
_COMPRESSION_NAMES = {1: "RLE", 4: "PIZ", 5: "PXR24", 6: "B44", 7: "B44A", 8: "DWAA", 9: "DWAB"}


def _cstr(buf, pos):
    end = buf.index(b"\0", pos)
    return buf[pos:end].decode("latin-1"), end + 1


def _unzip_block(data, n_raw):
    if len(data) == n_raw:  # stored raw when deflate did not help
        return np.frombuffer(data, np.uint8)
    d = np.frombuffer(zlib.decompress(data), np.uint8).copy()
    if d.size != n_raw:
        raise ValueError("EXR: block inflates to %d bytes, expected %d" % (d.size, n_raw))
    d[1:] -= 128          # predictor: d[i] = d[i-1] + d[i] - 128 (mod 256) ...
    d = np.cumsum(d, dtype=np.uint8)  # ... as a wrapping prefix sum
    half = (n_raw + 1) // 2
    out = np.empty(n_raw, np.uint8)   # bytes were split into even / odd halves
    out[0::2] = d[:half]
    out[1::2] = d[half:]
    return out


def read_exr(path):
    with open(path, "rb") as f:
        buf = f.read()
    magic, version = struct.unpack_from("<ii", buf, 0)
    if magic != _MAGIC:
        raise ValueError("not an OpenEXR file: {}".format(path))
    if version & 0x200 or version & 0x1000 or version & 0x800:
        raise NotImplementedError("EXR: tiled / multi-part / deep files are not supported: {}".format(path))
    pos, attrs = 8, {}
    while True:
        name, pos = _cstr(buf, pos)
        if not name:
            break
        typ, pos = _cstr(buf, pos)
        (size,) = struct.unpack_from("<i", buf, pos)
        attrs[name] = (typ, buf[pos + 4:pos + 4 + size])
        pos += 4 + size
    channels, cp, cbuf = [], 0, attrs["channels"][1]
    while cbuf[cp] != 0:
        cname, cp = _cstr(cbuf, cp)
        ptype, _lin, xs, ys = struct.unpack_from("<iB3xii", cbuf, cp)
        cp += 16
        if xs != 1 or ys != 1:
            raise NotImplementedError("EXR: subsampled channels are not supported: {}".format(path))
        channels.append((cname, _PIXEL[ptype]))
    compression = attrs["compression"][1][0]
    if compression not in _LINES:
        raise NotImplementedError("EXR: compression {} is not supported (NONE, ZIPS, ZIP are): {}".format(_COMPRESSION_NAMES.get(compression, compression), path))
    xmin, ymin, xmax, ymax = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = xmax - xmin + 1, ymax - ymin + 1
    lines = _LINES[compression]
    n_chunks = (h + lines - 1) // lines
    offsets = struct.unpack_from("<%dQ" % n_chunks, buf, pos)
    row_bytes = sum(dt.itemsize for _, dt in channels) * w
    planes = {c: np.empty((h, w), np.float32) for c, _ in channels}
    for off in offsets:
        y, size = struct.unpack_from("<ii", buf, off)
        y0 = y - ymin
        n = min(lines, h - y0)
        raw = buf[off + 8:off + 8 + size]
        block = np.frombuffer(raw, np.uint8) if compression == 0 else _unzip_block(raw, n * row_bytes)
        p = 0
        for r in range(n):  # per scanline: the channels one after the other (alphabetical order of the header)
            for cname, dt in channels:
                nb = dt.itemsize * w
                planes[cname][y0 + r] = block[p:p + nb].view(dt).astype(np.float32)
                p += nb
    names = [c for c, _ in channels]
    if all(k in planes for k in "RGB"):
        order = ["R", "G", "B"] + (["A"] if "A" in planes else [])
    elif len(names) == 1:
        return planes[names[0]]
    else:
        order = names
    return np.stack([planes[k] for k in order], axis=-1)


def write_exr(path, image, compression="zip", pixel_type="float"):
    """float32 (H, W), (H, W, 3) RGB or (H, W, 4) RGBA -> FLOAT (or HALF) channels, scanline file (what
    cv2.imwrite(..., IMWRITE_EXR_TYPE_FLOAT) stores)."""
    img = np.asarray(image, np.float32)
    if img.ndim == 2:
        planes = {"Y": img}
    elif img.shape[2] in (3, 4):
        planes = dict(zip("RGBA", np.moveaxis(img, 2, 0)))
    else:
        raise ValueError("EXR: expected 1, 3 or 4 channels")
    h, w = img.shape[:2]
    names = sorted(planes)
    comp = {"none": 0, "zips": 2, "zip": 3}[compression]
    lines = _LINES[comp]

    def attr(name, typ, value):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(value)) + value

    ptype, dt = (2, "<f4") if pixel_type == "float" else (1, "<f2")
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", ptype, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    header = struct.pack("<ii", _MAGIC, 2) + attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([comp])) + attr("dataWindow", "box2i", box) + \
        attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + \
        attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    chunks = []
    for y0 in range(0, h, lines):
        n = min(lines, h - y0)
        raw = np.concatenate([np.ascontiguousarray(planes[c][y0 + r].astype(dt)).view(np.uint8) for r in range(n) for c in names])
        if comp == 0:
            data = raw.tobytes()
        else:
            t = np.concatenate([raw[0::2], raw[1::2]]).astype(np.int16)
            t[1:] = (t[1:] - t[:-1] + 128) & 255
            z = zlib.compress(t.astype(np.uint8).tobytes(), 6)
            data = z if len(z) < raw.size else raw.tobytes()
        chunks.append(struct.pack("<ii", y0, len(data)) + data)
    table_pos = len(header)
    off, offsets = table_pos + 8 * len(chunks), []
    for c in chunks:
        offsets.append(off)
        off += len(c)
    with open(path, "wb") as f:
        f.write(header + struct.pack("<%dQ" % len(chunks), *offsets) + b"".join(chunks))
