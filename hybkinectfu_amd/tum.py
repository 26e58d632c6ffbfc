"""Writer of synthetic TUM RGB-D style directories (tests and examples): depth/*.png (16-bit, 1/5000 m), rgb/*.png (8-bit RGB),
depth.txt, rgb.txt, groundtruth.txt -- the layout src/DataSourceProducerRGBDDataset.cpp and src/CameraPoseFinderFromFile.cpp read.
PNG encoding is done here with zlib so every scanline filter type can be exercised; nothing in this file is on the product path."""
import os
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def write_png(path, img, filter_type=None, idat_split=0):
    """img: (H, W) uint16 grey, (H, W) uint8 grey, (H, W, 3) uint8 RGB or (H, W, 4) uint8 RGBA.  filter_type: 0..4 for every row,
    None = cycle through all five.  idat_split > 0 cuts the compressed stream into several IDAT chunks."""
    img = np.asarray(img)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    bits = 16 if img.dtype == np.uint16 else 8
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    rows = img.astype(">u2").tobytes() if bits == 16 else img.astype(np.uint8).tobytes()
    bpp = ch * bits // 8
    stride = w * bpp
    raw = bytearray()
    prev = bytes(stride)
    for y in range(h):
        cur = rows[y * stride:(y + 1) * stride]
        ft = (y % 5) if filter_type is None else filter_type
        raw.append(ft)
        if ft == 0:
            raw += cur
        else:
            c = np.frombuffer(cur, np.uint8).astype(np.int32)
            up = np.frombuffer(prev, np.uint8).astype(np.int32)
            left = np.concatenate([np.zeros(bpp, np.int32), c[:-bpp]])
            upleft = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]])
            if ft == 1:
                pred = left
            elif ft == 2:
                pred = up
            elif ft == 3:
                pred = (left + up) >> 1
            else:
                pa, pb, pc = np.abs(up - upleft), np.abs(left - upleft), np.abs(left + up - 2 * upleft)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
            raw += ((c - pred) & 0xFF).astype(np.uint8).tobytes()
        prev = cur
    comp = zlib.compress(bytes(raw), 6)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bits, ctype, 0, 0, 0))
    if idat_split > 0:
        step = max(1, len(comp) // idat_split)
        for i in range(0, len(comp), step):
            out += _chunk(b"IDAT", comp[i:i + step])
    else:
        out += _chunk(b"IDAT", comp)
    out += _chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(out)


def quat_xyzw_from_rotation(R):
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(np.asarray(R, np.float64)).as_quat()


def write_dataset(directory, depth_mm_frames, poses=None, rgb_frames=None, t0=1305031102.175304, dt=1.0 / 30.0, rgb_offset=0.004, raw_scale=5):
    """Frames -> <directory>/ with depth.txt / rgb.txt / groundtruth.txt (three '#' header lines each, as the TUM files have).
    depth PNG value = millimetres * raw_scale (the dataset's 1/5000 m unit for raw_scale 5)."""
    os.makedirs(os.path.join(directory, "depth"), exist_ok=True)
    stamps = [t0 + k * dt for k in range(len(depth_mm_frames))]
    with open(os.path.join(directory, "depth.txt"), "w") as f:
        f.write("# depth maps\n# file: 'synthetic'\n# timestamp filename\n")
        for k, mm in enumerate(depth_mm_frames):
            name = "depth/%.6f.png" % stamps[k]
            write_png(os.path.join(directory, name), (np.asarray(mm, np.uint32) * raw_scale).astype(np.uint16))
            f.write("%.6f %s\n" % (stamps[k], name))
    if rgb_frames is not None:
        os.makedirs(os.path.join(directory, "rgb"), exist_ok=True)
        with open(os.path.join(directory, "rgb.txt"), "w") as f:
            f.write("# color images\n# file: 'synthetic'\n# timestamp filename\n")
            for k, img in enumerate(rgb_frames):
                ts = stamps[k] - rgb_offset
                name = "rgb/%.6f.png" % ts
                write_png(os.path.join(directory, name), np.asarray(img, np.uint8))
                f.write("%.6f %s\n" % (ts, name))
    if poses is not None:
        with open(os.path.join(directory, "groundtruth.txt"), "w") as f:
            f.write("# ground truth trajectory\n# file: 'synthetic'\n# timestamp tx ty tz qx qy qz qw\n")
            for k, p in enumerate(poses):
                p = np.asarray(p, np.float64)
                q = quat_xyzw_from_rotation(p[:3, :3])
                f.write("%.4f %.6f %.6f %.6f %.7f %.7f %.7f %.7f\n" % (stamps[k] + 0.001, p[0, 3], p[1, 3], p[2, 3], q[0], q[1], q[2], q[3]))
    return stamps
