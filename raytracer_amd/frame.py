"""Headless presentation of a rendered frame.

The reference blits its framebuffer to the Windows console with SetDIBitsToDevice
(src/screen.h:81-89) using a 24-bit BI_RGB bitmap header with a positive height
(src/screen.h:102-112): rows bottom-up, bytes B,G,R, which is exactly the layout the
tracer's `bgr` output has (row 0 = bottom scanline).  Here the same bytes go to a file.
"""
from __future__ import annotations

import struct

import numpy as np


def write_bmp(path: str, bgr: np.ndarray) -> None:
    """24-bit bottom-up BMP.  bgr: [height, width, 3] uint8, row 0 = bottom scanline, as
    rendered; rows are padded to a multiple of 4 bytes as the format requires."""
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w, c = bgr.shape
    assert c == 3
    pad = (-3 * w) % 4
    row_bytes = 3 * w + pad
    image_size = row_bytes * h
    header = struct.pack("<2sIHHI", b"BM", 14 + 40 + image_size, 0, 0, 14 + 40)
    info = struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, image_size, 2835, 2835, 0, 0)
    with open(path, "wb") as f:
        f.write(header)
        f.write(info)
        if pad == 0:
            f.write(bgr.tobytes())
        else:
            rows = np.zeros((h, row_bytes), dtype=np.uint8)
            rows[:, : 3 * w] = bgr.reshape(h, 3 * w)
            f.write(rows.tobytes())


def read_bmp(path: str) -> np.ndarray:
    """Inverse of write_bmp (24-bit, uncompressed, bottom-up only)."""
    with open(path, "rb") as f:
        data = f.read()
    magic, _size, _r1, _r2, offset = struct.unpack_from("<2sIHHI", data, 0)
    (_isz, w, h, _planes, bits, comp) = struct.unpack_from("<IiiHHI", data, 14)
    if magic != b"BM" or bits != 24 or comp != 0 or h <= 0:
        raise ValueError("not a 24-bit bottom-up BI_RGB bitmap")
    row_bytes = 3 * w + ((-3 * w) % 4)
    rows = np.frombuffer(data, dtype=np.uint8, count=row_bytes * h, offset=offset).reshape(h, row_bytes)
    return rows[:, : 3 * w].reshape(h, w, 3).copy()


def write_ppm(path: str, bgr: np.ndarray) -> None:
    """Binary PPM (P6): top-down RGB, so the rows are flipped and the channels swapped."""
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w, _ = bgr.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(bgr[::-1, :, ::-1]).tobytes())
