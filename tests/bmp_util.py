"""Minimal BMP writer/reader for the driver tests (independent of the driver's own codec)."""
import struct

import numpy as np


def write_bmp24(path, rgb, top_down=False):
    """rgb: [3, H, W] uint8 planar."""
    _, H, W = rgb.shape
    row = (W * 3 + 3) // 4 * 4
    data = bytearray()
    rows = range(H) if top_down else range(H - 1, -1, -1)
    for y in rows:
        line = np.stack([rgb[2, y], rgb[1, y], rgb[0, y]], axis=1).astype(np.uint8).tobytes()
        data += line + b"\0" * (row - W * 3)
    hdr = b"BM" + struct.pack("<IHHI", 54 + len(data), 0, 0, 54)
    hdr += struct.pack("<IiiHHIIiiII", 40, W, -H if top_down else H, 1, 24, 0, len(data), 2835, 2835, 0, 0)
    open(path, "wb").write(hdr + bytes(data))


def write_bmp8(path, gray, palette=None):
    """gray: [H, W] uint8 indices; palette: [256, 3] RGB (default: grey ramp)."""
    H, W = gray.shape
    row = (W + 3) // 4 * 4
    if palette is None:
        palette = np.stack([np.arange(256)] * 3, axis=1)
    pal = b"".join(bytes([int(p[2]), int(p[1]), int(p[0]), 0]) for p in palette)
    data = bytearray()
    for y in range(H - 1, -1, -1):
        data += gray[y].astype(np.uint8).tobytes() + b"\0" * (row - W)
    off = 54 + len(pal)
    hdr = b"BM" + struct.pack("<IHHI", off + len(data), 0, 0, off)
    hdr += struct.pack("<IiiHHIIiiII", 40, W, H, 1, 8, 0, len(data), 2835, 2835, 256, 0)
    open(path, "wb").write(hdr + pal + bytes(data))


def read_bmp8(path):
    """8-bit BMP -> [H, W] uint8 of palette-resolved grey (palette must be grey)."""
    d = open(path, "rb").read()
    assert d[:2] == b"BM"
    off, = struct.unpack_from("<I", d, 10)
    hdr, W, H, planes, bpp, comp = struct.unpack_from("<IiiHHI", d, 14)
    assert bpp == 8 and comp == 0 and H > 0
    pal = np.frombuffer(d, np.uint8, 1024, 14 + hdr).reshape(256, 4)
    assert np.array_equal(pal[:, 0], pal[:, 1]) and np.array_equal(pal[:, 1], pal[:, 2])
    row = (W + 3) // 4 * 4
    px = np.frombuffer(d, np.uint8, row * H, off).reshape(H, row)[::-1, :W]
    return pal[:, 0][px]
