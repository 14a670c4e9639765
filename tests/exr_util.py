"""A minimal, independent OpenEXR scanline reader / writer for the CLI tests (numpy + zlib):
HALF / FLOAT channels, NO / ZIPS / ZIP compression.  Written from the file-format layout, not
from cli/lrp_image_io.cpp, so that the two implementations check each other."""
import struct
import zlib

import numpy as np


def _attr(name, typ, payload):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


def write_exr(path, channels, compression=0):
    """channels: dict name -> 2-D array (float16 stored as HALF, float32 as FLOAT)."""
    names = sorted(channels)
    h, w = channels[names[0]].shape
    chlist = b""
    for n in names:
        t = 1 if channels[n].dtype == np.float16 else 2
        chlist += n.encode() + b"\0" + struct.pack("<iBBBBii", t, 0, 0, 0, 0, 1, 1)
    chlist += b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    head = struct.pack("<II", 20000630, 2)
    head += _attr("channels", "chlist", chlist) + _attr("compression", "compression", bytes([compression]))
    head += _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box)
    head += _attr("lineOrder", "lineOrder", b"\0") + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    head += _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0))
    head += _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines = {0: 1, 2: 1, 3: 16}[compression]
    blocks = []
    for y0 in range(0, h, lines):
        raw = b""
        for y in range(y0, min(h, y0 + lines)):
            for n in names:
                raw += np.ascontiguousarray(channels[n][y]).tobytes()
        data = raw
        if compression:
            a = np.frombuffer(raw, dtype=np.uint8)
            t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
            d = t.copy()
            d[1:] = (t[1:] - t[:-1] + 128) & 0xFF
            z = zlib.compress(d.astype(np.uint8).tobytes(), 6)
            if len(z) < len(raw):
                data = z
        blocks.append(struct.pack("<ii", y0, len(data)) + data)
    off = len(head) + 8 * len(blocks)
    table = b""
    for b in blocks:
        table += struct.pack("<Q", off)
        off += len(b)
    with open(path, "wb") as f:
        f.write(head + table + b"".join(blocks))


def read_exr(path):
    """-> dict name -> 2-D array (float16 or float32 as stored)."""
    b = open(path, "rb").read()
    magic, version = struct.unpack_from("<II", b, 0)
    assert magic == 20000630 and (version & 0xFF) == 2 and not (version & 0x1A00)
    pos = 8
    attrs = {}
    while b[pos] != 0:
        e = b.index(b"\0", pos)
        name = b[pos:e].decode()
        pos = e + 1
        e = b.index(b"\0", pos)
        pos = e + 1
        (size,) = struct.unpack_from("<i", b, pos)
        pos += 4
        attrs[name] = b[pos:pos + size]
        pos += size
    pos += 1
    ch = []
    c = attrs["channels"]
    p = 0
    while c[p] != 0:
        e = c.index(b"\0", p)
        n = c[p:e].decode()
        (t,) = struct.unpack_from("<i", c, e + 1)
        ch.append((n, t))
        p = e + 1 + 16
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    comp = attrs["compression"][0]
    lines = {0: 1, 2: 1, 3: 16}[comp]
    nblocks = (h + lines - 1) // lines
    offs = struct.unpack_from("<%dQ" % nblocks, b, pos)
    out = {n: np.zeros((h, w), dtype=np.float16 if t == 1 else np.float32) for n, t in ch}
    line_bytes = sum(w * (2 if t == 1 else 4) for _, t in ch)
    for o in offs:
        yb, size = struct.unpack_from("<ii", b, o)
        data = b[o + 8:o + 8 + size]
        nl = min(lines, h - (yb - y0))
        want = line_bytes * nl
        if comp and size != want:
            d = np.frombuffer(zlib.decompress(data), dtype=np.uint8).astype(np.int64)
            t = np.zeros_like(d)
            acc = 0
            # prefix sums modulo 256 undo the predictor
            t = (np.cumsum(d - 128) + 128) & 0xFF
            half = (len(t) + 1) // 2
            raw = np.empty(len(t), dtype=np.uint8)
            raw[0::2] = t[:half]
            raw[1::2] = t[half:]
            data = raw.tobytes()
        p = 0
        for l in range(nl):
            for n, t in ch:
                nb = w * (2 if t == 1 else 4)
                out[n][yb - y0 + l] = np.frombuffer(data[p:p + nb], dtype=np.float16 if t == 1 else np.float32)
                p += nb
    return out
