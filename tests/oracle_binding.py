"""ctypes binding of oracle/liblrp_oracle.so — the CPU restatement of the
reference hot path.  Test infrastructure only (see oracle/lrp_oracle.h)."""
import ctypes
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "liblrp_oracle.so")


class OLens(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int32), ("u", ctypes.c_float * 4), ("sensor_width", ctypes.c_float),
                ("sensor_height", ctypes.c_float)]


class OImage(ctypes.Structure):
    _fields_ = [("lens", OLens), ("width", ctypes.c_int32), ("height", ctypes.c_int32), ("channels", ctypes.c_int32),
                ("data", ctypes.c_void_p), ("data_layout", ctypes.c_int32)]


assert ctypes.sizeof(OLens) == 28 and ctypes.sizeof(OImage) == 56

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = ctypes.CDLL(LIB_PATH)
        P = ctypes.POINTER
        L.lrpo_reproject_rows.restype = ctypes.c_int
        L.lrpo_reproject_rows.argtypes = [P(OImage), P(OImage), ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_int, ctypes.c_int]
        L.lrpo_reproject.restype = ctypes.c_int
        L.lrpo_reproject.argtypes = [P(OImage), P(OImage), ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.lrpo_post_process.restype = None
        L.lrpo_post_process.argtypes = [P(OImage), ctypes.c_float, ctypes.c_float]
        L.lrpo_source_coords.restype = ctypes.c_int
        L.lrpo_source_coords.argtypes = [P(OImage), P(OImage), ctypes.c_void_p, ctypes.c_void_p]
        L.lrpo_rotation_matrix.restype = None
        L.lrpo_rotation_matrix.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_float, P(ctypes.c_float)]
        L.lrpo_synth_value.restype = ctypes.c_float
        L.lrpo_synth_value.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int]
        L.lrpo_synth_fill.restype = None
        L.lrpo_synth_fill.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                                      ctypes.c_int]
        L.lrpo_checksum.restype = ctypes.c_uint64
        L.lrpo_checksum.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        _lib = L
    return _lib


def _lens(l):
    """l: product LensInfo-like (type, params[4], sensor_width, sensor_height)."""
    o = OLens()
    o.type = int(l.type)
    for i in range(4):
        o.u[i] = l.params[i]
    o.sensor_width = l.sensor_width
    o.sensor_height = l.sensor_height
    return o


def _image(lens, width, height, channels, arr):
    o = OImage()
    o.lens = _lens(lens)
    o.width, o.height, o.channels = width, height, channels
    o.data = arr.ctypes.data if arr is not None else None
    o.data_layout = 0
    return o


def _rot(rotation):
    if rotation is None:
        return None, None
    r = np.ascontiguousarray(np.asarray(rotation, dtype=np.float32).reshape(9))
    return r, r.ctypes.data


def reproject(in_lens, src, out_lens, out_w, out_h, num_samples, interpolation, rotation=None, threads=1,
              out=None):
    """Oracle reproject(): src is (H, W, C) float32; returns (out_h, out_w, C).
    threads > 1 splits the output into row bands (rows are independent)."""
    L = lib()
    src = np.ascontiguousarray(src, dtype=np.float32)
    h, w, c = src.shape
    if out is None:
        out = np.full((out_h, out_w, c), np.float32(-12345.0), dtype=np.float32)
    cin = _image(in_lens, w, h, c, src)
    cout = _image(out_lens, out_w, out_h, c, out)
    keep, rp = _rot(rotation)
    if threads <= 1:
        rc = L.lrpo_reproject(ctypes.byref(cin), ctypes.byref(cout), num_samples, int(interpolation), rp)
        if rc != 0:
            raise OracleError(rc)
        return out
    bands = [(out_h * i // threads, out_h * (i + 1) // threads) for i in range(threads)]

    def run(b):
        return L.lrpo_reproject_rows(ctypes.byref(cin), ctypes.byref(cout), num_samples, int(interpolation), rp, b[0],
                                     b[1])

    with ThreadPoolExecutor(threads) as ex:
        for rc in ex.map(run, bands):
            if rc != 0:
                raise OracleError(rc)
    return out


class OracleError(RuntimeError):
    def __init__(self, rc):
        self.status = rc
        super().__init__({1: "Output lens type not supported.", 2: "Input lens type not supported.",
                          3: "Interpolation method not supported."}.get(rc, f"oracle error {rc}"))


def post_process(img, exposure, reinhard):
    """In place on an (H, W, C) float32 array."""
    h, w, c = img.shape

    class _L:
        type, params, sensor_width, sensor_height = 0, [0, 0, 0, 0], 0.0, 0.0

    o = _image(_L, w, h, c, img)
    lib().lrpo_post_process(ctypes.byref(o), exposure, reinhard)
    return img


def source_coords(in_lens, in_w, in_h, out_lens, out_w, out_h, rotation=None):
    cin = _image(in_lens, in_w, in_h, 1, None)
    cout = _image(out_lens, out_w, out_h, 1, None)
    keep, rp = _rot(rotation)
    sxy = np.empty((out_h, out_w, 2), dtype=np.float32)
    rc = lib().lrpo_source_coords(ctypes.byref(cin), ctypes.byref(cout), rp, sxy.ctypes.data)
    if rc != 0:
        raise OracleError(rc)
    return sxy


def rotation_matrix(pan, pitch, roll):
    out = (ctypes.c_float * 9)()
    lib().lrpo_rotation_matrix(pan, pitch, roll, out)
    return np.array(out, dtype=np.float32)


def synth_frame(width, height, channels, seed, depth_channel=-1):
    a = np.empty((height, width, channels), dtype=np.float32)
    lib().lrpo_synth_fill(a.ctypes.data, width, height, channels, seed & 0xFFFFFFFF, depth_channel)
    return a


def checksum(array):
    """Host twin of lrp_checksum_device (the product's order-independent 64-bit checksum)."""
    a = np.ascontiguousarray(array, dtype=np.float32)
    return int(lib().lrpo_checksum(a.ctypes.data, a.size))


def reproject_rows(in_lens, src, out_lens, out_w, out_h, num_samples, interpolation, rotation, rows):
    """Oracle output rows `rows` of the (out_h, out_w) image: {y: (out_w, C) array}.
    Rows are independent in the reference loop (src/reproject.cpp:284), so this is
    the full-size oracle on a bounded sample."""
    L = lib()
    src = np.ascontiguousarray(src, dtype=np.float32)
    h, w, c = src.shape
    res = {}
    keep, rp = _rot(rotation)
    cin = _image(in_lens, w, h, c, src)
    for y in rows:
        # a one-row buffer placed so that row y of the virtual image lands in it
        row = np.full((1, out_w, c), np.float32(-12345.0), dtype=np.float32)
        cout = _image(out_lens, out_w, out_h, c, None)
        cout.data = row.ctypes.data - y * out_w * c * 4
        rc = L.lrpo_reproject_rows(ctypes.byref(cin), ctypes.byref(cout), num_samples, int(interpolation), rp, y, y + 1)
        if rc != 0:
            raise OracleError(rc)
        res[y] = row[0]
    return res
