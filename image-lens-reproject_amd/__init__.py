"""image-lens-reproject_amd — MI355X-native lens reprojection (hot path only).

Python mirror of the reference operator interface (reference src/reproject.hpp:7-27,
src/config.hpp:7-37): same names and argument meaning —

    reproject(in_image, out_image, num_samples, interpolation, rotation_matrix)
    post_process(image, exposure, reinhard)
    test_conversion_math()

over the C ABI of include/lrp.h (liblrp_hip.so, hand-written HIP for gfx950).
Image data may be a C-contiguous float32 numpy array (host path: upload, kernel,
download) or a float32 torch CUDA tensor (device-resident path, asynchronous on
the given / current torch stream).  There is no CPU implementation here: without
the HIP library and a GPU every compute call raises.

The directory name contains a hyphen (it is the name the build contract fixes);
import it with importlib.import_module("image-lens-reproject_amd").
"""
import ctypes
import enum

import numpy as np

from . import _native
from ._native import LrpImage, LrpLens, LrpPost


class LensType(enum.IntEnum):  # reference src/config.hpp:7-13
    RECTILINEAR = 0
    FISHEYE_EQUIDISTANT = 1
    FISHEYE_EQUISOLID = 2
    FISHEYE_STEREOGRAPHIC = 3
    EQUIRECTANGULAR = 4


class Interpolation(enum.IntEnum):  # reference src/reproject.hpp:16-20
    NEAREST = 0
    BILINEAR = 1
    BICUBIC = 2


class DataLayout(enum.IntEnum):  # reference src/reproject.hpp:7
    RGB = 0
    RGBA = 1
    RGBZ = 2
    RGBAZ = 3


class Status(enum.IntEnum):
    OK = 0
    OUTPUT_LENS = 1
    INPUT_LENS = 2
    INTERPOLATION = 3
    CHANNELS = 4
    BAD_DIMS = 5
    NULL = 6
    NO_DEVICE = 7
    HIP = 8
    OOM = 9
    BAD_ARG = 10


class LrpError(RuntimeError):
    """A non-zero lrp_status.  For the three dispatch failures the message is the
    exact line the reference prints before exit(1) (src/reproject.cpp:365,396,416)."""

    def __init__(self, status, detail=""):
        self.status = Status(status) if status in Status._value2member_map_ else status
        lib = _native.load()
        base = lib.lrp_strerror(int(status)).decode()
        super().__init__(f"{base} [{detail}]" if detail and detail != base else base)


def _check(status):
    if status != 0:
        raise LrpError(status, _native.load().lrp_last_error().decode())


class LensInfo:
    """reference struct LensInfo (src/config.hpp:15-37).  Units: mm, radians."""

    def __init__(self, type, params=(0.0, 0.0, 0.0, 0.0), sensor_width=0.0, sensor_height=0.0):
        self.type = int(type)
        p = list(params) + [0.0] * (4 - len(params))
        self.params = [float(np.float32(v)) for v in p]
        self.sensor_width = float(np.float32(sensor_width))
        self.sensor_height = float(np.float32(sensor_height))

    @classmethod
    def _from_c(cls, c):
        return cls(c.type, list(c.raw), c.sensor_width, c.sensor_height)

    @classmethod
    def rectilinear(cls, focal_length, sensor_width, res_x, res_y):
        """--rectilinear focal_len,sensor_width (src/main.cpp:15-29)."""
        c = LrpLens()
        _native.load().lrp_lens_rectilinear(ctypes.byref(c), focal_length, sensor_width, res_x, res_y)
        return cls._from_c(c)

    @classmethod
    def equidistant(cls, fov):
        """--equidistant fov (src/main.cpp:49-56)."""
        c = LrpLens()
        _native.load().lrp_lens_equidistant(ctypes.byref(c), fov)
        return cls._from_c(c)

    @classmethod
    def equirectangular(cls, longitude_min=None, longitude_max=None, latitude_min=None, latitude_max=None):
        """--equirectangular full | lon_min,lon_max,lat_min,lat_max (src/main.cpp:58-95)."""
        c = LrpLens()
        lib = _native.load()
        if longitude_min is None:
            lib.lrp_lens_equirectangular_full(ctypes.byref(c))
        else:
            lib.lrp_lens_equirectangular(ctypes.byref(c), longitude_min, longitude_max, latitude_min, latitude_max)
        return cls._from_c(c)

    def to_c(self):
        c = LrpLens()
        c.type = self.type
        for i in range(4):
            c.raw[i] = self.params[i]
        c.sensor_width = self.sensor_width
        c.sensor_height = self.sensor_height
        return c


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Image:
    """reference struct Image (src/reproject.hpp:9-14): lens + width/height/channels +
    interleaved float32 data[(y*W + x)*C + c]."""

    def __init__(self, lens, width, height, channels, data=None, data_layout=DataLayout.RGBA):
        self.lens = lens
        self.width = int(width)
        self.height = int(height)
        self.channels = int(channels)
        self.data = data
        self.data_layout = int(data_layout)

    def _ptr(self):
        d = self.data
        if d is None:
            return None
        n = self.width * self.height * self.channels
        if _is_torch(d):
            import torch

            if d.dtype != torch.float32 or not d.is_contiguous() or d.numel() != n:
                raise ValueError("image tensor must be contiguous float32 with width*height*channels elements")
            return d.data_ptr()
        if not isinstance(d, np.ndarray) or d.dtype != np.float32 or not d.flags["C_CONTIGUOUS"] or d.size != n:
            raise ValueError("image array must be C-contiguous float32 with width*height*channels elements")
        return d.ctypes.data

    def on_device(self):
        return _is_torch(self.data) and self.data.is_cuda

    def to_c(self):
        c = LrpImage()
        c.lens = self.lens.to_c()
        c.width, c.height, c.channels = self.width, self.height, self.channels
        c.data = self._ptr()
        c.data_layout = self.data_layout
        return c


def _rotation_arg(rotation_matrix):
    if rotation_matrix is None:
        return None, None
    r = np.ascontiguousarray(np.asarray(rotation_matrix, dtype=np.float32).reshape(9))
    return r, r.ctypes.data


def _stream_handle(stream):
    if stream is None:
        import torch

        return torch.cuda.current_stream().cuda_stream
    if hasattr(stream, "cuda_stream"):
        return stream.cuda_stream
    return int(stream)


def rotation_matrix(pan, pitch, roll):
    """computeRotationMatrix (src/main.cpp:110-142); radians; row-major 9 floats."""
    out = (ctypes.c_float * 9)()
    _native.load().lrp_rotation_matrix(pan, pitch, roll, out)
    return np.array(out, dtype=np.float32)


def device_count():
    return _native.load().lrp_device_count()


PIXEL_KERNEL, TILE_KERNEL, WINDOW_KERNEL = 0, 1, 2


def debug_kernel(choice=-1):
    """lrp_debug_kernel: select the HIP kernel family (0 pixel, 1 tile, 2 tile + LDS-window
    bicubic, 3 the same without shared tap coefficients; all produce the same bits);
    returns the previous choice.  -1 only queries."""
    return _native.load().lrp_debug_kernel(int(choice))


def debug_set(name, value=-1):
    """lrp_debug_set: the named A/B switch ("xsep", "quad", "mirror_modes", "win_edge", "win_split", "geo_cache",
    "batch_frames", "multi_fork", "geo_strip", "geo_big", "geo_lists", "geo_fill_stream", "geo_fill_fused", "kernel"; "listed_launches" is a
    counter); returns the previous value, -1 only queries."""
    prev = _native.load().lrp_debug_set(str(name).encode(), int(value))
    if prev < 0:
        raise ValueError(f"unknown debug switch {name!r}")
    return prev


def geometry_cache_configure(max_bytes=-1, min_sightings=-1):
    """lrp_geometry_cache_configure: bytes per device (0 switches the cache off and frees it), launches of a geometry
    before it is cached; negative values keep the current setting."""
    _check(_native.load().lrp_geometry_cache_configure(int(max_bytes), int(min_sightings)))


def geometry_cache_stats():
    """lrp_geometry_cache_stats as a dict: bytes, max_bytes, entries, fills, hits, bypasses, evictions."""
    info = _native.LrpGeometryCacheInfo()
    _native.load().lrp_geometry_cache_stats(ctypes.byref(info))
    return {n: int(getattr(info, n)) for n, _ in info._fields_}


def release_cached_tables():
    """lrp_release_cached_tables: frees the lens tables and the geometry cache of every device."""
    _native.load().lrp_release_cached_tables()


def reproject(in_image, out_image, num_samples, interpolation, rotation_matrix=None, post=None, device=None,
              stream=None):
    """reproject::reproject (src/reproject.cpp:405-419).  `post=(exposure, reinhard)`
    fuses post_process into the store.  Device tensors: asynchronous on `stream`
    (default: torch's current stream); numpy arrays: synchronous."""
    lib = _native.load()
    cin, cout = in_image.to_c(), out_image.to_c()
    keep, rot = _rotation_arg(rotation_matrix)
    cpost = LrpPost(float(post[0]), float(post[1])) if post is not None else None
    ppost = ctypes.byref(cpost) if cpost is not None else None
    if in_image.on_device() != out_image.on_device():
        raise ValueError("input and output image data must both be on the host or both on the device")
    if in_image.on_device():
        dev = in_image.data.device.index if device is None else device
        st = lib.lrp_reproject_device(ctypes.byref(cin), ctypes.byref(cout), int(num_samples), int(interpolation), rot,
                                      ppost, dev, _stream_handle(stream))
    else:
        st = lib.lrp_reproject(ctypes.byref(cin), ctypes.byref(cout), int(num_samples), int(interpolation), rot, ppost,
                               0 if device is None else device)
    del keep
    _check(st)


def reproject_multi(in_image, out_images, num_samples, interpolation, rotation_matrices=None, post=None, device=None,
                    stream=None):
    """One resident source, several target lenses / rotations (device tensors only)."""
    lib = _native.load()
    cin = in_image.to_c()
    arr = (LrpImage * len(out_images))(*[o.to_c() for o in out_images])
    rot = None
    keep = None
    if rotation_matrices is not None:
        keep = np.ascontiguousarray(np.asarray(rotation_matrices, dtype=np.float32).reshape(len(out_images), 9))
        rot = keep.ctypes.data
    cpost = LrpPost(float(post[0]), float(post[1])) if post is not None else None
    dev = in_image.data.device.index if device is None else device
    st = lib.lrp_reproject_multi_device(ctypes.byref(cin), arr, len(out_images), int(num_samples), int(interpolation),
                                        rot, ctypes.byref(cpost) if cpost is not None else None, dev,
                                        _stream_handle(stream))
    del keep
    _check(st)


def reproject_rows(in_image, out_image, num_samples, interpolation, row_first, row_count, rotation_matrix=None, post=None,
                   device=None, stream=None):
    """lrp_reproject_rows_device: rows [row_first, row_first + row_count) of the output only (device tensors)."""
    lib = _native.load()
    cin, cout = in_image.to_c(), out_image.to_c()
    keep, rot = _rotation_arg(rotation_matrix)
    cpost = LrpPost(float(post[0]), float(post[1])) if post is not None else None
    dev = in_image.data.device.index if device is None else device
    st = lib.lrp_reproject_rows_device(ctypes.byref(cin), ctypes.byref(cout), int(num_samples), int(interpolation), rot,
                                       ctypes.byref(cpost) if cpost is not None else None, int(row_first), int(row_count), dev,
                                       _stream_handle(stream))
    del keep
    _check(st)


def reproject_multi_gpu(in_image, out_images, num_samples, interpolation, rotation_matrices=None, post=None, devices=(0,)):
    """lrp_reproject_multi: one host source, several host outputs, row bands of every output on every listed GPU."""
    lib = _native.load()
    cin = in_image.to_c()
    arr = (LrpImage * len(out_images))(*[o.to_c() for o in out_images])
    rot, keep = None, None
    if rotation_matrices is not None:
        keep = np.ascontiguousarray(np.asarray(rotation_matrices, dtype=np.float32).reshape(len(out_images), 9))
        rot = keep.ctypes.data
    cpost = LrpPost(float(post[0]), float(post[1])) if post is not None else None
    devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
    st = lib.lrp_reproject_multi(ctypes.byref(cin), arr, len(out_images), int(num_samples), int(interpolation), rot,
                                 ctypes.byref(cpost) if cpost is not None else None, devs, len(devices))
    del keep
    _check(st)


def reproject_batch(in_images, out_images, num_samples, interpolation, rotation_matrix=None, post=None, device=None,
                    stream=None):
    """n images of one geometry (same sizes, channels, lenses), device tensors only: one kernel
    launch per 16 images (lrp_reproject_batch_device); results equal n reproject() calls."""
    lib = _native.load()
    if len(in_images) != len(out_images):
        raise ValueError("in_images and out_images must have the same length")
    if not in_images:
        return
    ins = (LrpImage * len(in_images))(*[i.to_c() for i in in_images])
    outs = (LrpImage * len(out_images))(*[o.to_c() for o in out_images])
    rot = None
    keep = None
    if rotation_matrix is not None:
        keep = np.ascontiguousarray(np.asarray(rotation_matrix, dtype=np.float32).reshape(9))
        rot = keep.ctypes.data
    cpost = LrpPost(float(post[0]), float(post[1])) if post is not None else None
    dev = in_images[0].data.device.index if device is None else device
    st = lib.lrp_reproject_batch_device(ins, outs, len(in_images), int(num_samples), int(interpolation), rot,
                                        ctypes.byref(cpost) if cpost is not None else None, dev, _stream_handle(stream))
    del keep
    _check(st)


class PreparedBatch:
    """The arguments of one lrp_reproject_batch_device call, marshalled once: launch() is then a single
    foreign call (a directory of frames of one geometry is launched many times with the same descriptors)."""

    def __init__(self, in_images, out_images, num_samples, interpolation, rotation_matrix=None, post=None, device=None):
        if len(in_images) != len(out_images) or not in_images:
            raise ValueError("in_images and out_images must be non-empty and of equal length")
        self._lib = _native.load()
        self._keep = (list(in_images), list(out_images))
        self._n = len(in_images)
        self._ins = (LrpImage * self._n)(*[i.to_c() for i in in_images])
        self._outs = (LrpImage * self._n)(*[o.to_c() for o in out_images])
        self._rot_keep, self._rot = _rotation_arg(rotation_matrix)
        self._post = LrpPost(float(post[0]), float(post[1])) if post is not None else None
        self._args = (int(num_samples), int(interpolation))
        self._dev = in_images[0].data.device.index if device is None else device

    def launch(self, stream=None):
        _check(self._lib.lrp_reproject_batch_device(self._ins, self._outs, self._n, self._args[0], self._args[1], self._rot,
                                                    ctypes.byref(self._post) if self._post is not None else None, self._dev,
                                                    _stream_handle(stream)))


def post_process(image, exposure, reinhard, device=None, stream=None):
    """reproject::post_process (src/reproject.cpp:421-437), in place."""
    lib = _native.load()
    c = image.to_c()
    if image.on_device():
        dev = image.data.device.index if device is None else device
        st = lib.lrp_post_process_device(ctypes.byref(c), exposure, reinhard, dev, _stream_handle(stream))
    else:
        st = lib.lrp_post_process(ctypes.byref(c), exposure, reinhard, 0 if device is None else device)
    _check(st)


def test_conversion_math():
    """reproject::test_conversion_math (src/reproject.cpp:467): an empty function in the reference."""
    return None


class BatchContext:
    """lrp_context: n_streams streams on one device for batches of independent
    host images (the reference's one-file-per-pool-thread path, src/main.cpp:538-622)."""

    def __init__(self, device=0, n_streams=3):
        self._lib = _native.load()
        self._h = ctypes.c_void_p()
        _check(self._lib.lrp_context_create(ctypes.byref(self._h), device, n_streams))
        self._keep = []

    def submit(self, in_image, out_image, num_samples, interpolation, rotation_matrix=None, post=None):
        cin, cout = in_image.to_c(), out_image.to_c()
        keep, rot = _rotation_arg(rotation_matrix)
        cpost = LrpPost(float(post[0]), float(post[1])) if post is not None else None
        self._keep.append((in_image, out_image, keep))
        _check(self._lib.lrp_context_submit(self._h, ctypes.byref(cin), ctypes.byref(cout), int(num_samples),
                                            int(interpolation), rot, ctypes.byref(cpost) if cpost else None))

    def submit_packed(self, in_image, in_format, in_data, out_image, out_format, out_data, out_fill, num_samples,
                      interpolation, rotation_matrix=None, post=None):
        """lrp_context_submit_packed: `in_data` / `out_data` are C-contiguous numpy arrays of shape
        (height, width, packed_channels) in the file formats (float16 / uint8 / float32); the images
        describe the float geometry the kernels see.  Returns a ticket for wait_ticket()."""
        cin, cout = in_image.to_c(), out_image.to_c()
        cin.data = in_data.ctypes.data
        cout.data = out_data.ctypes.data
        keep, rot = _rotation_arg(rotation_matrix)
        cpost = LrpPost(float(post[0]), float(post[1])) if post is not None else None
        ticket = ctypes.c_int(-1)
        self._keep.append((in_data, out_data, keep))
        _check(self._lib.lrp_context_submit_packed(self._h, ctypes.byref(cin), int(in_format), int(in_data.shape[-1]),
                                                   ctypes.byref(cout), int(out_format), int(out_data.shape[-1]),
                                                   int(out_fill), int(num_samples), int(interpolation), rot,
                                                   ctypes.byref(cpost) if cpost else None, ctypes.byref(ticket)))
        return ticket.value

    def wait_ticket(self, ticket):
        _check(self._lib.lrp_context_wait_ticket(self._h, int(ticket)))

    def wait(self):
        st = self._lib.lrp_context_wait(self._h)
        self._keep.clear()
        _check(st)

    def close(self):
        if self._h:
            self._lib.lrp_context_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def synth_fill(tensor, width, height, channels, seed, depth_channel=-1, stream=None):
    """Fill a float32 CUDA tensor with the synthetic frame of SURVEY.md §8d."""
    lib = _native.load()
    _check(lib.lrp_synth_fill_device(tensor.data_ptr(), width, height, channels, seed & 0xFFFFFFFF, depth_channel,
                                     tensor.device.index, _stream_handle(stream)))


def math_eval(func, a, b=None, stream=None):
    """Run one device math routine element-wise on CUDA tensors (tests)."""
    import torch

    lib = _native.load()
    out = torch.empty_like(a)
    _check(lib.lrp_math_eval_device(int(func), a.data_ptr(), b.data_ptr() if b is not None else None, out.data_ptr(),
                                    a.numel(), a.device.index, _stream_handle(stream)))
    return out


def checksums(tensors, stream=None):
    """Order-independent 64-bit checksums (lrp_checksum_device) of float32 CUDA tensors, one
    device pass per tensor and a single read-back: a list of Python ints."""
    import torch

    lib = _native.load()
    if not tensors:
        return []
    dev = tensors[0].device
    out = torch.zeros(len(tensors), dtype=torch.int64, device=dev)
    for i, t in enumerate(tensors):
        _check(lib.lrp_checksum_device(t.data_ptr(), t.numel(), out.data_ptr() + 8 * i, dev.index, _stream_handle(stream)))
    return [int(v) & 0xFFFFFFFFFFFFFFFF for v in out.cpu().tolist()]


def checksum_host(array):
    """The same checksum evaluated with numpy on a host array (tests)."""
    bits = np.ascontiguousarray(array, dtype=np.float32).reshape(-1).view(np.uint32)
    idx = np.arange(bits.size, dtype=np.uint32)

    def mix32(seed, index):
        with np.errstate(over="ignore"):
            h = index * np.uint32(0x9E3779B9) + seed
            h ^= h >> np.uint32(16)
            h *= np.uint32(0x7FEB352D)
            h ^= h >> np.uint32(15)
            h *= np.uint32(0x846CA68B)
            h ^= h >> np.uint32(16)
        return h

    with np.errstate(over="ignore"):
        lo = mix32(bits, idx).astype(np.uint64)
        hi = mix32(bits ^ np.uint32(0xA5A5A5A5), idx * np.uint32(2) + np.uint32(0x7F4A7C15)).astype(np.uint64)
        return int(np.sum((hi << np.uint64(32)) | lo, dtype=np.uint64))


class PixelFormat(enum.IntEnum):  # include/lrp.h lrp_pixel_format
    F32 = 0
    F16 = 1
    U8_GAMMA = 2


def decode_pixels(src, src_format, dst, stream=None):
    """lrp_decode_pixels_device: `src` a CUDA tensor (..., src_channels) of uint8 / int16 (half bits) / float16 /
    float32 samples, `dst` a float32 CUDA tensor (..., dst_channels) with the same number of pixels."""
    lib = _native.load()
    n = dst.numel() // dst.shape[-1]
    _check(lib.lrp_decode_pixels_device(src.data_ptr(), int(src_format), int(src.shape[-1]), dst.data_ptr(), int(dst.shape[-1]),
                                        n, dst.device.index, _stream_handle(stream)))


def encode_pixels(src, dst, dst_format, fill=0, stream=None):
    """lrp_encode_pixels_device: float32 CUDA tensor (..., C) -> `dst` (..., dst_channels) in `dst_format`."""
    lib = _native.load()
    n = src.numel() // src.shape[-1]
    _check(lib.lrp_encode_pixels_device(src.data_ptr(), int(src.shape[-1]), dst.data_ptr(), int(dst_format), int(dst.shape[-1]),
                                        int(fill), n, src.device.index, _stream_handle(stream)))


def pixel_tables():
    """(decode[256], threshold[256]) of LRP_PIXEL_U8_GAMMA as the host's powf made them."""
    dec = (ctypes.c_float * 256)()
    thr = (ctypes.c_float * 256)()
    _native.load().lrp_pixel_tables(dec, thr)
    return np.frombuffer(dec, dtype=np.float32).copy(), np.frombuffer(thr, dtype=np.float32).copy()
