/*
 * lrp.h — C ABI of the MI355X-native lens reprojection library (liblrp_hip.so).
 *
 * This is the drop-in boundary for the reference's L4 -> L1 call: the worker
 * lambda in reference src/main.cpp:597-603 calls
 *     reproject::reproject(&input, &output, num_samples, interpolation, rotation_matrix);
 *     reproject::post_process(&output, exposure, reinhard);
 * declared in reference src/reproject.hpp:22-27.  The reference has no FFI layer
 * of its own; a maintainer binds these entry points from the C++
 * wrapper in include/lens_reproject.hpp.
 *
 * Plain C: pointers, sizes, fixed-width enums.  No torch / HIP types appear in
 * the signatures (a HIP stream travels as void*).  Every function returns an
 * lrp_status; nothing exits the process or throws.  The library never falls
 * back to a CPU path: without a usable gfx950 device every compute entry point
 * fails with LRP_ERR_NO_DEVICE / LRP_ERR_HIP.
 *
 * Thread safety: all entry points may be called concurrently from different
 * host threads (the reference calls reproject() from -j N pool threads,
 * src/main.cpp:538-544).  An lrp_context must be used by one thread at a time.
 */
#ifndef LRP_H
#define LRP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRP_ABI_VERSION 3

/* ---- enums: numbering identical to the reference's ------------------------ */

/* reference src/config.hpp:7-13 (enum LensType) */
typedef enum lrp_lens_type {
  LRP_RECTILINEAR = 0,
  LRP_FISHEYE_EQUIDISTANT = 1,
  LRP_FISHEYE_EQUISOLID = 2,     /* declared by the reference, rejected by reproject() */
  LRP_FISHEYE_STEREOGRAPHIC = 3, /* declared by the reference, rejected by reproject() */
  LRP_EQUIRECTANGULAR = 4
} lrp_lens_type;

/* reference src/reproject.hpp:16-20 (enum Interpolation) */
typedef enum lrp_interpolation { LRP_NEAREST = 0, LRP_BILINEAR = 1, LRP_BICUBIC = 2 } lrp_interpolation;

/* reference src/reproject.hpp:7 (enum DataLayout); carried, never read by the kernels */
typedef enum lrp_data_layout { LRP_RGB = 0, LRP_RGBA = 1, LRP_RGBZ = 2, LRP_RGBAZ = 3 } lrp_data_layout;

typedef enum lrp_status {
  LRP_OK = 0,
  LRP_ERR_OUTPUT_LENS = 1,   /* reference prints "Output lens type not supported." and exit(1), src/reproject.cpp:415-417 */
  LRP_ERR_INPUT_LENS = 2,    /* "Input lens type not supported.",  src/reproject.cpp:395-397 */
  LRP_ERR_INTERPOLATION = 3, /* "Interpolation method not supported.", src/reproject.cpp:364-366 */
  LRP_ERR_CHANNELS = 4,      /* in->channels != out->channels (unchecked precondition in the reference) or < 1 */
  LRP_ERR_BAD_DIMS = 5,      /* non-positive size, or an image of more than 2^31 floats (what the reference's int indexing addresses) */
  LRP_ERR_NULL = 6,          /* NULL image / data pointer */
  LRP_ERR_NO_DEVICE = 7,     /* no HIP device, or device index out of range */
  LRP_ERR_HIP = 8,           /* a HIP runtime call failed; see lrp_last_error() */
  LRP_ERR_OOM = 9,           /* device or pinned-host allocation failed */
  LRP_ERR_BAD_ARG = 10
} lrp_status;

/* ---- PODs: layout identical to the reference's ----------------------------- */

/* reference src/config.hpp:15-37 (struct LensInfo): sizeof 28; type @0, union @4,
 * sensor_width @20, sensor_height @24.  Millimetres and radians. */
typedef struct lrp_lens {
  int32_t type; /* lrp_lens_type */
  union {
    struct { float focal_length; } rectilinear;
    struct { float fov; } fisheye_equidistant;
    struct { float focal_length; float fov; } fisheye_equisolid;
    struct { float latitude_min, latitude_max, longitude_min, longitude_max; } equirectangular;
    float raw[4];
  } u;
  float sensor_width;
  float sensor_height;
} lrp_lens;

/* reference src/reproject.hpp:9-14 (struct Image): sizeof 56; lens @0, width @28,
 * height @32, channels @36, data @40, data_layout @48.  Interleaved row-major
 * float32, data[(y*width + x)*channels + c], no row padding. */
typedef struct lrp_image {
  lrp_lens lens;
  int32_t width, height, channels;
  float *data;
  int32_t data_layout; /* lrp_data_layout */
} lrp_image;

/* Optional fused epilogue == reference post_process(img, exposure, reinhard)
 * (src/reproject.cpp:421-437) applied to the freshly written output.  The
 * reference CLI runs it when exposure != 1 || reinhard != 1 (src/main.cpp:601). */
typedef struct lrp_post {
  float exposure;
  float reinhard;
} lrp_post;

/* ---- library / device ------------------------------------------------------ */

int lrp_abi_version(void);
/* Number of usable HIP devices (0 when there is none; never negative). */
int lrp_device_count(void);
/* Static text for a status code. */
const char *lrp_strerror(int status);
/* Detail of the calling thread's last failure (HIP error string etc.), "" if none. */
const char *lrp_last_error(void);

/* Three HIP kernel families compute the same bits: 0 = one pixel per lane
 * (any channel count), 1 = tile kernel (RGB / RGBA / RGBAZ), 2 = tile kernel + LDS-window
 * bicubic (default; the library picks the pixel kernel by itself where the others
 * do not apply), 3 = as 2 with the window kernel's shared-coefficient tier switched off.  Testing / A-B knob: sets the family for subsequent calls of all
 * threads and returns the previous one; an out-of-range value only queries.  (The library reads no environment variable.) */
int lrp_debug_kernel(int choice);
/* The other testing / A-B switches, by name: "kernel" (as lrp_debug_kernel), "xsep", "quad", "mirror_modes",
 * "win_edge", "win_split", "geo_cache" (0 / 1: a sharing or staging path of the tile / window kernels off / on — the bits
 * do not change), "batch_frames" (frames per wavefront of a batched launch, 0 = automatic),
 * "multi_fork" (side streams of lrp_reproject_multi_device, 0-5), "geo_strip" (blocks per wavefront of a launch that reads the
 * geometry cache, 0 = automatic), "geo_big" (the big-window variant of those kernels: 1 = a rectilinear view rendered into a panorama and every geometry whose census says that
 * 30 % of its in-view blocks are too large for a 10 KiB window, 0 = never, 2 = always),
 * "geo_lists" (rendering by block class from the lists of a geometry-cache entry: 0 never, 1 where corner blocks are at least
 * 30 % of the frame, 2 whenever the lists are known), "geo_fill_fused" (0: the corner runs of such a launch always by the fill kernel, not as a share per
 * wavefront of the window kernel), "geo_fill_stream" (1: that fill kernel on a side stream), "big_launches" (a counter), "context_streams" (0: an lrp_context keeps all its kernels on one compute stream instead of alternating two),
 * "win_ss" (0: bicubic with num_samples 2-4 through the tile kernel instead of the window kernel's supersampling instantiations),
 * "geo_census" (0: no census of a new entry's windows), "geo_list_recs" (0: listed wavefronts read their block's record from the box array),
 * "win_tapdma" (0: passes of the big-window variant whose window fits no buffer gather per lane instead of fetching their taps quad by quad through LDS-DMA),
 * "listed_launches" (a counter: launches rendered by block class so far; 0 resets).  Sets the value for subsequent calls of all threads
 * and returns the previous one; a value outside the switch's range only queries; an unknown name returns -1. */
int lrp_debug_set(const char *name, int value);
/* Frees the cached per-output-lens tables and the geometry cache of every device (after
 * synchronising them).  Optional: both caches are bounded and reused across calls. */
void lrp_release_cached_tables(void);

/* ---- geometry cache ---------------------------------------------------------- */

/* The reference renders a whole run with ONE geometry — lenses, sizes and rotation are command-line
 * constants, only the pixels change from file to file (src/main.cpp:576-598) — and derives the source
 * coordinates of every output pixel again for every file (src/reproject.cpp:287-324).  Here the first
 * single-image bicubic launch of a geometry (lrp_reproject, lrp_reproject_device,
 * lrp_reproject_multi_device, lrp_context_submit*) leaves those coordinates in device memory as a side
 * output (8 bytes per output pixel + 2 per 16 pixels) and later launches of the same geometry on that device load them
 * instead of computing them: same values, same rendered bits, 1.2-1.9x the kernel rate.  Keyed on
 * (device, both lenses, both sizes, rotation, num_samples); least recently used entries are dropped when
 * `max_bytes` per device would be exceeded; a launch being captured into a hipGraph does not use it.
 * num_samples 2-4 (the reference's --samples) keep an entry of their own kind, shared by the three samplers: a coordinate
 * pair per SUB-SAMPLE, 8 x num_samples^2 bytes per output pixel.
 *   max_bytes      bytes per device (default: min(4 GiB, 2 % of the device's memory), at least two entries of the largest
 *                  geometry seen; -2 restores it); 0 switches the cache off and frees it; -1 keeps the value
 *   min_sightings  a geometry is cached from its n-th launch on (default 1; 2 suits callers whose
 *                  rotation changes with every call — the library switches to 2 by itself after
 *                  evicting several entries that were never read); < 1 keeps the value */
int lrp_geometry_cache_configure(long long max_bytes, int min_sightings);
typedef struct lrp_geometry_cache_info {
  uint64_t bytes, max_bytes, entries; /* device memory held now (all devices), the per-device limit, geometries held */
  uint64_t fills, hits, bypasses, evictions; /* launches that wrote an entry / read one / ran without the cache; entries dropped */
} lrp_geometry_cache_info;
void lrp_geometry_cache_stats(lrp_geometry_cache_info *out);

/* ---- one image, host buffers (the reference's calling convention) ---------- */

/* Drop-in for reproject::reproject (src/reproject.cpp:405-419) with in->data and
 * out->data in host memory (pageable or pinned).  Uploads the source, runs the
 * kernel on `device`, downloads the result; returns when out->data is complete.
 * rotation: 9 floats row-major or NULL (= no rotation, src/reproject.cpp:303).
 * post: NULL, or the fused post_process epilogue.
 * num_samples <= 0 leaves out->data untouched, as the reference loop does. */
int lrp_reproject(const lrp_image *in, lrp_image *out, int num_samples, int interpolation,
                  const float *rotation, const lrp_post *post, int device);

/* Drop-in for reproject::post_process (src/reproject.cpp:421-437), host buffer. */
int lrp_post_process(lrp_image *img, float exposure, float reinhard, int device);

/* ---- one image, device-resident buffers ------------------------------------ */

/* Same operation with in->data / out->data being device pointers on `device`.
 * Asynchronous: enqueues on `stream` (a hipStream_t, NULL = default stream) and
 * returns; the caller synchronises.  The first call with a new (output lens,
 * output size, num_samples) builds that lens's per-column / per-row tables
 * (one small allocation + a synchronous 2-microsecond kernel, cached afterwards);
 * every later call neither allocates nor synchronises and is capturable into a
 * hipGraph. */
int lrp_reproject_device(const lrp_image *in, lrp_image *out, int num_samples, int interpolation,
                         const float *rotation, const lrp_post *post, int device, void *stream);

int lrp_post_process_device(lrp_image *img, float exposure, float reinhard, int device, void *stream);

/* One resident source, n_out target lenses/rotations (cubemap faces etc.):
 * outs[i] is rendered with rotations + 9*i (or no rotation when rotations is
 * NULL).  Equivalent to n_out reference calls sharing `in`. */
int lrp_reproject_multi_device(const lrp_image *in, lrp_image *outs, int n_out, int num_samples,
                               int interpolation, const float *rotations, const lrp_post *post,
                               int device, void *stream);

/* Rows [row_first, row_first + row_count) of the output only (out->data is the whole image; the rows
 * of the reference loop are independent, src/reproject.cpp:284): what a job that splits one output over
 * several GPUs or streams launches.  The bytes are those of the same rows of a whole-image call. */
int lrp_reproject_rows_device(const lrp_image *in, lrp_image *out, int num_samples, int interpolation,
                              const float *rotation, const lrp_post *post, int row_first, int row_count,
                              int device, void *stream);

/* n images of ONE geometry (same sizes, channel count, lenses; one rotation, one
 * post setting — a directory of frames from one camera, src/main.cpp:540-622) on
 * device-resident buffers: rendered by one kernel launch per 16 images instead of
 * one per image, so the GPU does not drain and refill between frames.  Results are
 * those of n lrp_reproject_device calls. */
int lrp_reproject_batch_device(const lrp_image *ins, lrp_image *outs, int n, int num_samples,
                               int interpolation, const float *rotation, const lrp_post *post,
                               int device, void *stream);

/* ---- one source, several outputs, several GPUs (BASELINE configs[4]) ---------- */

/* One host source, n_out host outputs (lenses in outs[i].lens, rotations + 9 * i or none): the
 * cubemap job — six reference invocations over one 8192^2 panorama — on `n_devices` GPUs.
 * The source is uploaded ONCE, to devices[0], and copied to the other GPUs device to device
 * (hipMemcpyPeerAsync over xGMI where peer access is available, a second upload where not);
 * with n_out >= n_devices whole outputs are dealt round-robin over the GPUs (output i on
 * devices[i % n_devices]), with fewer outputs than GPUs every GPU renders band d of n_devices of
 * EVERY output (rows are independent); results are downloaded straight into outs[i].data.  No collective, no exchange of results.  The bytes are those of
 * n_out lrp_reproject calls on one GPU.  devices may name a GPU more than once. */
int lrp_reproject_multi(const lrp_image *in, lrp_image *outs, int n_out, int num_samples, int interpolation,
                        const float *rotations, const lrp_post *post, const int *devices, int n_devices);

/* ---- batches of independent images (the reference's --input-dir path) ------ */

/* A context owns a three-stage pipeline on one device — an upload stream, a
 * compute stream and a download stream, chained by events — and `n_streams`
 * image slots (device source + destination buffers, sized for the largest image
 * submitted so far).  Images are independent (src/main.cpp:540-622: one file per
 * pool thread); with n_streams >= 2 the H2D copy of image i+1, the kernel of
 * image i and the D2H copy of image i-1 run at the same time on both PCIe
 * directions (pinned host buffers make the copies asynchronous). */
typedef struct lrp_context lrp_context;

int lrp_context_create(lrp_context **ctx, int device, int n_streams);
void lrp_context_destroy(lrp_context *ctx);
/* Enqueue one image (host buffers).  Returns once the work is queued; in->data
 * must stay valid and out->data must not be read until lrp_context_wait. */
int lrp_context_submit(lrp_context *ctx, const lrp_image *in, lrp_image *out, int num_samples,
                       int interpolation, const float *rotation, const lrp_post *post);
/* Wait for everything submitted so far; returns the first error seen. */
int lrp_context_wait(lrp_context *ctx);

/* ---- pixel formats of the file path -------------- */

/* What the reference's codecs do on the host between a file and the float buffers of the
 * hot path, done on the device so that a frame crosses PCIe in its file format (8 or 4 bytes
 * per RGBA pixel instead of 16):
 *   LRP_PIXEL_F16       interleaved IEEE binary16 — an OpenEXR HALF channel; widened exactly,
 *                       narrowed round-to-nearest-even like Imath's half
 *                       (src/image_formats.cpp:266-295, 318-333);
 *   LRP_PIXEL_U8_GAMMA  8-bit samples; decode v = pow(p / 255, 2.2) (read_png :196-198, read_jpeg
 *                       :64-66), encode uint8(255.9 * pow(clamp(v, 0, 1), 1 / 2.2)) (save_png
 *                       :155-158) — bit for bit what the host's powf gives (tables made by it).
 * Channels: the first min(src_channels, dst_channels) are converted; extra destination
 * channels are filled (decode: 0.0f; encode: `fill`, e.g. 255 for the alpha byte save_png
 * writes when the image has no fourth channel). */
typedef enum lrp_pixel_format { LRP_PIXEL_F32 = 0, LRP_PIXEL_F16 = 1, LRP_PIXEL_U8_GAMMA = 2 } lrp_pixel_format;

/* Device buffers, asynchronous on `stream`. */
int lrp_decode_pixels_device(const void *src, int src_format, int src_channels, float *dst, int dst_channels,
                             size_t n_pixels, int device, void *stream);
int lrp_encode_pixels_device(const float *src, int src_channels, void *dst, int dst_format, int dst_channels,
                             unsigned fill, size_t n_pixels, int device, void *stream);
/* The two 256-entry tables behind LRP_PIXEL_U8_GAMMA as the host's powf made them: decode[k] =
 * pow(k / 255, 2.2), threshold[k] = smallest s in [0, 1] whose code is >= k.  No device needed. */
void lrp_pixel_tables(float decode[256], float threshold[256]);

/* Page-locked host memory for the buffers handed to a context (pageable buffers work too, but
 * make every copy synchronous and half as fast).  lrp_host_free(NULL) is a no-op. */
int lrp_host_alloc(void **ptr, size_t bytes);
void lrp_host_free(void *ptr);

/* lrp_context_submit with the host buffers in a file format: in->data holds in->width x
 * in->height pixels of `in_packed_channels` samples in `in_format` (e.g. the RGBA8 libpng
 * decodes to, of which in->channels = 3 are used; or in->channels HALF samples), out->data
 * receives `out_packed_channels` samples per pixel in `out_format` (`out_fill` for channels
 * beyond out->channels).  Upload, decode kernel, reproject (+ fused post_process), encode
 * kernel, download — pipelined over the context's slots like lrp_context_submit.
 * *ticket (may be NULL) identifies the submission for lrp_context_wait_ticket, which blocks
 * until that image's output has landed while later submissions keep flowing.  A context may
 * be shared by several host threads (submissions are serialised internally). */
int lrp_context_submit_packed(lrp_context *ctx, const lrp_image *in, int in_format, int in_packed_channels,
                              lrp_image *out, int out_format, int out_packed_channels, unsigned out_fill,
                              int num_samples, int interpolation, const float *rotation, const lrp_post *post,
                              int *ticket);
int lrp_context_wait_ticket(lrp_context *ctx, int ticket);

/* ---- synthetic frames (bench / tests) -------------------------------------- */

/* Fill a device buffer with the counter-based synthetic frame of SURVEY.md §8d
 * (identical bits to oracle lrpo_synth_fill on the host).  depth_channel = -1
 * for colour-only frames. */
int lrp_synth_fill_device(float *data, int width, int height, int channels, uint32_t seed,
                          int depth_channel, int device, void *stream);

/* Order-independent 64-bit checksum of the bit patterns of n floats on the device, written to
 * the device word *out (asynchronous on `stream`): sum mod 2^64 of h(bits[i], i) with
 * h = mix32(bits ^ 0xA5A5A5A5, 2 i + 0x7F4A7C15) << 32 | mix32(bits, i), mix32 as in the
 * synthetic generator.  bench.py and the multi-GPU tests compare per-image values between
 * 1-rank and N-rank runs without moving the images. */
int lrp_checksum_device(const float *data, size_t n, uint64_t *out, int device, void *stream);

/* Evaluate the device math routines on device arrays (lets tests prove the
 * device build of the math matches the host libm): func 0 sinf, 1 cosf,
 * 2 sincosf.sin, 3 sincosf.cos, 4 atanf, 5 asinf, 6 atan2f(a, b), 7 a / b,
 * 8 sqrtf(a), 9 (float)int(a) with the x86 cvttss2si convention.
 * `b` may be NULL for unary functions. */
int lrp_math_eval_device(int func, const float *a, const float *b, float *out, size_t n, int device,
                         void *stream);

/* ---- caller-side producers of hot-path inputs (host, no device needed) ------ */

/* computeRotationMatrix (reference src/main.cpp:98-142): R = R_y(pan) * R_x(pitch)
 * * R_z(roll), row-major, angles in radians, float sin/cos. */
void lrp_rotation_matrix(float pan, float pitch, float roll, float *out9);

/* Lens constructors with the reference CLI's conventions (src/main.cpp:15-95):
 * rectilinear: sensor_height = res_y / res_x * sensor_width (:27);
 * equidistant: sensor 36 x 36 mm (:53-54);
 * equirectangular: sensor 0 (:93); lrp_lens_equirectangular_full = "full"
 * (-pi..pi, -pi/2..pi/2 narrowed to float, :62-66). */
void lrp_lens_rectilinear(lrp_lens *lens, float focal_length, float sensor_width, float res_x, float res_y);
void lrp_lens_equidistant(lrp_lens *lens, float fov);
void lrp_lens_equirectangular(lrp_lens *lens, float longitude_min, float longitude_max, float latitude_min,
                              float latitude_max);
void lrp_lens_equirectangular_full(lrp_lens *lens);

#ifdef __cplusplus
}
#endif
#endif /* LRP_H */
