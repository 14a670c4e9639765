// lrp_kernel_v2.h — the tile and window kernels: the hot path as it runs on gfx950.
//
// Work decomposition (tile kernel: nearest, bilinear, RGBAZ / super-sampled bicubic)
//   * A 256-thread workgroup (4 wavefronts) owns a 64 x 4R tile of output pixels
//     (R = 4 rows per wavefront for nearest, 2 for bilinear / bicubic).  Lane l of
//     every wavefront owns output column tile_x*64 + l; wavefront w owns the R
//     contiguous rows Rw .. Rw+R-1, so each of its stores is one contiguous 1 KiB
//     run (64 lanes x float4) and every per-row quantity is wave-uniform.
//   * Tiles are numbered so that the workgroups the dispatcher deals to one XCD
//     (blockIdx % 8 equal) walk bands of neighbouring tile rows (xcd_tile()): neighbouring
//     tiles read neighbouring source rows, which then hit in that XCD's 4 MiB L2, and
//     every XCD gets its share of every part of the frame.
//
// Instruction diet (the reference loop is VALU-bound on this chip, not HBM-bound: every
// packed, SGPR-operand, compare / select / convert / divide-helper instruction costs 4 issue
// cycles per wavefront, an IEEE divide 11 instructions, a sqrt ~12 — DESIGN.md section 5).
// Nothing below changes an operation, an operand or an order of the reference; work is only
// evaluated fewer times:
//   * separable output-lens terms come from per-column / per-row tables (lrp_tables.hip)
//     instead of 4 divides or 3 double-precision polynomials per pixel;
//   * lens-only constants (sensor_width / fov, angular spans) come from the host;
//   * x / -z, y / -z is skipped when -z == 1 for the whole wavefront (division by one is the
//     identity in IEEE arithmetic);
//   * the horizontal source coordinate of a rectilinear / equirectangular source comes from a
//     per-column table whenever the rotated ray's x and z do not depend on the row;
//   * without a rotation the four mirror images of a pixel share stage 1 of the coordinate
//     math (P.quad); for the equidistant target they share the ray under any rotation.
//
// Bicubic taps through LDS (window kernel, RGBA and RGB, one wavefront per workgroup)
//   The 16 taps of neighbouring pixels overlap almost completely, and 16 float4 gathers per
//   pixel saturate the texture-address path long before HBM.  A wavefront reduces the
//   tap-index bounding box of its 16 x 16 block with DPP-fused min / max, fetches that window
//   with LDS-DMA into its private 10 KiB of LDS (no barrier anywhere), evaluates the
//   weight-independent part of the vertical Catmull-Rom cubics once per window column and
//   reads 12 coefficient vectors + 4 taps per pixel at compile-time offsets from one address.
//   Windows larger than the LDS budget (poles, seam, strong minification) and non-consecutive
//   taps fall back, per wavefront, to explicit per-tap addressing.
//
// Files: lrp_kernel_common.h (shared device code), lrp_tile_kernel.h (tile kernel + launcher), lrp_win_kernel.h (window
// kernel + launcher).  Translation units include this umbrella.
#pragma once

#include "lrp_tile_kernel.h"
#include "lrp_win_kernel.h"
