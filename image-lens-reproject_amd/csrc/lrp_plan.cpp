// lrp_plan.cpp — the launch planner (see lrp_plan.h).  Pure host code: no HIP, no state.
#include "lrp_plan.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace lrp {

PlanFamily plan_family(const PlanRequest &r, const PlanSwitches &s) {
  PlanFamily f;
  // The tile / window kernels are instantiated for RGB, RGBA and RGBAZ (what the reference's codecs deliver,
  // src/image_formats.cpp), address through 32-bit byte offsets and pack source coordinates into 16 + 15 bits in places.
  const bool tile_channels = r.channels >= 3 && r.channels <= 5;
  f.tile = s.kernel != 0 && tile_channels && r.byte_offsets_fit && r.in_w <= 65535 && r.in_h <= 32767 &&
           (long long)r.out_w * r.num_samples < (1ll << 30) && (long long)r.out_h * r.num_samples < (1ll << 30);
  // separable output-lens terms: the rectilinear and the equirectangular target (the equidistant one is not separable)
  f.wants_tables = f.tile && r.out_type != kPlanEquidistant;
  return f;
}

PlanRotation plan_rotation(const PlanRequest &r, const PlanSwitches &s, const PlanFamily &f, const TableFacts &t) {
  PlanRotation p;
  p.has_rot = r.has_rot;
  if (!f.tile || !f.wants_tables || !t.built || !t.plain) return p;
  // Every ray component is finite and no -0.0f: multiplying by the exact identity matrix (what the CLI passes for
  // --rotation 0,0,0, src/main.cpp:312-325) changes no bit of (vx, vy, vz) — drop it.
  static const float kIdentity[9] = {1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f};
  if (p.has_rot && std::memcmp(r.rot, kIdentity, sizeof(kIdentity)) == 0) p.has_rot = false;
  // Column-separable source x (lrp_tables.hip): ray x and z independent of the row.
  const bool rows_free = !p.has_rot || (r.rot[1] == 0.0f && r.rot[7] == 0.0f);
  p.wants_xsep = s.xsep != 0 && rows_free && r.in_mode != kPlanInEquidistant;
  return p;
}

PlanSharing plan_sharing(const PlanRequest &r, const PlanSwitches &s, const PlanFamily &f, const TableFacts &t, const PlanRotation &rot,
                         bool xsep_available) {
  PlanSharing p;
  if (!f.tile) return p;
  const int symmetry = (f.wants_tables && t.built) ? t.symmetry : 0;
  const bool out_eqd = r.out_type == kPlanEquidistant;
  const int ns = r.num_samples, interp = r.interpolation, k = s.kernel;
  // Mirrored pixels / blocks: without a rotation the mapping is symmetric about both image axes and the lens-plane
  // coordinates of the four mirror pixels differ in sign only.  Rectilinear / equirectangular target: the ray tables must be
  // mirror images bit for bit; equidistant target: the ray is odd in cx, cy by construction.  An equirectangular source
  // takes part through the column-separable x table only (its longitude is not odd in x).  Family 3 keeps every sharing
  // path of the tile / window kernels off (cross-checks).
  const bool in_eqr = r.in_mode == kPlanInEquirect || r.in_mode == kPlanInEquirectLoop;
  const bool sym_out = out_eqd ? true : symmetry == 3;
  const bool sharing = s.quad != 0 && k != 3;
  p.quad = (!r.band && sharing && ns == 1 && !rot.has_rot && sym_out && (!in_eqr || xsep_available)) ? 1 : 0;
  // A batch of nearest-neighbour frames shares its coordinates between up to 16 frames (the plain path of the tile
  // kernel keeps them in registers), which beats sharing them between four mirror pixels: 71 -> 66 us per 4K frame.
  // ... and likewise for bilinear (same-box A/B, 16-frame launches: equirect -> rect 91.7 -> 83.7 us, fisheye -> rect 90.9 ->
  // 84.2, equirect -> fisheye rotated 127.4 -> 118.6 with plain pixels + frames instead of mirrored pixels / rays).
  const bool batch_plain = r.n_batch >= 4 && interp != kPlanBicubic && ns == 1;
  if (p.quad != 0 && batch_plain) p.quad = 0;
  // (num_samples 2-4 — the reference's --samples, src/main.cpp:192-196 — have their own instantiations: plain blocks, no
  // mirror mode, no cache)
  p.window = interp == kPlanBicubic && k >= 2 && (ns == 1 || (ns >= 2 && ns <= kMaxWindowSamples && s.win_ss != 0)) &&
             (r.channels == 4 || r.channels == 3 || r.channels == 5);
  p.window1 = p.window && ns == 1;
  // Equidistant target, rotated (or an equirectangular source): the four mirror pixels still share the ray through the
  // output lens (tile kernels only).
  if (!r.band && p.quad == 0 && !p.window && !batch_plain && sharing && ns == 1 && out_eqd) p.quad = 2;
  p.win_coef = k == 2 ? 1 : 0;
  p.win_edge = (k == 2 && s.win_edge != 0) ? 1 : 0;
  p.win_split = (k == 2 && s.win_split != 0) ? 1 : 0;
  p.win_tapdma = s.win_tapdma != 0 ? 1 : 0;
  // Mirror mode of the window kernel (lrp_win_kernel.h QMode): both axes without a rotation; rows only for a pan,
  // columns only for a pitch into a rectilinear target.  Signed zeros count as zeros in the matrix tests.
  p.win_mode = p.quad == 1 ? 1 : 0;
  const bool mirror_modes = p.window1 && !r.band && sharing && s.mirror_modes != 0;
  // equidistant target that is not fully mirrored (a rotation, or an equirectangular source): the four mirror pixels
  // still share the ray through the output lens
  // (not for a batch: its wavefronts share ALL of the coordinate math between up to 16 frames on plain blocks at
  // four wavefronts per SIMD — equirect -> fisheye rotated 143 us against 147 with shared rays at three)
  if (mirror_modes && p.win_mode == 0 && out_eqd && r.n_batch < 4) p.win_mode = 4;
  if (mirror_modes && p.win_mode == 0 && rot.has_rot) {
    const float *R = r.rot;
    auto tiny = [](float v) { return !(std::fabs(v) >= 0x1p-20f); }; // (also true for a NaN)
    if (xsep_available && (symmetry & 2) && R[3] == 0.0f && R[5] == 0.0f && !tiny(R[4]))
      p.win_mode = 2; // ny = R4 vy exactly: odd in vy; nx, nz come from the column table
    else if (r.out_type == kPlanRect && (symmetry & 1) && R[1] == 0.0f && R[2] == 0.0f && R[3] == 0.0f && R[6] == 0.0f && !tiny(R[0]) &&
             R[5] != 0.0f && R[8] != 0.0f && std::isfinite(R[4]) && std::isfinite(R[5]) && std::isfinite(R[7]) && std::isfinite(R[8]))
      p.win_mode = 3; // nx = R0 vx exactly: odd in vx; ny, nz end in the non-zero terms R5 vz, R8 vz (vz = -1: no underflow)
  }
  // The view's copy behind the camera sits half a turn away, upside down (lrp_win_kernel.h "alias pairs"): only when
  // the panorama spans the full turn and the rotation neither pitches nor rolls.
  if (r.out_type == kPlanEquirect && r.in_type == kPlanRect) {
    const bool yaw_only = !rot.has_rot || (r.rot[1] == 0.0f && r.rot[3] == 0.0f && r.rot[5] == 0.0f && r.rot[7] == 0.0f);
    p.alias_pairs = (std::fabs(r.out_lon_span - 6.2831855f) < 1e-3f && yaw_only) ? 1 : 0;
  }
  p.frames_per_wave = s.batch_frames; // 0: the launcher decides
  // Geometry cache (lrp_geocache.h): a single whole-image launch of the window kernel loads the coordinates of its
  // pixels and the window extremes of its blocks when an earlier launch of the same geometry has left them in HBM,
  // and leaves them there when it is the first.  Both run plain blocks: the entry is a plain per-pixel map.
  // Batched bicubic launches read the entry as well (their wavefronts load a block's coordinates once and walk its frames);
  // the first frame of a batch whose geometry has no entry yet is rendered by a launch of its own, which writes it.
  // Nearest / bilinear single launches (tile kernel, one sample per pixel) use the coordinate map of the same entries.
  // (nearest without a rotation: the mirrored pixels of the compute kernel are as fast as a load per pixel — 75.8 against 78.2 us
  // per 4K frame — and need no entry)
  // (... and a rectilinear source under a rectilinear / equirectangular target: four divides a pixel cost less than the 8 bytes
  // a pixel the map adds to these memory-bound kernels — rect -> equirect nearest 105 -> 117 us, bilinear 144 -> 159 with it)
  const bool cheap_coordinates = r.in_mode == kPlanInRect && !out_eqd;
  // Batched bilinear launches read the map too, a frame per workgroup (8 wavefronts per SIMD against the 4 of the instantiations
  // that hold coordinates across frames: equirect -> fisheye rotated 129.4 -> 107.1 us per frame, equirect -> rect 89.3 -> 84.3);
  // batched nearest keeps the frame loop (69.4 against 72.9 us).
  const bool tile_single = !p.window && interp != kPlanBicubic && ns == 1 && (r.n_batch <= 0 || interp == kPlanBilinear) &&
                           !(interp == kPlanNearest && p.quad != 0) && !cheap_coordinates;
  // The supersampling instantiations (num_samples 2-4) keep an entry of their own kind — a coordinate pair per SUB-SAMPLE, no
  // records — written by the first launch of the geometry and read by every later one (no lens math in those); element indices
  // are 32-bit.
  // (not for a rectilinear view rendered into a panorama: most of that frame is out of view — corner blocks, edge rows — and its
  // coordinates are four divides: rect -> equirect ns 2 208 us computing, 248 reading 8 bytes per sub-sample; rect -> rect gains
  // like everybody: 176 -> 160)
  const bool window_ss = p.window && !p.window1 && !(r.in_mode == kPlanInRect && r.out_type == kPlanEquirect) &&
                         (long long)r.out_w * r.out_h * ns * ns < (1ll << 31);
  // ... and so do nearest / bilinear with num_samples 2-4: the first launch (tile kernel, computing) writes the entry, later ones
  // read it with a lane per SUB-SAMPLE (lrp_ss_gather_kernel.h: coalesced loads, the ordered DPP sum) — one entry serves the three
  // samplers.  (Through the tile kernel, a lane per PIXEL, the loads of num_samples 3, 4 are 72 / 128 bytes apart between lanes and
  // lose to computing: profiles/r06_experiments_ab.txt item 10.)  Not where the coordinates are cheap: a rectilinear source
  // (above), or a source x that comes from the column table (equirect -> rect bilinear without a rotation: 81 / 87 us computing,
  // 91 / 91 loading at ns 3 / 4).  What it is for: equirect -> fisheye bilinear rotated ns 2 / 3 / 4 172 / 218 / 330 -> 94 / 102 /
  // 101 us, equirect -> rect bilinear rotated 110-120 -> 78-87.
  const bool tile_ss = !p.window && interp != kPlanBicubic && ns >= 2 && ns <= kMaxWindowSamples && !cheap_coordinates && !xsep_available &&
                       (long long)r.out_w * r.out_h * ns * ns < (1ll << 31);
  p.wants_geo = (p.window1 || tile_single || window_ss || tile_ss) && !r.band && k == 2 && s.geo_cache != 0;
  p.geo_want_boxes = p.window1;
  return p;
}

PlanGeo plan_geo(const PlanRequest &r, const PlanSwitches &s, const PlanSharing &sh, const GeoFacts &g) {
  PlanGeo p;
  p.win_mode = sh.win_mode;
  p.quad = sh.quad;
  if (g.mode == 0) return p;
  p.geo_mode = g.mode;
  p.win_mode = 0;
  p.quad = 0; // (tile kernels: the plain path writes / the GeoRead kernels read the map)
  if (r.num_samples > 1) { // an entry of sub-samples: written (1) or read (2) by the SS instantiations / the tile kernel, nothing else applies
    if (g.mode != 1 && g.mode != 2) p.geo_mode = 0;
    return p;
  }
  p.blocks_per_wave = s.geo_strip; // 0: the launcher decides
  p.rgbaz_runs = (r.out_type == kPlanEquirect && r.in_type == kPlanRect) ? 1 : 0;
  // The big-window variant (lrp_win_kernel.h kBigWin: 17.5-20 KiB of LDS per wavefront, two wavefronts per SIMD, tap DMA): a
  // rectilinear view rendered into a panorama; and any geometry out of a rectilinear or panorama source whose census
  // (lrp_geo_lists.hip) says that at least kBigWidePercent % of its in-view blocks have windows the 10 KiB buffer of the
  // four-wavefront kernels cannot stage (a cubemap's pole faces: 97 -> 80 us, a rectilinear view into a fisheye frame
  // 160 -> 145; a cubemap's side faces and the ~1:1 mappings have no such block and lose 25-30 % there).
  const bool wide = g.mode == 2 && g.lists && r.in_mode != kPlanInEquidistant && g.n_inview != 0 &&
                    (unsigned long long)g.n_wide * 100u >= (unsigned long long)g.n_inview * kBigWidePercent;
  p.big_windows = s.geo_big == 2 ? 1 : s.geo_big != 0 ? ((p.rgbaz_runs != 0 || wide) ? 1 : 0) : 0; // (2: wherever the variant is instantiated)
  // Rendering by block class (lrp_params.h "Block lists"): once the lists of the entry are known, the corner blocks
  // — every pixel the one clamped corner texel — are written by the store-only fill kernel and the window kernel
  // walks the work list, which holds no corner block.  Where corner blocks are rare the plain enumeration stays.
  if (sh.window && r.in_mode == kPlanInRect && g.mode == 2 && g.lists && s.geo_lists != 0 && g.n_blocks != 0 &&
      (s.geo_lists == 2 || (unsigned long long)g.n_corner_blocks * 100u >= (unsigned long long)g.n_blocks * kListedCornerPercent)) {
    p.listed = true;
    p.list_recs = s.geo_list_recs != 0;
    // the corner runs: a share per wavefront of the window launch where it has enough wavefronts to spread them over,
    // else (few or no blocks to render: the frame is nearly all corners) the fill kernel at its own, full occupancy
    if (s.geo_fill_fused != 0 && g.n_work >= kMinWavesForFusedFill && g.n_runs != 0) {
      // every stride-th wavefront (odd stride: all XCDs) writes at least one whole run (16 row segments)
      const unsigned long long segs = (unsigned long long)g.n_runs * 16u;
      unsigned stride = (unsigned)std::max<unsigned long long>(1, 16ull * g.n_work / segs) | 1u;
      stride = std::min(stride, std::max(1u, g.n_work / 1024u) | 1u); // (at least ~1024 filling wavefronts)
      const unsigned fillers = (g.n_work + stride - 1) / stride;
      p.fill_stride = stride;
      p.fill_per_wave = (unsigned)((segs + fillers - 1) / fillers);
    }
  }
  return p;
}

} // namespace lrp
