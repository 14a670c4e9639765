// lrp_tile_win.hip — bicubic window-kernel instantiations (lrp_kernel_v2.h): RGBA, plain blocks; the dispatcher.
#include "lrp_kernel_v2.h"

namespace lrp {
// one translation unit per (channel count, mirror mode): they compile in parallel
hipError_t launch_win_bicubic_c4_m1(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winq.hip
hipError_t launch_win_bicubic_c4_m2(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winy.hip
hipError_t launch_win_bicubic_c4_m3(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winx.hip
hipError_t launch_win_bicubic_c4_m4(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winr.hip
hipError_t launch_win_bicubic_c3_m4(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winr3.hip
hipError_t launch_win_bicubic_c5_m4(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winr5.hip
hipError_t launch_win_bicubic_c3_m0(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_win3.hip
hipError_t launch_win_bicubic_c3_m1(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winq3.hip
hipError_t launch_win_bicubic_c3_m2(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winy3.hip
hipError_t launch_win_bicubic_c3_m3(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winx3.hip
hipError_t launch_win_bicubic_c5_m0(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_win5.hip
hipError_t launch_win_bicubic_c5_m1(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winq5.hip
hipError_t launch_win_bicubic_c5_m2(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winy5.hip
hipError_t launch_win_bicubic_c5_m3(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winx5.hip
hipError_t launch_win_bicubic_geo_c3(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_wing3.hip
hipError_t launch_win_bicubic_geo_c4(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_wing.hip
hipError_t launch_win_bicubic_geo_c5(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_wing5.hip
hipError_t launch_win_bicubic_ss_c3(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_wins3.hip
hipError_t launch_win_bicubic_ss_c4(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_wins.hip
hipError_t launch_win_bicubic_ss_c5(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_wins5.hip
hipError_t launch_win_bicubic_ssg_c3(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winsg3.hip
hipError_t launch_win_bicubic_ssg_c4(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winsg.hip
hipError_t launch_win_bicubic_ssg_c5(const KParams &P, int out_idx, int in_mode, hipStream_t stream); // lrp_tile_winsg5.hip
// P.channels must be 3, 4 or 5, P.num_samples 1 to 4; P.win_mode = the mirror mode (lrp_kernel_v2.h QMode).
// P.geo_mode == 2: the instantiations that load their coordinates from the geometry cache (plain blocks).
// P.num_samples 2, 3, 4: the supersampling instantiations (plain blocks; P.geo_mode 1 / 2: they write / read an entry of sub-samples).
hipError_t launch_win_bicubic(const KParams &P, int out_idx, int in_mode, hipStream_t stream) {
  using Fn = hipError_t (*)(const KParams &, int, int, hipStream_t);
  if (P.num_samples >= 2) {
    static const Fn ss_table[3] = {launch_win_bicubic_ss_c3, launch_win_bicubic_ss_c4, launch_win_bicubic_ss_c5};
    static const Fn ssg_table[3] = {launch_win_bicubic_ssg_c3, launch_win_bicubic_ssg_c4, launch_win_bicubic_ssg_c5};
    if (P.win_mode != 0 || P.geo_mode == 3) return hipErrorInvalidValue;
    return (P.geo_mode == 2 ? ssg_table : ss_table)[P.channels - 3](P, out_idx, in_mode, stream); // (2: the entry of sub-samples is read)
  }
  if (P.geo_mode == 2) {
    static const Fn geo_table[3] = {launch_win_bicubic_geo_c3, launch_win_bicubic_geo_c4, launch_win_bicubic_geo_c5};
    if (P.win_mode != 0) return hipErrorInvalidValue;
    return geo_table[P.channels - 3](P, out_idx, in_mode, stream);
  }
  static const Fn table[3][5] = {
      {launch_win_bicubic_c3_m0, launch_win_bicubic_c3_m1, launch_win_bicubic_c3_m2, launch_win_bicubic_c3_m3, launch_win_bicubic_c3_m4},
      {[](const KParams &Q, int o, int i, hipStream_t s) { return launch_win_bicubic_impl<0, 4>(Q, o, i, s); }, launch_win_bicubic_c4_m1,
       launch_win_bicubic_c4_m2, launch_win_bicubic_c4_m3, launch_win_bicubic_c4_m4},
      {launch_win_bicubic_c5_m0, launch_win_bicubic_c5_m1, launch_win_bicubic_c5_m2, launch_win_bicubic_c5_m3, launch_win_bicubic_c5_m4}};
  if (P.win_mode < 0 || P.win_mode > 4) return hipErrorInvalidValue;
  return table[P.channels - 3][P.win_mode](P, out_idx, in_mode, stream);
}
} // namespace lrp

#if defined(LRP_TIER_STATS)
extern "C" void lrp_debug_read_tiers_plain(unsigned out[8]) { // plain RGBA instantiations only
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(lrp::g_tier_stats), sizeof(unsigned) * 8);
  unsigned zero[8] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(lrp::g_tier_stats), zero, sizeof(zero));
}
#endif
