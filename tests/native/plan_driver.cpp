// plan_driver.cpp — CPU driver of the launch planner (csrc/lrp_plan.cpp: pure functions, no HIP): reads one request per line
// on stdin, prints the plan as one JSON object per line.  tests/test_plan.py feeds it the BASELINE configs, a cubemap's pole
// and side face, batches and the switches, and compares against the table in the test.
//
// line: key=value pairs separated by blanks — the fields of PlanRequest, PlanSwitches (prefix s.), TableFacts (prefix t.),
// xsep=0|1 (does the column-separable table exist when asked for; default: as wanted) and GeoFacts (prefix g.);
// rot=r0,r1,...,r8.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <sstream>
#include <string>

#include "lrp_plan.h"

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::map<std::string, std::string> kv;
    std::istringstream in(line);
    std::string tok;
    while (in >> tok) {
      const size_t eq = tok.find('=');
      if (eq == std::string::npos) {
        std::fprintf(stderr, "bad token %s\n", tok.c_str());
        return 2;
      }
      kv[tok.substr(0, eq)] = tok.substr(eq + 1);
    }
    auto geti = [&](const char *k, int d) { return kv.count(k) ? std::atoi(kv[k].c_str()) : d; };
    auto getf = [&](const char *k, float d) { return kv.count(k) ? (float)std::atof(kv[k].c_str()) : d; };
    lrp::PlanRequest r;
    r.out_type = geti("out_type", r.out_type), r.in_type = geti("in_type", r.in_type), r.in_mode = geti("in_mode", r.in_mode);
    r.out_w = geti("out_w", 4096), r.out_h = geti("out_h", 4096), r.in_w = geti("in_w", 4096), r.in_h = geti("in_h", 4096);
    r.channels = geti("channels", 4), r.num_samples = geti("ns", 1), r.interpolation = geti("interp", 2);
    r.out_lon_span = getf("out_lon_span", 0.0f);
    r.n_batch = geti("n_batch", 0), r.band = geti("band", 0) != 0, r.byte_offsets_fit = geti("fits", 1) != 0;
    if (kv.count("rot")) {
      r.has_rot = true;
      std::istringstream rs(kv["rot"]);
      std::string v;
      for (int i = 0; i < 9 && std::getline(rs, v, ','); ++i) r.rot[i] = (float)std::atof(v.c_str());
    }
    lrp::PlanSwitches s;
    s.kernel = geti("s.kernel", s.kernel), s.xsep = geti("s.xsep", s.xsep), s.quad = geti("s.quad", s.quad);
    s.mirror_modes = geti("s.mirror_modes", s.mirror_modes), s.win_edge = geti("s.win_edge", s.win_edge), s.win_split = geti("s.win_split", s.win_split);
    s.win_tapdma = geti("s.win_tapdma", s.win_tapdma), s.win_ss = geti("s.win_ss", s.win_ss), s.batch_frames = geti("s.batch_frames", s.batch_frames);
    s.geo_cache = geti("s.geo_cache", s.geo_cache), s.geo_strip = geti("s.geo_strip", s.geo_strip), s.geo_big = geti("s.geo_big", s.geo_big);
    s.geo_lists = geti("s.geo_lists", s.geo_lists), s.geo_fill_fused = geti("s.geo_fill_fused", s.geo_fill_fused), s.geo_list_recs = geti("s.geo_list_recs", s.geo_list_recs);
    lrp::TableFacts t;
    t.built = geti("t.built", 1) != 0, t.plain = geti("t.plain", 1) != 0, t.symmetry = geti("t.symmetry", 3);
    lrp::GeoFacts g;
    g.mode = geti("g.mode", 0), g.lists = geti("g.lists", 0) != 0;
    g.n_work = (uint32_t)geti("g.n_work", 0), g.n_runs = (uint32_t)geti("g.n_runs", 0), g.n_corner_blocks = (uint32_t)geti("g.n_corner_blocks", 0);
    g.n_blocks = (uint32_t)geti("g.n_blocks", 0), g.n_wide = (uint32_t)geti("g.n_wide", 0), g.n_inview = (uint32_t)geti("g.n_inview", 0);

    // the stages in the launcher's order (lrp_capi.cpp enqueue_reproject)
    lrp::PlanFamily f = lrp::plan_family(r, s);
    lrp::PlanRotation pr;
    pr.has_rot = r.has_rot;
    if (f.wants_tables) {
      if (!t.built)
        f.tile = false; // (no memory for the tables: the pixel kernel)
      else
        pr = lrp::plan_rotation(r, s, f, t);
    }
    const bool xsep = pr.wants_xsep && geti("xsep", 1) != 0;
    const lrp::PlanSharing sh = lrp::plan_sharing(r, s, f, t, pr, xsep);
    lrp::PlanGeo pg = lrp::plan_geo(r, s, sh, sh.wants_geo ? g : lrp::GeoFacts{});
    const char *family = !f.tile ? "pixel" : sh.window ? "window" : "tile";
    std::printf("{\"family\":\"%s\",\"wants_tables\":%d,\"has_rot\":%d,\"wants_xsep\":%d,\"quad\":%d,\"win_mode\":%d,\"win_coef\":%d,\"win_edge\":%d,"
                "\"win_split\":%d,\"win_tapdma\":%d,\"alias_pairs\":%d,\"frames_per_wave\":%d,\"wants_geo\":%d,\"geo_want_boxes\":%d,\"geo_mode\":%d,"
                "\"blocks_per_wave\":%d,\"rgbaz_runs\":%d,\"big_windows\":%d,\"listed\":%d,\"list_recs\":%d,\"fill_stride\":%u,\"fill_per_wave\":%u}\n",
                family, f.wants_tables ? 1 : 0, pr.has_rot ? 1 : 0, pr.wants_xsep ? 1 : 0, pg.quad, pg.win_mode, sh.win_coef, sh.win_edge, sh.win_split,
                sh.win_tapdma, sh.alias_pairs, sh.frames_per_wave, sh.wants_geo ? 1 : 0, sh.geo_want_boxes ? 1 : 0, pg.geo_mode, pg.blocks_per_wave,
                pg.rgbaz_runs, pg.big_windows, pg.listed ? 1 : 0, pg.list_recs ? 1 : 0, pg.fill_stride, pg.fill_per_wave);
  }
  return 0;
}
