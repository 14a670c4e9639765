"""CPU: the measurement tools that turn profiler output into the tables DESIGN.md quotes (the launch-shape summary
read the wrong template argument; the roofline table is generated, not hand-copied)."""
import csv
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_frames_per_launch_reads_the_frame_loop_argument():
    k = _load("kernel_trace_summary")
    win = "void lrp::reproject_bicubic_win_kernel<{}>(lrp::KParams)"
    # <OutLens, InMode, QMode, CH, Frames, GeoRead[, SS]>: the frame loop is `Frames`, not the last argument
    assert k.frames_per_launch(win.format("0, 1, 0, 4, true, true"), 1, 16) == 16
    assert k.frames_per_launch(win.format("0, 1, 0, 4, false, true"), 1, 16) == 1  # a single GeoRead launch is ONE frame
    assert k.frames_per_launch(win.format("0, 1, 0, 4, false, true"), 16, 16) == 16  # 16 frames, a frame per grid row
    assert k.frames_per_launch(win.format("0, 3, 1, 4, false, false"), 1, 16) == 1
    assert k.frames_per_launch(win.format("0, 1, 0, 4, true, true, false"), 1, 16) == 16  # (round 5: a seventh argument, SS)
    assert k.frames_per_launch(win.format("0, 1, 0, 4, false, true, false"), 1, 16) == 1
    assert k.frames_per_launch(win.format("0, 1, 0, 4, false, false, true"), 1, 16) == 1
    assert k.frames_per_launch("void lrp::reproject_tile_kernel<0, 3, 1, 4, false, true>(lrp::KParams)", 16, 16) == 16
    assert k.frames_per_launch("void lrp::reproject_tile_kernel<0, 3, 0, 4, true, false>(lrp::KParams)", 1, 16) == 16
    assert k.frames_per_launch("lrp::(anonymous namespace)::corner_fill_kernel<5>(lrp::KParams)", 16, 16) == 16


def test_kernel_trace_summary_on_a_synthetic_trace(tmp_path):
    rows = [("void lrp::reproject_bicubic_win_kernel<0, 1, 0, 4, true, true>(lrp::KParams)", 1, 0, 1500000),
            ("void lrp::reproject_bicubic_win_kernel<0, 1, 0, 4, false, true>(lrp::KParams)", 1, 0, 120000),
            ("void lrp::reproject_bicubic_win_kernel<0, 1, 0, 4, false, true>(lrp::KParams)", 1, 0, 124000),
            ("lrp::(anonymous namespace)::synth_fill_kernel(float*, unsigned int, int, unsigned int, int)", 1, 0, 50000)]
    path = tmp_path / "kernel_trace.csv"
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Grid_Size_Y", "Workgroup_Size_Y", "Start_Timestamp", "End_Timestamp"])
        for name, gy, t0, t1 in rows:
            w.writerow([name, gy, 1, t0, t1])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_trace_summary.py"), str(path)], capture_output=True, text=True, check=True).stdout
    lines = [ln for ln in out.splitlines() if "reproject" in ln]
    assert len(lines) == 2 and "synth_fill" not in out
    single = [ln for ln in lines if "false, true>" in ln][0].split()
    batched = [ln for ln in lines if "true, true>" in ln][0].split()
    assert single[-5:] == ["1", "2", "122.0", "120.0", "122.0"], single  # frames / launch, launches, avg, min, us / frame
    assert batched[-5:] == ["16", "1", "1500.0", "1500.0", "93.8"], batched


@pytest.mark.parametrize("name", ["r04_bench.json", "r06_bench.json"])  # (round 4's single line; round 6's detail file)
def test_roofline_table_is_generated_from_the_bench_line(tmp_path, name):
    bench = os.path.join(ROOT, "profiles", name)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_table.py"), bench], capture_output=True, text=True, check=True).stdout
    rec = json.load(open(bench))
    head = [ln for ln in out.splitlines() if ln.startswith("| fisheye_to_rect_bicubic:")][0]
    cells = [c.strip() for c in head.strip("|").split("|")]
    assert cells[1] == "537" and float(cells[3]) == round(rec["roofline"]["frac"], 3) and float(cells[2]) == round(rec["roofline"]["us_per_frame"], 1)
    assert sum(1 for ln in out.splitlines() if ln.startswith("| ") and ":" in ln.split("|")[1]) == 1 + len(rec["secondary"])


def test_the_drivers_bench_line_is_compact_and_complete():
    """bench.py's LAST stdout line is what the driver parses (BENCH_rNN.json): one JSON object with the contract keys, the
    roofline and cpu_baseline objects and one {frac, us_per_frame} pair per secondary workload — and SHORT.  Round 5's line was
    24.7 KB (seven secondaries x two traffic blocks x four prose notes) and the driver recorded `parsed: null`.  Built here from
    that very result (profiles/r05_bench.json is a full detail record)."""
    sys.path.insert(0, ROOT)
    import bench

    detail = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    assert len(json.dumps(detail)) > 20000  # (the canned result is the one that broke the driver)
    line = bench.compact_line(detail, "bench_detail.json")
    assert "\n" not in line and len(line) < bench.COMPACT_LIMIT < 6000
    rec = json.loads(line)
    assert set(rec) == set(bench.TOP_KEYS)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        want = detail[k]
        assert rec[k] == (pytest.approx(want, rel=1e-5) if isinstance(want, float) else want), k
    assert rec["config"]["workload"] == detail["config"]["workload"] and "model" not in rec["config"]
    roof = rec["roofline"]
    assert set(roof) == set(bench.ROOFLINE_KEYS)
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-4)
    assert roof["achieved"] == pytest.approx(roof["algorithmic_bytes_per_launch"] / (roof["kernel_ms_avg"] * 1e-3) / 1e9, rel=1e-4)
    assert roof["traffic"] == pytest.approx(detail["roofline"]["traffic"], rel=1e-5)
    cpu = rec["cpu_baseline"]
    assert set(cpu) == set(bench.CPU_KEYS) and cpu["kind"] == "port" and cpu["cores"] == 16 and len(cpu["sample"]) <= 200
    assert set(rec["secondary"]) == set(detail["secondary"])
    for name, r in rec["secondary"].items():
        assert set(r) == {"frac", "us_per_frame"} and 0 < r["frac"] < 1, name  # (the cubemap's entry is frac_source_once, below 1)
    assert rec["outputs_match_golden"] is True and rec["detail_file"] == "bench_detail.json"
    assert "NaN" not in line and "Infinity" not in line

    # an N > 1 line: no cpu_baseline, no secondary, no single-launch legs — still every contract key
    multi = {k: v for k, v in detail.items() if k not in ("cpu_baseline", "staged")}
    multi["n_gpus"], multi["secondary"] = 8, {}
    multi["roofline"] = {k: v for k, v in detail["roofline"].items() if not k.startswith("single_launch")}
    rec8 = json.loads(bench.compact_line(multi))
    assert set(rec8) == set(bench.TOP_KEYS) - {"cpu_baseline", "detail_file"} and rec8["secondary"] == {}
    assert "single_launch_us" not in rec8["roofline"] and rec8["roofline"]["frac"] == roof["frac"]

    # non-finite figures never reach the line as NaN / Infinity tokens, and an oversized line is refused, not printed
    broken = dict(detail, value=float("nan"))
    assert json.loads(bench.compact_line(broken))["value"] is None
    fat = dict(detail, secondary={f"workload_{i}_{'x' * 40}": {"frac": 0.5, "us_per_frame": 100.0} for i in range(60)})
    with pytest.raises(AssertionError):
        bench.compact_line(fat)


def test_every_bench_workload_has_a_row_label_and_the_compact_line_of_round_6_is_on_file():
    """tools/roofline_table.py names every workload bench.py can measure; profiles/r06_bench_line.json — what the driver's parser
    was given on the measurement pass — is one line, below the limit, and agrees with the detail file it points to."""
    sys.path.insert(0, ROOT)
    import bench

    spec = importlib.util.spec_from_file_location("roofline_table", os.path.join(ROOT, "tools", "roofline_table.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert set(bench.WORKLOADS) <= set(mod.CONFIG_OF), sorted(set(bench.WORKLOADS) - set(mod.CONFIG_OF))
    assert set(bench.DEFAULT_SECONDARY.split(",")) <= set(bench.WORKLOADS)
    text = open(os.path.join(ROOT, "profiles", "r06_bench_line.json")).read().rstrip("\n")
    last = text.splitlines()[-1]
    assert len(last) < bench.COMPACT_LIMIT
    rec, detail = json.loads(last), json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    assert set(rec) <= set(bench.TOP_KEYS) and rec["value"] == pytest.approx(detail["value"], rel=1e-5)
    assert json.loads(bench.compact_line(detail, rec.get("detail_file"))) == rec  # the line IS compact_line(detail)
    assert set(rec["secondary"]) == set(bench.DEFAULT_SECONDARY.split(","))
