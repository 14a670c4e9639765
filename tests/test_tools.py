"""CPU: the measurement tools that turn profiler output into the tables DESIGN.md quotes (the launch-shape summary
read the wrong template argument; the roofline table is generated, not hand-copied)."""
import csv
import importlib.util
import json
import os
import subprocess
import sys

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


def test_roofline_table_is_generated_from_the_bench_line(tmp_path):
    bench = os.path.join(ROOT, "profiles", "r04_bench.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_table.py"), bench], capture_output=True, text=True, check=True).stdout
    rec = json.load(open(bench))
    head = [ln for ln in out.splitlines() if ln.startswith("| fisheye_to_rect_bicubic:")][0]
    cells = [c.strip() for c in head.strip("|").split("|")]
    assert cells[1] == "537" and float(cells[3]) == round(rec["roofline"]["frac"], 3) and float(cells[2]) == round(rec["roofline"]["us_per_frame"], 1)
    assert sum(1 for ln in out.splitlines() if ln.startswith("| ") and ":" in ln.split("|")[1]) == 1 + len(rec["secondary"])
