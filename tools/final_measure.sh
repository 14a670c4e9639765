#!/usr/bin/env bash
# One pass over everything DESIGN.md quotes; run on an MI355X box from the repo root.
# Writes gpurun_out/final/* (copy what is to be judged into profiles/).  usage: final_measure.sh [tag]
set -uo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$R"
out=gpurun_out/final; rm -rf $out gpurun_out/traffic; mkdir -p $out
(timeout 1200 python -m pytest tests -m gpu -q > $out/gpu_tests.log 2>&1; echo "exit $?" >> $out/gpu_tests.log); tail -3 $out/gpu_tests.log
# the PMC traffic first: bench.py reports roofline.traffic only from a file stamped with the sources it runs
bash tools/collect_traffic.sh > $out/traffic.log 2>&1; tail -2 $out/traffic.log; cp gpurun_out/traffic_r02.json profiles/traffic_r02.json
(timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?")
timeout 200 ./tools/kbench --sum --reps 30 > $out/kbench_rgba.log 2>&1
timeout 300 ./tools/kbench --sum --reps 8 --batch 16 eqd_rect_bc eqr_rect_bc eqr_rect_bc_rot eqr_rect_bc_pitch rect_rect_bc eqd_eqd_bc eqr_eqr_bc_rot rect_eqr_bc eqr_eqd_bl_rot eqr_rect_bl eqr_rect_nn > $out/kbench_rgba_batched.log 2>&1
timeout 200 ./tools/kbench --sum --reps 20 --channels 3 eqd_rect_bc eqr_rect_bc eqr_rect_bl eqr_rect_nn rect_eqr_bc > $out/kbench_rgb.log 2>&1
timeout 200 ./tools/kbench --sum --reps 20 --channels 5 eqd_rect_bc eqr_rect_bc eqr_rect_bl eqr_rect_nn rect_eqr_bc > $out/kbench_rgbaz.log 2>&1
timeout 200 ./tools/kbench --sum --reps 20 --channels 5 --post rect_eqr_bc > $out/kbench_rgbaz_post.log 2>&1
timeout 200 ./tools/kbench --sum --reps 20 --size 8192 --out-size 2048 --channels 3 eqr_rect_bc eqr_rect_bc_rot eqr_rect_bc_pitch > $out/kbench_cubemap_faces.log 2>&1
timeout 300 ./tools/staged_bench > $out/staged.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_bench -- python3 $R/bench.py --no-cpu-baseline > $R/$out/prof_bench.log 2>&1; echo "rocprof rc=$?"
cat $R/$out/prof_bench/*/*kernel_stats.csv
python3 $R/tools/kernel_trace_summary.py $(ls -t $R/$out/prof_bench/*/*kernel_trace.csv | head -1) | tee $R/$out/kernel_trace_by_launch_shape.txt
cat $R/$out/bench.json
