#!/usr/bin/env bash
# One pass over everything DESIGN.md quotes; run on an MI355X box from the repo root.
# Writes gpurun_out/final/*.  (tools/collect_traffic.sh is run separately.)
set -uo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$R"
out=gpurun_out/final; rm -rf $out; mkdir -p $out
(timeout 1200 python -m pytest tests -m gpu -q > $out/gpu_tests.log 2>&1; echo "exit $?" >> $out/gpu_tests.log); tail -3 $out/gpu_tests.log
timeout 200 ./tools/kbench --sum --reps 30 > $out/kbench_rgba.log 2>&1
timeout 300 ./tools/kbench --sum --reps 8 --batch 16 eqd_rect_bc eqr_rect_bc eqr_rect_bc_rot rect_rect_bc eqr_eqd_bl_rot eqr_rect_bl eqr_rect_nn > $out/kbench_rgba_batched.log 2>&1
timeout 200 ./tools/kbench --sum --reps 20 --channels 3 eqd_rect_bc eqr_rect_bc eqr_rect_bl eqr_rect_nn rect_eqr_bc > $out/kbench_rgb.log 2>&1
timeout 200 ./tools/kbench --sum --reps 20 --channels 5 eqd_rect_bc eqr_rect_bc eqr_rect_bl eqr_rect_nn rect_eqr_bc > $out/kbench_rgbaz.log 2>&1
LRP_KERNEL=pixel timeout 200 ./tools/kbench --sum --reps 20 eqd_rect_bc eqr_rect_bc eqr_rect_bl eqr_rect_nn eqd_rect_nn eqd_rect_bl > $out/kbench_pixel_kernel.log 2>&1
timeout 300 ./tools/staged_bench > $out/staged.log 2>&1
(timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?")
for w in equirect_to_rect_bicubic equirect_to_fisheye_bilinear equirect_to_rect_nearest; do timeout 300 python bench.py --workload $w --no-cpu-baseline >> $out/bench_other_workloads.json 2>> $out/bench.err; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_bench -- python3 $R/bench.py --no-cpu-baseline > $R/$out/prof_bench.log 2>&1; echo "rocprof rc=$?"
cat $R/$out/prof_bench/*/*kernel_stats.csv
cat $R/$out/bench.json
