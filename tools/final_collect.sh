#!/usr/bin/env bash
# Copies the summaries of gpurun_out/final (tools/final_measure.sh) into profiles/ under this round's names.  usage: final_collect.sh [round: r06]
set -euo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$R"
f=gpurun_out/final; r=${1:-r06}
cp gpurun_out/traffic_${r}.json profiles/traffic_${r}.json # (written on the GPU box; only gpurun_out/ travels back)
cp $f/bench.json profiles/${r}_bench.json            # the detail file
cp $f/bench_line.json profiles/${r}_bench_line.json  # the compact line (what BENCH_rNN.json parses)
cat "$(ls -t $f/prof_bench/*/*kernel_stats.csv | head -1)" > profiles/${r}_bench_kernel_stats.csv # (the newest: gpurun merges the passes of a round into one directory)
cp $f/kernel_trace_by_launch_shape.txt profiles/${r}_bench_kernel_trace_by_launch_shape.txt
cp $f/roofline.md profiles/${r}_roofline.md
(echo "# single-frame launches, geometry cache on (default)"; cat $f/kbench_rgba_single.log; echo; echo "# single-frame launches, geometry cache off (every launch computes its coordinates)"; cat $f/kbench_rgba_single_geo0.log) > profiles/${r}_kbench_rgba_single.txt
(echo "# 16-frame launches, default"; cat $f/kbench_rgba_batched.log; echo; echo "# 16-frame launches, geometry cache off"; cat $f/kbench_rgba_batched_geo0.log) > profiles/${r}_kbench_rgba_batched.txt
(cat $f/kbench_rgb_batched.log; echo; cat $f/kbench_rgbaz_batched.log; echo; cat $f/kbench_rgbaz_single.log) > profiles/${r}_kbench_rgb_rgbaz.txt
cp $f/kbench_configs3_lists.log profiles/${r}_kbench_configs3_lists.txt
(cat $f/kbench_cubemap_faces.log; echo "# geometry cache off"; cat $f/kbench_cubemap_faces_geo0.log; echo "# the whole cubemap (lrp_reproject_multi_device)"; grep multi_fork $f/cubemap_bench.log) > profiles/${r}_kbench_cubemap_faces.txt
cp $f/kbench_supersampling.log profiles/${r}_kbench_supersampling.txt
[ -f $f/kbench_tap_dma.log ] && cp $f/kbench_tap_dma.log profiles/${r}_kbench_tap_dma.txt
cp $f/kbench_two_streams.log profiles/${r}_kbench_two_streams.txt
cp $f/sq_counters.txt profiles/${r}_sq_counters.txt
cp $f/staged.log profiles/${r}_staged_pcie.txt
cp $f/fov_sweep.log profiles/${r}_rect_eqr_fov_sweep.txt
[ -f $f/hbm_stream.log ] && cp $f/hbm_stream.log profiles/${r}_hbm_stream_microbench.txt
[ -f $f/policy_check.txt ] && cp $f/policy_check.txt profiles/${r}_policy_check.txt
[ -f $f/tier_account.md ] && cp $f/tier_account.md profiles/${r}_tier_account.md && cp $f/tier_account_counters.txt profiles/${r}_tier_account_counters.txt
tail -3 $f/gpu_tests.log > profiles/${r}_gpu_tests_tail.txt
echo "collected into profiles/${r}_*"
