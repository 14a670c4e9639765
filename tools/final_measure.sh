#!/usr/bin/env bash
# ONE pass over everything DESIGN.md quotes; run on an MI355X box from the repo root.  usage: final_measure.sh [round: r06]
# Writes gpurun_out/final/* (copy what is to be judged into profiles/ with tools/final_collect.sh <round>).
set -uo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$R"
round="${1:-r06}"
out=gpurun_out/final; rm -rf $out gpurun_out/traffic; mkdir -p $out
(timeout 2400 python -m pytest tests -m gpu -q > $out/gpu_tests.log 2>&1; echo "exit $?" >> $out/gpu_tests.log); tail -3 $out/gpu_tests.log
# the PMC traffic first: bench.py reports roofline.traffic only from a file stamped with the sources it runs
bash tools/collect_traffic.sh $round > $out/traffic.log 2>&1; tail -2 $out/traffic.log; cp gpurun_out/traffic_$round.json profiles/traffic_$round.json
# bench.json = the detail file (everything measured); bench_line.json = the compact line the driver reads (stdout)
(timeout 1200 python bench.py --detail-file $out/bench.json > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"; tail -c 4200 $out/bench_line.json)
ALL="eqd_rect_bc eqr_rect_bc eqr_rect_bc_rot eqr_rect_bc_pitch eqr_rect_bc_gen rect_rect_bc eqd_eqd_bc eqr_eqr_bc_rot eqr_eqd_bc_rot rect_eqd_bc rect_eqr_bc eqr_eqd_bl_rot eqr_rect_bl eqr_rect_nn eqr_rect_bl_rot eqr_rect_nn_rot"
BC="eqd_rect_bc eqr_rect_bc eqr_rect_bc_rot eqr_rect_bc_pitch eqr_rect_bc_gen rect_rect_bc eqr_eqd_bc_rot rect_eqr_bc"
timeout 300 ./tools/kbench --sum --reps 32 --distinct 16 $ALL > $out/kbench_rgba_single.log 2>&1
timeout 300 ./tools/kbench --sum --reps 32 --distinct 16 --geo 0 $ALL > $out/kbench_rgba_single_geo0.log 2>&1
timeout 400 ./tools/kbench --sum --reps 8 --batch 16 --distinct 16 $ALL > $out/kbench_rgba_batched.log 2>&1
timeout 400 ./tools/kbench --sum --reps 8 --batch 16 --distinct 16 --geo 0 $ALL > $out/kbench_rgba_batched_geo0.log 2>&1
timeout 300 ./tools/kbench --sum --reps 8 --batch 16 --distinct 16 --channels 3 $BC eqr_rect_bl eqr_rect_nn > $out/kbench_rgb_batched.log 2>&1
timeout 300 ./tools/kbench --sum --reps 8 --batch 16 --distinct 16 --channels 5 $BC eqr_rect_bl eqr_rect_nn > $out/kbench_rgbaz_batched.log 2>&1
timeout 300 ./tools/kbench --sum --reps 24 --distinct 16 --channels 5 $BC > $out/kbench_rgbaz_single.log 2>&1
for L in 1 0; do
  (echo "# geo_lists=$L"; timeout 200 ./tools/kbench --sum --reps 8 --batch 16 --distinct 16 --channels 5 --post --set geo_lists=$L rect_eqr_bc; timeout 200 ./tools/kbench --sum --reps 24 --distinct 16 --channels 5 --post --set geo_lists=$L rect_eqr_bc;
   timeout 200 ./tools/kbench --sum --reps 8 --batch 16 --distinct 16 --set geo_lists=$L rect_eqr_bc rect_eqd_bc; timeout 200 ./tools/kbench --sum --reps 24 --distinct 16 --set geo_lists=$L rect_eqr_bc rect_eqd_bc) >> $out/kbench_configs3_lists.log 2>&1
done
(for T in 1 0; do
   echo "# win_tapdma=$T: rect_eqr_bc single RGBA / batch16 RGBA / single RGBAZ + tonemap / batch16 RGBAZ + tonemap / single RGB"
   timeout 200 ./tools/kbench --sum --reps 24 --distinct 16 --set win_tapdma=$T rect_eqr_bc | grep -v "^#"; timeout 200 ./tools/kbench --sum --reps 8 --batch 16 --distinct 16 --set win_tapdma=$T rect_eqr_bc | grep -v "^#"
   timeout 200 ./tools/kbench --sum --reps 24 --distinct 16 --channels 5 --post --set win_tapdma=$T rect_eqr_bc | grep -v "^#"; timeout 200 ./tools/kbench --sum --reps 8 --batch 16 --distinct 16 --channels 5 --post --set win_tapdma=$T rect_eqr_bc | grep -v "^#"
   timeout 200 ./tools/kbench --sum --reps 24 --distinct 16 --channels 3 --set win_tapdma=$T rect_eqr_bc | grep -v "^#"
 done
 for B in 1 0 2; do
   echo "# geo_big=$B (1: by the census of the entry; 0: never; 2: always): rect_eqd_bc single RGBA / batch16 RGBA / batch16 RGBAZ + tonemap, eqr_rect_bc_gen single RGBA, 8192^2 -> 2048^2 RGB side face / pole face"
   timeout 200 ./tools/kbench --sum --reps 24 --distinct 16 --set geo_big=$B rect_eqd_bc | grep -v "^#"; timeout 200 ./tools/kbench --sum --reps 8 --batch 16 --distinct 16 --set geo_big=$B rect_eqd_bc | grep -v "^#"
   timeout 200 ./tools/kbench --sum --reps 8 --batch 16 --distinct 16 --channels 5 --post --set geo_big=$B rect_eqd_bc | grep -v "^#"
   timeout 200 ./tools/kbench --sum --reps 24 --distinct 16 --set geo_big=$B eqr_rect_bc_gen | grep -v "^#"
   timeout 200 ./tools/kbench --sum --reps 20 --size 8192 --out-size 2048 --channels 3 --distinct 4 --set geo_big=$B eqr_rect_bc_rot eqr_rect_bc_pitch | grep -v "^#"
 done) > $out/kbench_tap_dma.log 2>&1
timeout 200 ./tools/kbench --sum --reps 20 --size 8192 --out-size 2048 --channels 3 --distinct 4 eqr_rect_bc eqr_rect_bc_rot eqr_rect_bc_pitch > $out/kbench_cubemap_faces.log 2>&1
timeout 200 ./tools/kbench --sum --reps 20 --size 8192 --out-size 2048 --channels 3 --distinct 4 --geo 0 eqr_rect_bc eqr_rect_bc_rot eqr_rect_bc_pitch > $out/kbench_cubemap_faces_geo0.log 2>&1
timeout 300 python3 tools/cubemap_bench.py 30 > $out/cubemap_bench.log 2>&1
SSW="eqd_rect_bc eqr_rect_bc eqr_rect_bc_gen rect_eqr_bc eqd_eqd_bc rect_rect_bc"
for N in "2 2048" "3 1365" "4 1024"; do
  set -- $N
  (echo "# num_samples $1, 4096^2 -> $2^2 (the reference's --scale / --samples pairs, src/main.cpp:192-196), single launches"
   echo "# window kernel's SS instantiations reading their entry of sub-samples (default)"; timeout 300 ./tools/kbench --sum --reps 16 --distinct 16 --ns $1 --size 4096 --out-size $2 $SSW | grep -v "^#"
   echo "# ... geometry cache off (every launch computes its sub-sample coordinates)"; timeout 300 ./tools/kbench --sum --reps 16 --distinct 16 --geo 0 --ns $1 --size 4096 --out-size $2 $SSW | grep -v "^#"
   echo "# ... win_ss=0: the tile kernel"; timeout 300 ./tools/kbench --sum --reps 10 --distinct 16 --ns $1 --size 4096 --out-size $2 --set win_ss=0 $SSW | grep -v "^#"
   echo "# ... 16-frame launches (default)"; timeout 300 ./tools/kbench --sum --reps 4 --batch 16 --distinct 16 --ns $1 --size 4096 --out-size $2 eqd_rect_bc eqr_rect_bc_gen | grep -v "^#") >> $out/kbench_supersampling.log 2>&1
done
for N in "2 2048" "3 1365" "4 1024"; do
  set -- $N
  (echo "# nearest / bilinear, num_samples $1, 4096^2 -> $2^2: reading the entry of sub-samples (default: the gather kernel, a lane per sub-sample; equirect -> rect without a rotation computes) / geometry cache off (the tile kernel computes)"
   timeout 300 ./tools/kbench --sum --reps 16 --distinct 16 --ns $1 --size 4096 --out-size $2 eqr_rect_bl eqr_rect_nn eqr_eqd_bl_rot eqr_rect_bl_rot eqr_rect_nn_rot | grep -v "^#"
   timeout 300 ./tools/kbench --sum --reps 16 --distinct 16 --geo 0 --ns $1 --size 4096 --out-size $2 eqr_rect_bl eqr_rect_nn eqr_eqd_bl_rot eqr_rect_bl_rot eqr_rect_nn_rot | grep -v "^#") >> $out/kbench_supersampling.log 2>&1
done
(echo "# num_samples 5 (the tile kernel) and 1 for reference, 4096^2 -> 2048^2"; timeout 300 ./tools/kbench --sum --reps 6 --distinct 16 --ns 5 --size 4096 --out-size 819 eqd_rect_bc | grep -v "^#"; timeout 300 ./tools/kbench --sum --reps 20 --distinct 16 --ns 1 --size 4096 --out-size 2048 eqd_rect_bc eqr_rect_bc | grep -v "^#"
 echo "# num_samples 2 at scale 1 (4096^2 -> 4096^2)"; timeout 300 ./tools/kbench --sum --reps 8 --distinct 16 --ns 2 eqd_rect_bc eqr_rect_bc | grep -v "^#") >> $out/kbench_supersampling.log 2>&1
for S in 1 2; do timeout 200 ./tools/kbench --reps 64 --distinct 16 --streams $S eqd_rect_bc eqr_rect_bc eqr_eqd_bc_rot rect_eqr_bc; done > $out/kbench_two_streams.log 2>&1
timeout 300 ./tools/staged_bench > $out/staged.log 2>&1
(echo "# RGBA"; python3 tools/fov_sweep.py 4 2>&1 | grep focal; echo "# RGBAZ + tonemap"; python3 tools/fov_sweep.py 5 post 2>&1 | grep focal) > $out/fov_sweep.log
timeout 100 tools/microbench/hbm_stream > $out/hbm_stream.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_bench -- python3 $R/bench.py --no-cpu-baseline --no-staged --detail-file "" > $R/$out/prof_bench.log 2>&1; echo "rocprof rc=$?"
cat $R/$out/prof_bench/*/*kernel_stats.csv | cut -c1-200
python3 $R/tools/kernel_trace_summary.py $(ls -t $R/$out/prof_bench/*/*kernel_trace.csv | head -1) | tee $R/$out/kernel_trace_by_launch_shape.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$out/sq/b$i -- $R/tools/kbench --batch 16 --reps 4 --warmup 20 --distinct 16 eqd_rect_bc eqr_rect_bc_gen > $R/$out/sq_b$i.log 2>&1
  echo "sq batched pass $i rc=$?"
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$out/sq/s$i -- $R/tools/kbench --reps 8 --warmup 20 --distinct 16 eqd_rect_bc eqr_rect_bc_gen > $R/$out/sq_s$i.log 2>&1
  echo "sq single pass $i rc=$?"
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$out/sq/f$i -- $R/tools/kbench --reps 6 --warmup 6 --size 8192 --out-size 2048 --channels 3 --distinct 4 eqr_rect_bc_rot eqr_rect_bc_pitch > $R/$out/sq_f$i.log 2>&1
  echo "sq faces pass $i rc=$?"
done
python3 $R/tools/pmc_summary.py $R/$out/sq > $R/$out/sq_counters.txt
cd $R
(timeout 1800 python3 tools/policy_check.py 0.05 2>&1 | grep -v amdgpu.ids > $out/policy_check.txt; echo "policy_check rc=$?" >> $out/policy_check.txt)
(bash tools/raw_tap_account.sh > $out/account.log 2>&1; cp gpurun_out/account/table.md $out/tier_account.md; cp gpurun_out/account/counters.txt $out/tier_account_counters.txt)
python3 tools/roofline_table.py $out/bench.json $out/kernel_trace_by_launch_shape.txt > $out/roofline.md
cat $out/bench_line.json
