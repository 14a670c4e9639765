#!/usr/bin/env bash
# SQ counters of BASELINE configs[3] (rect -> equirect bicubic, 16-frame launches): RGBAZ + tonemap, RGBAZ without it, RGBA.
# Run on an MI355X box from the repo root; writes gpurun_out/sq_configs3.txt (-> profiles/rNN_sq_counters_configs3.txt).
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out=$R/gpurun_out/sqc3; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/rgbaz_post$i -- $R/tools/kbench --batch 16 --reps 4 --warmup 20 --distinct 16 --channels 5 --post rect_eqr_bc > $out/a$i.log 2>&1
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/rgbaz$i -- $R/tools/kbench --batch 16 --reps 4 --warmup 20 --distinct 16 --channels 5 rect_eqr_bc > $out/b$i.log 2>&1
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/rgba$i -- $R/tools/kbench --batch 16 --reps 4 --warmup 20 --distinct 16 rect_eqr_bc > $out/c$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $out | grep -A9 "reproject_bicubic_win_kernel<4, 0, 0, [45], false, true, false>" > $R/gpurun_out/sq_configs3.txt
cat $R/gpurun_out/sq_configs3.txt
