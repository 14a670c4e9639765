#!/usr/bin/env bash
# SQ counter passes over the headline kernels (run on an MI355X box from the repo root);
# summary -> gpurun_out/sq/summary.txt (copied to profiles/r01_sq_counters_window_kernel.txt).
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out=$R/gpurun_out/sq; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_IFETCH GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- $R/tools/kbench --reps 6 --warmup 20 --distinct 8 eqd_rect_bc eqr_rect_bc eqr_rect_bc_rot > $out/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 $R/tools/pmc_summary.py $out > $out/summary.txt
cat $out/summary.txt
