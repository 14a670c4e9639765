#!/usr/bin/env bash
# Per-pass account of the window kernel's tiers: SQ counters of three GeoRead launches — the coefficient tier
# (headline, 16-frame launch), the raw-tap tier on a ~1:1 mapping (rect -> rect, 16-frame launch) and on a cubemap's side face
# (8192^2 RGB -> 2048^2, single launches) — divided by the 16 x 4 passes each launch renders.  Run on an MI355X box from the
# repo root; writes gpurun_out/account/{counters.txt,table.md}.  Counter passes are separate rocprofv3 runs (--pmc with
# --kernel-trace only).
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out=$R/gpurun_out/account; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/coef$i -- $R/tools/kbench --batch 16 --reps 4 --warmup 20 --distinct 16 eqd_rect_bc > $out/coef$i.log 2>&1
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/rect$i -- $R/tools/kbench --batch 16 --reps 4 --warmup 20 --distinct 16 rect_rect_bc > $out/rect$i.log 2>&1
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/side$i -- $R/tools/kbench --reps 8 --warmup 12 --size 8192 --out-size 2048 --channels 3 --distinct 4 eqr_rect_bc_rot > $out/side$i.log 2>&1
  echo "pass $i done"
done
python3 $R/tools/pmc_summary.py $out > $out/counters.txt
python3 $R/tools/raw_tap_table.py $out/counters.txt > $out/table.md
cat $out/table.md
