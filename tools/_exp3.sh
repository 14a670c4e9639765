cd /root/repo
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for q in 0 1; do
  echo "== LRP_QUAD=$q"
  LRP_QUAD=$q timeout 120 tools/kbench --reps 40 --distinct 8 --sum eqr_eqd_bl_rot eqd_eqd_bc | grep -v "^#"
  LRP_QUAD=$q timeout 120 tools/kbench --reps 20 --distinct 4 --channels 3 --sum eqr_eqd_bl_rot eqd_eqd_bc| grep -v "^#"
done
