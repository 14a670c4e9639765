cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for q in 0 1; do
for d in 1 8; do
  echo "== LRP_QUAD=$q distinct=$d"
  LRP_QUAD=$q timeout 120 tools/kbench --reps 48 --distinct $d --sum eqd_rect_bc eqr_rect_bc rect_rect_bc eqd_eqd_bc | grep -v "^#"
done
done
