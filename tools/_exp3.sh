cd /root/repo
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for q in 0 1; do
  echo "== LRP_QUAD=$q"
  LRP_QUAD=$q timeout 120 tools/kbench --reps 40 --distinct 8 --sum eqd_rect_bl eqd_rect_nn eqr_rect_bl eqr_rect_nn rect_eqr_nn rect_eqr_bl | grep -v "^#"
  LRP_QUAD=$q timeout 120 tools/kbench --reps 20 --distinct 4 --channels 3 --sum eqd_rect_bc eqr_rect_bc eqr_rect_bl | grep -v "^#"
  LRP_QUAD=$q timeout 120 tools/kbench --reps 20 --distinct 4 --channels 5 --sum eqd_rect_bc eqr_rect_bc eqr_rect_bl | grep -v "^#"
done
