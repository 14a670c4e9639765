cd /root/repo
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for k in "" tile; do
  echo "== LRP_KERNEL=$k"
  LRP_KERNEL=$k timeout 120 tools/kbench --reps 20 --distinct 4 --channels 3 --sum eqd_rect_bc eqr_rect_bc eqr_rect_bc_rot rect_rect_bc eqd_eqd_bc rect_eqr_bc | grep -v "^#"
done
