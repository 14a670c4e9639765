cd /root/repo
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for v in base ""; do
  echo "== variant '$v'"
  if [ -n "$v" ]; then export LD_LIBRARY_PATH=$PWD/tools/_ablate/$v; else unset LD_LIBRARY_PATH; fi
  timeout 120 tools/kbench --reps 20 --distinct 4 --channels 5 --sum eqd_rect_bc eqr_rect_bc eqr_rect_bc_rot rect_eqr_bc | grep -v "^#"
  LRP_KERNEL=tile timeout 120 tools/kbench --reps 20 --distinct 4 --channels 3 --sum eqd_rect_bc rect_eqr_bc | grep -v "^#"
done
