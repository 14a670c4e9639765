cd /root/repo
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for v in base ""; do
  echo "== variant '$v'"
  if [ -n "$v" ]; then export LD_LIBRARY_PATH=$PWD/tools/_ablate/$v; else unset LD_LIBRARY_PATH; fi
  timeout 120 tools/kbench --reps 40 --distinct 8 --sum eqd_rect_bc eqr_rect_bc eqr_rect_bc_rot rect_rect_bc eqd_eqd_bc eqr_eqr_bc_rot rect_eqr_bc | grep -v "^#"
  timeout 120 tools/kbench --reps 20 --distinct 4 --channels 3 --sum eqd_rect_bc eqr_rect_bc eqr_rect_bc_rot | grep -v "^#"
done
