cd /root/repo
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for v in base ""; do
  echo "== variant '$v'"
  if [ -n "$v" ]; then export LD_LIBRARY_PATH=$PWD/tools/_ablate/$v; else unset LD_LIBRARY_PATH; fi
  timeout 120 tools/kbench --reps 48 --distinct 8 --sum eqd_rect_bc eqr_rect_bc eqr_rect_bc_rot rect_rect_bc | grep -v "^#"
done
done
