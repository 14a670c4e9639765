cd /root/repo
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
LD_LIBRARY_PATH=$PWD/tools/_ablate/tiers timeout 120 tools/kbench --reps 2 --warmup 0 --distinct 1 eqd_rect_bc eqr_rect_bc | grep tier
for v in base ""; do
  echo "== variant '$v'"
  if [ -n "$v" ]; then export LD_LIBRARY_PATH=$PWD/tools/_ablate/$v; else unset LD_LIBRARY_PATH; fi
  timeout 120 tools/kbench --reps 30 --distinct 8 --sum eqd_rect_bc eqr_rect_bc eqr_rect_bc_rot eqd_rect_bl eqr_rect_bl | grep -v "^#"
done
