cd /root/repo
for v in "" plainstore; do
  echo "== variant '$v' distinct=8"
  if [ -n "$v" ]; then export LD_LIBRARY_PATH=$PWD/tools/_ablate/$v; else unset LD_LIBRARY_PATH; fi
  timeout 120 tools/kbench --reps 48 --distinct 8 eqd_rect_bc eqr_rect_bc eqr_rect_bc_rot eqr_rect_nn eqr_rect_bl | grep -v "^#"
done
