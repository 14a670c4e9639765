cd /root/repo
for v in "" cap608 cap512; do
  echo "== variant '$v' distinct=8"
  if [ -n "$v" ]; then export LD_LIBRARY_PATH=$PWD/tools/_ablate/$v; else unset LD_LIBRARY_PATH; fi
  timeout 120 tools/kbench --reps 48 --distinct 8 --sum eqd_rect_bc eqr_rect_bc eqr_rect_bc_rot rect_rect_bc | grep -v "^#"
done
