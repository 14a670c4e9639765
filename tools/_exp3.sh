cd /root/repo
for v in base ""; do
  echo "== variant '$v'"
  if [ -n "$v" ]; then export LD_LIBRARY_PATH=$PWD/tools/_ablate/$v; else unset LD_LIBRARY_PATH; fi
  timeout 120 tools/kbench --reps 30 --distinct 8 --sum rect_eqr_bc rect_eqrp_bc eqd_rect_bc rect_rect_bc eqd_eqd_bc | grep -v "^#"
  timeout 120 tools/kbench --reps 20 --distinct 4 --channels 5 --sum eqd_rect_bc rect_eqr_bc | grep -v "^#"
  LRP_KERNEL=tile timeout 120 tools/kbench --reps 20 --distinct 4 --sum eqd_rect_bc rect_eqr_bc | grep -v "^#"
done
