cd /root/repo
timeout 120 tools/kbench --reps 20 --distinct 4 --sum eqr_rect_bc_pitch | grep -v "^#"
echo "== config 5 shape: 8192^2 RGB equirect -> 2048^2 faces"
timeout 200 tools/kbench --size 8192 --out-size 2048 --channels 3 --reps 30 --distinct 2 --sum eqr_rect_bc eqr_rect_bc_rot eqr_rect_bc_pitch | grep -v "^#"
LRP_KERNEL=pixel timeout 200 tools/kbench --size 8192 --out-size 2048 --channels 3 --reps 10 --distinct 2 --sum eqr_rect_bc eqr_rect_bc_rot eqr_rect_bc_pitch | grep -v "^#"
