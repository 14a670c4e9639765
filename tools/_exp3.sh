cd /root/repo
for s in 1 2 3 4 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --streams $s 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('streams',d['config']['streams'],'value',round(d['value']),'ms/step',round(d['ms_per_step'],3),'kernel_avg_us',round(d['roofline']['kernel_ms_avg']*1e3,1),'frac',round(d['roofline']['frac'],3))"
done
