(time timeout 900 python bench.py --no-extras --no-cpu-baseline) > gpurun_out/r06_pmc_live.log 2>&1
grep '^{"metric"' gpurun_out/r06_pmc_live.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['traffic'], r['traffic_over_algorithmic'], r['traffic_all_kernels']); print(r['traffic_source'])"
grep real gpurun_out/r06_pmc_live.log
(time timeout 900 python bench.py --res t63 --no-extras --no-cpu-baseline) 2>&1 | grep -E '^\{"metric"|real' | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], r['traffic'], r['traffic_over_algorithmic'], r['traffic_all_kernels']); print(r['traffic_source'][:200])
    else: print(l.strip())"
