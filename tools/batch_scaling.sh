#!/usr/bin/env bash
# GPU box: throughput of the fused path vs batch size (fields per launch).
for b in ${@:-1536 3072 6144 12288 24576 49152}; do
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --fused 1 --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms']
print('B=%6d  %.2f M rt/s  s2g %.1f us  g2s %.1f us' % (d['config']['fields_per_gpu'], d['value']/1e6, k.get('s2g_fused',0)*1e3, k.get('g2s_fused',0)*1e3))"
done
