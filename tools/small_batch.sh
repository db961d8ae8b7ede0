#!/usr/bin/env bash
# GPU box: fused vs four-kernel path for the model-shaped small batches.
for b in 8 48 73 91 256 512; do for f in 0 1; do
  timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --fused $f --batch $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_kernels_ms']
print('B=%4d fused=$f  %8.3f M rt/s  %7.1f us/step  ' % (d['config']['fields_per_gpu'], d['value']/1e6, d['ms_per_step']*1e3), {a: round(v*1e3,1) for a,v in k.items()})"
done; done
