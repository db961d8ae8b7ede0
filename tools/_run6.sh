timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06c_bench.json') if l.startswith('{')][-1])
print('value',d['value'],'errors',d['errors'])
for k,v in d['roofline'].items():
    if k.startswith('opfused') or k in('frac','launch_ms','step_t63_l16_us','step_t30_l8_us'): print(k,v)
PY
timeout 600 python -m pytest tests/test_gpu_fused_ops.py tests/test_gpu_determinism.py -x -q -m gpu 2>&1 | tail -3
