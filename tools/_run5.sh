mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r06b_pytest_gpu.log
cat gpurun_out/r06b_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r06b_bench_t30.json 2> gpurun_out/r06b_bench_t30.err
tail -c 600 gpurun_out/r06b_bench_t30.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06b_bench_t30.json') if l.startswith('{')][-1])
print('value',d['value'],'errors',d['errors'])
print(json.dumps(d['roofline'],indent=0)[:3000])
print(json.dumps(d['cpu_baseline'],indent=0)[:800])
PY
