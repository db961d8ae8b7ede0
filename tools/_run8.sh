python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --force-multi > gpurun_out/r06_torchrun1.log 2>&1
echo rc=$?
grep '^{"metric"' gpurun_out/r06_torchrun1.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], d['errors']); print(list(d['multi_gpu'].keys())); print(d['multi_gpu']['sharded_step']['transposed_form'].get('us_with_exchanges'))"
tail -3 gpurun_out/r06_torchrun1.log | cut -c1-300
