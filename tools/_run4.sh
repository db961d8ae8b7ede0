SPDY_COMM_FORCE=1 timeout 600 python bench.py --force-multi --no-extras --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r06_force_multi.log 2>&1
tail -5 gpurun_out/r06_force_multi.log | cut -c1-3000
