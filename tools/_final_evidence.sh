# GPU box: round-6 evidence set -> gpurun_out/r06_* (copied to profiles/ afterwards)
set -x
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r06_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r06_bench_t30.json 2> gpurun_out/r06_bench_t30.err
timeout 900 python bench.py --res t63 > gpurun_out/r06_bench_t63.json 2> gpurun_out/r06_bench_t63.err
timeout 900 bash tools/profile_round.sh r06_fused_t30 > /dev/null 2>&1
cp gpurun_out/r06_fused_t30_pmc.json profiles/pmc_traffic.json 2>/dev/null
timeout 900 bash tools/profile_round.sh r06_fused_t63 --res t63 > /dev/null 2>&1
cp gpurun_out/r06_fused_t63_pmc.json gpurun_out/r06_pmc_traffic.json 2>/dev/null
timeout 900 bash tools/profile_step.sh r06_dynamics_step_kernels > /dev/null 2>&1
timeout 600 python tools/t63_f1_launches.py > gpurun_out/r06_t63_f1_launches.txt 2>&1
timeout 1500 python tools/soak_determinism.py 20000 > gpurun_out/r06_soak_determinism.txt 2>&1
SPDY_COMM_FORCE=1 timeout 600 python bench.py --force-multi --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"' > gpurun_out/r06_bench_force_multi.json
ls -la gpurun_out/r06_*
tail -3 gpurun_out/r06_pytest_gpu.log; tail -3 gpurun_out/r06_soak_determinism.txt
