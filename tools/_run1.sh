set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fused_ops.py tests/test_gpu_determinism.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r06a_tests.log
cat gpurun_out/r06a_tests.log
for r in 1 2; do
for cfg in "t63 16" "t63 8"; do
  echo "derive $cfg: $(timeout 300 python tools/dynamics_step_profile.py $cfg 2>&1 | tail -1)" | tee -a gpurun_out/r06a_ab.txt
  echo "noderive $cfg: $(SPDY_T63_NODERIVE=1 timeout 300 python tools/dynamics_step_profile.py $cfg 2>&1 | tail -1)" | tee -a gpurun_out/r06a_ab.txt
done
done
timeout 900 bash tools/profile_step.sh r06a_step 2>&1 | tail -40
