mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused_ops.py -x -q -m gpu 2>&1 | tail -8
python tools/phase_trace_t63_derive.py 8 2>&1 | grep -A12 "derive = 1" | head -14
for r in 1 2; do
for cfg in "t63 16" "t63 8"; do
  echo "derive $cfg: $(timeout 300 python tools/dynamics_step_profile.py $cfg 2>&1 | tail -1)"
  echo "noderive $cfg: $(SPDY_T63_NODERIVE=1 timeout 300 python tools/dynamics_step_profile.py $cfg 2>&1 | tail -1)"
done
done
