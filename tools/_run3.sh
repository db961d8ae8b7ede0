timeout 1500 python -m pytest tests/test_gpu_sharded_step.py tests/test_gpu_sharding.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
