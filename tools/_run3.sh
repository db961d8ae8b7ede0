timeout 1500 python -m pytest tests/test_gpu_sharded_step.py -x -q -m gpu -s -k "world1 or rccl" 2>&1 | tail -30
