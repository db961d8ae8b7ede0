timeout 1500 python -m pytest tests/test_gpu_sharded_step.py -x -q -m gpu -k "transposed or in_process_ranks" 2>&1 | tail -30
