#!/bin/bash
python -m pytest tests/test_gpu_sift.py tests/test_sift_reference_run.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python scratch/sift_time.py 96 4000 3000 32 2>&1 | grep -v amdgpu | grep -E "orient|describe|sum of|slots"
python scratch/soak.py 5 40 2>&1 | tail -1
