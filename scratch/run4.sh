#!/bin/bash
python -m pytest tests/test_gpu_sift.py tests/test_sift_reference_run.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python scratch/sift_time.py 96 4000 3000 32 2>&1 | grep -v amdgpu
python scratch/pipe_time.py 96 4000 3000 32 "base:" "small3big2:stream_waves_small=3,stream_waves_big=2" "small3:stream_waves_small=3" "base:" "small3big2:stream_waves_small=3,stream_waves_big=2" 2>&1 | grep -v amdgpu
