#!/bin/bash
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -c processor /proc/cpuinfo; python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"
for t in 1 2 4 8 16 32; do MI355_ALIGN_DBG=1 MI355_HOST_THREADS=$t python scratch/align_c4.py 500 25 2>&1 | grep -E "cholesky|moments|threads" | tail -3; done
python -m pytest tests/test_sift_reference_run.py -m gpu -x -q 2>&1 | tail -2
