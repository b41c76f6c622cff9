#!/bin/bash
mkdir -p gpurun_out
python scratch/small_batch_diag.py each > gpurun_out/diag_each.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_small -- python $GRAFT_REPO_ROOT/scratch/small_batch_diag.py trace > $GRAFT_REPO_ROOT/gpurun_out/diag_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/rocpd_summary.py $(ls gpurun_out/prof_small/*/*.db | head -1) > gpurun_out/diag_trace_stats.txt 2>&1
cat gpurun_out/diag_each.txt; head -40 gpurun_out/diag_trace_stats.txt
