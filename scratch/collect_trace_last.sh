#!/bin/bash
# the traced lines again on the round's last code commit (extrema_stream changed after the main collection)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05t; mkdir -p $O
python bench.py --steps 5 --warmup 2 > $O/r05_bench_n1.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r05_bench_n1_under_rocprofv3.json 2>> $O/bench.err
DB=$(find $O/trace -name "*.db" | head -1)
python profiles/rocpd_summary.py $DB $O/r05_rocprofv3_kernel_stats_bench_n1.txt
python profiles/rocpd_by_grid.py $DB > $O/r05_rocprofv3_kernel_stats_by_grid_bench_n1.txt 2>/dev/null
python profiles/rocpd_overlap.py $DB > $O/r05_rocpd_overlap.txt 2>/dev/null
rm -rf $O/trace
rocprofv3 --kernel-trace --stats -d $O/trace2 -o t -- python scratch/sift_time.py 96 4000 3000 32 serial > $O/r05_sift_time_serial.txt 2>> $O/bench.err
DB=$(find $O/trace2 -name "*.db" | head -1)
python profiles/rocpd_summary.py $DB $O/r05_rocprofv3_kernel_stats_serial_pass.txt
python profiles/rocpd_by_grid.py $DB > $O/r05_rocprofv3_kernel_stats_by_grid_serial_pass.txt 2>/dev/null
rm -rf $O/trace2
python scratch/sift_time.py 96 4000 3000 32 > $O/r05_sift_time.txt 2>> $O/bench.err
ls -la $O
