#!/bin/bash
# kernel-time breakdown of mi355_mosaic_blended_dev on F resident 12 MP frames (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
F=${1:-200}
O=gpurun_out/prof_blend; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python scratch/blend_dev_time.py $F > $O/log.txt 2>&1
DB=$(find $O/trace -name "*.db" | head -1)
python profiles/rocpd_summary.py $DB $O/blend_kernel_stats.txt
rm -rf $O/trace
grep "^blend\|^sha" $O/log.txt
head -30 $O/blend_kernel_stats.txt | cut -c1-200
