#!/bin/bash
# kernel-time breakdown of mi355_mosaic_blended_dev at C5 size: 2000 resident 12 MP frames on a ~20000^2 canvas (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_blend_c5; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python scratch/blend_dev_time.py 2000 50 > $O/log.txt 2>&1
DB=$(find $O/trace -name "*.db" | head -1)
python profiles/rocpd_summary.py $DB $O/blend_kernel_stats_c5.txt
rm -rf $O/trace
grep "^blend\|^sha" $O/log.txt
head -30 $O/blend_kernel_stats_c5.txt | cut -c1-200
