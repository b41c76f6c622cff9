#!/bin/bash
# collects the round's evidence on the GPU box into gpurun_out/r05 (copied to profiles/ afterwards)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
# counters first: bench.py reads the traffic of the dominant kernel from profiles/ (the copy on this box is refreshed here)
for C in FETCH_SIZE WRITE_SIZE; do
  MI355_BENCH_NO_STANDALONE=1 MI355_BENCH_NOPROF=1 rocprofv3 --pmc $C --kernel-include-regex blur16_stream --output-format csv -d $O/pmc_$C -o p -- python bench.py --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2>> $O/bench.err
  F=$(find $O/pmc_$C -name "*counter_collection.csv" | head -1)
  cp $F $O/r05_pmc_${C}_blur16_stream.csv; gzip -f $O/r05_pmc_${C}_blur16_stream.csv
  rm -rf $O/pmc_$C
done
python profiles/pmc_traffic.py <(zcat $O/r05_pmc_FETCH_SIZE_blur16_stream.csv.gz) <(zcat $O/r05_pmc_WRITE_SIZE_blur16_stream.csv.gz) blur16_stream $O/r05_pmc_blur16_stream.json frames=500 frame=4000x3000 batch=32 "command=rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (two passes) --kernel-include-regex blur16_stream --output-format csv -- python bench.py --no-cpu-baseline --steps 1 --warmup 0, with MI355_BENCH_NO_STANDALONE=1 MI355_BENCH_NOPROF=1"
cp $O/r05_pmc_blur16_stream.json profiles/r05_pmc_blur16_stream.json
python bench.py --steps 5 --warmup 2 > $O/r05_bench_n1.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/r05_bench_n1_driver_style.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/r05_bench_n1_under_rocprofv3.json 2>> $O/bench.err
DB=$(find $O/trace -name "*.db" | head -1)
python profiles/rocpd_summary.py $DB $O/r05_rocprofv3_kernel_stats_bench_n1.txt
python profiles/rocpd_by_grid.py $DB > $O/r05_rocprofv3_kernel_stats_by_grid_bench_n1.txt 2>/dev/null
python profiles/rocpd_overlap.py $DB > $O/r05_rocpd_overlap.txt 2>/dev/null
rm -rf $O/trace
# the serial_heavy pass alone under the tracer: exclusive kernel durations recomputable from a committed file
rocprofv3 --kernel-trace --stats -d $O/trace2 -o t -- python scratch/sift_time.py 96 4000 3000 32 serial > $O/r05_sift_time_serial.txt 2>> $O/bench.err
DB=$(find $O/trace2 -name "*.db" | head -1)
python profiles/rocpd_summary.py $DB $O/r05_rocprofv3_kernel_stats_serial_pass.txt
python profiles/rocpd_by_grid.py $DB > $O/r05_rocprofv3_kernel_stats_by_grid_serial_pass.txt 2>/dev/null
rm -rf $O/trace2
# C4 on one GPU (74 029 window pairs) and the 2-rank dry run of the strong-scaling path on one device (gloo, torch transport)
MI355_BENCH_NO_STANDALONE=1 python bench.py --window 182 --steps 2 --warmup 1 > $O/r05_bench_c4_n1.json 2>> $O/bench.err
# C5: 2000 frames piled onto a 20000^2-class canvas, window 182, warp + LaplacianPyramidBlending with everything co-resident
MI355_BENCH_NO_STANDALONE=1 python bench.py --frames 2000 --layout block --blend --window 182 --steps 1 --warmup 1 > $O/r05_bench_c5_blend_n1.json 2>> $O/bench.err
# C4 under the tracer: bf_match_kernel / ransac_kernel / select_kernel durations recomputable from a committed file
MI355_BENCH_NO_STANDALONE=1 rocprofv3 --kernel-trace --stats -d $O/trace3 -o t -- python bench.py --window 182 --steps 1 --warmup 1 --no-cpu-baseline > $O/r05_bench_c4_n1_under_rocprofv3.json 2>> $O/bench.err
DB=$(find $O/trace3 -name "*.db" | head -1)
python profiles/rocpd_summary.py $DB $O/r05_rocprofv3_kernel_stats_bench_c4.txt
rm -rf $O/trace3
# the pair stage alone at C4 size (74 029 pairs x 3 calls) and the pipe / issue-rate micro-benchmarks its comments quote
python scratch/match_time.py 500 182 > $O/r05_match_time_c4.txt 2>> $O/bench.err
MI355_RANSAC_DBG=1 python scratch/ransac_time.py 300 > $O/r05_ransac_time.txt 2>&1
python bench.py --as-rank 0,3,7 --of 8 --steps 8 --warmup 2 > $O/r05_rank_share_proxy_c3.json 2>> $O/bench.err
python bench.py --as-rank 0,7 --of 8 --window 182 --steps 5 --warmup 1 > $O/r05_rank_share_proxy_c4.json 2>> $O/bench.err
# the default compositing path shared by stripes (C5 size): the whole blended canvas on one GPU against a rank's stripe
MI355_BENCH_NO_STANDALONE=1 python bench.py --as-rank 0,3,7 --of 8 --frames 2000 --layout block --window 182 --blend --steps 1 --warmup 1 > $O/r05_rank_share_proxy_c5_blend.json 2>> $O/bench.err
python scratch/small_batch_time.py 2>&1 | grep "^ransac_split" > $O/r05_small_batch_time.txt
python scratch/sift_time.py 96 4000 3000 32 > $O/r05_sift_time.txt 2>> $O/bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --backend gloo --all-ranks-on-device0 --steps 2 --warmup 1 --frames 96 --no-cpu-baseline > $O/r05_bench_dryrun_2ranks_1device.json 2>> $O/bench.err
# the blend's kernel breakdown (200 resident 12 MP chips) and the counters behind DESIGN's RANSAC paragraph
bash scratch/prof_blend.sh 200 > /dev/null 2>&1; cp gpurun_out/prof_blend/blend_kernel_stats.txt $O/r05_blend_kernel_stats.txt; grep "^blend" gpurun_out/prof_blend/log.txt >> $O/r05_blend_kernel_stats.txt
bash scratch/pmc_ransac.sh 2>/dev/null | grep "^p[123] " > $O/r05_pmc_ransac.txt
# randomised parity soaks on this commit (GPU against the oracle): totals quoted in DESIGN.md
( python scratch/soak_blend.py 58 60; python scratch/soak.py 51 100; python scratch/soak.py 57 60 large; python scratch/soak_ransac.py 52 100; python scratch/soak_match.py 53 60; python scratch/soak_pairs.py 54 80; python scratch/soak_mosaic.py 55 60; python scratch/soak_api.py 56 60; python scratch/soak_surf.py 59 60 ) 2>&1 | grep -iv "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | grep -i "mismatch\|cases\|soak" > $O/r05_soak_totals.txt
tail -3 $O/bench.err
ls -la $O

python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" > $O/r05_pytest_gpu.txt; cat $O/r05_pytest_gpu.txt
