#!/bin/bash
# collects round 6's evidence on the GPU box into gpurun_out/r06 (copied to profiles/ afterwards)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O; rm -f $O/bench.err
# counters first: bench.py reads the traffic of the dominant kernel from profiles/ (the copy on this box is refreshed here)
for C in FETCH_SIZE WRITE_SIZE; do
  MI355_BENCH_NO_STANDALONE=1 MI355_BENCH_NOPROF=1 rocprofv3 --pmc $C --kernel-include-regex blur16_stream --output-format csv -d $O/pmc_$C -o p -- python bench.py --no-cpu-baseline --no-host-frames --steps 1 --warmup 0 > /dev/null 2>> $O/bench.err
  F=$(find $O/pmc_$C -name "*counter_collection.csv" | head -1)
  cp $F $O/r06_pmc_${C}_blur16_stream.csv; gzip -f $O/r06_pmc_${C}_blur16_stream.csv
  rm -rf $O/pmc_$C
done
python profiles/pmc_traffic.py <(zcat $O/r06_pmc_FETCH_SIZE_blur16_stream.csv.gz) <(zcat $O/r06_pmc_WRITE_SIZE_blur16_stream.csv.gz) blur16_stream $O/r06_pmc_blur16_stream.json frames=500 frame=4000x3000 batch=32 "command=rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (two passes) --kernel-include-regex blur16_stream --output-format csv -- python bench.py --no-cpu-baseline --no-host-frames --steps 1 --warmup 0, with MI355_BENCH_NO_STANDALONE=1 MI355_BENCH_NOPROF=1"
cp $O/r06_pmc_blur16_stream.json profiles/r06_pmc_blur16_stream.json
# C3: the driver's parameters, and the same command under the tracer
python bench.py --steps 20 --warmup 5 > $O/r06_bench_n1.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-frames > $O/r06_bench_n1_under_rocprofv3.json 2>> $O/bench.err
DB=$(find $O/trace -name "*.db" | head -1)
python profiles/rocpd_summary.py $DB $O/r06_rocprofv3_kernel_stats_bench_n1.txt
rm -rf $O/trace
# the serial pass alone under the tracer: exclusive kernel durations recomputable from a committed file
rocprofv3 --kernel-trace --stats -d $O/trace2 -o t -- python scratch/sift_time.py 96 4000 3000 32 serial > $O/r06_sift_time_serial.txt 2>> $O/bench.err
DB=$(find $O/trace2 -name "*.db" | head -1)
python profiles/rocpd_summary.py $DB $O/r06_rocprofv3_kernel_stats_serial_pass.txt
rm -rf $O/trace2
# C4 / C5 on one GPU: the line's roofline is ransac_kernel's; C4 once more under the tracer (the same command)
MI355_BENCH_NO_STANDALONE=1 python bench.py --window 182 --steps 2 --warmup 1 > $O/r06_bench_c4_n1.json 2>> $O/bench.err
MI355_BENCH_NO_STANDALONE=1 rocprofv3 --kernel-trace --stats -d $O/trace3 -o t -- python bench.py --window 182 --steps 2 --warmup 1 --no-cpu-baseline --no-host-frames > $O/r06_bench_c4_n1_under_rocprofv3.json 2>> $O/bench.err
DB=$(find $O/trace3 -name "*.db" | head -1)
python profiles/rocpd_summary.py $DB $O/r06_rocprofv3_kernel_stats_bench_c4.txt
rm -rf $O/trace3
MI355_BENCH_NO_STANDALONE=1 python bench.py --frames 2000 --layout block --blend --window 182 --steps 1 --warmup 1 > $O/r06_bench_c5_blend_n1.json 2>> $O/bench.err
MI355_ALIGN_DBG=1 python scratch/align_c5_synth.py 4 2>&1 | tail -12 > $O/r06_align_stages_c5.txt      # the C5-sized alignment on the box's host alone, stage by stage
# a rank's share of an 8-rank run (owner-only frames in blocks, exact stripe covers, moments everywhere + records to rank 0) and the round-5 forms for comparison
python bench.py --as-rank 0,3,7 --of 8 --steps 8 --warmup 2 > $O/r06_rank_share_proxy_c3.json 2>> $O/bench.err
python bench.py --as-rank 0,3,7 --of 8 --window 182 --steps 5 --warmup 1 > $O/r06_rank_share_proxy_c4.json 2>> $O/bench.err
MI355_BENCH_NO_STANDALONE=1 python bench.py --as-rank 0,3,7 --of 8 --frames 2000 --layout block --window 182 --blend --steps 2 --warmup 1 > $O/r06_rank_share_proxy_c5_blend.json 2>> $O/bench.err
MI355_BENCH_NO_STANDALONE=1 python bench.py --as-rank 0,7 --of 8 --frames 2000 --layout block --window 182 --steps 2 --warmup 1 --align-input records > $O/r06_rank_share_proxy_c5_records_everywhere.json 2>> $O/bench.err
MI355_BENCH_NO_STANDALONE=1 python bench.py --as-rank 0,3,7 --of 8 --frames 2000 --layout block --window 182 --steps 2 --warmup 1 --frame-owner mod > $O/r06_rank_share_proxy_c5_owner_mod.json 2>> $O/bench.err
python bench.py --as-rank 0,3,7 --of 8 --steps 8 --warmup 2 --frame-owner mod > $O/r06_rank_share_proxy_c3_owner_mod.json 2>> $O/bench.err
python bench.py --as-rank 0,3,7 --of 8 --steps 8 --warmup 2 --frames-resident replicas > $O/r06_rank_share_proxy_c3_replicas.json 2>> $O/bench.err
# 2 ranks on one device over gloo: the strong-scaling path end to end (owned frames, torch transport of the same records and frames)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --backend gloo --all-ranks-on-device0 --steps 2 --warmup 1 --frames 96 --no-cpu-baseline > $O/r06_bench_dryrun_2ranks_1device.json 2>> $O/bench.err
python scratch/match_time.py 500 182 > $O/r06_match_time_c4.txt 2>> $O/bench.err
# randomised parity soaks on this commit (GPU against the oracle)
( python scratch/soak_pairs.py 84 150 large; python scratch/soak_pairs.py 85 80; python scratch/soak.py 81 100; python scratch/soak.py 87 60 large; python scratch/soak_ransac.py 82 80; python scratch/soak_match.py 83 60; python scratch/soak_mosaic.py 86 60; python scratch/soak_blend.py 88 60; python scratch/soak_api.py 89 60; python scratch/soak_surf.py 90 60 ) 2>&1 | grep -iv "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | grep -i "mismatch\|cases\|soak" > $O/r06_soak_totals.txt
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" > $O/r06_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $O/r06_pytest_gpu.txt
cat $O/r06_pytest_gpu.txt $O/r06_soak_totals.txt; tail -5 $O/bench.err; ls -la $O
