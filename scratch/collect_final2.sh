#!/bin/bash
# the lines that contain the host alignment, again on the round's last commit (records and moments), into gpurun_out/r05
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
MI355_BENCH_NO_STANDALONE=1 python bench.py --window 182 --steps 2 --warmup 1 > $O/r05_bench_c4_n1.json 2>> $O/bench.err
MI355_BENCH_NO_STANDALONE=1 python bench.py --frames 2000 --layout block --blend --window 182 --steps 1 --warmup 1 > $O/r05_bench_c5_blend_n1.json 2>> $O/bench.err
python bench.py --as-rank 0,3,7 --of 8 --steps 8 --warmup 2 > $O/r05_rank_share_proxy_c3.json 2>> $O/bench.err
python bench.py --as-rank 0,7 --of 8 --window 182 --steps 5 --warmup 1 > $O/r05_rank_share_proxy_c4.json 2>> $O/bench.err
MI355_BENCH_NO_STANDALONE=1 python bench.py --as-rank 0,3,7 --of 8 --frames 2000 --layout block --window 182 --blend --steps 1 --warmup 1 > $O/r05_rank_share_proxy_c5_blend.json 2>> $O/bench.err
python bench.py --as-rank 0,3,7 --of 8 --steps 8 --warmup 2 --align-input moments > $O/r05_rank_share_proxy_c3_moments.json 2>> $O/bench.err
python bench.py --as-rank 0,7 --of 8 --window 182 --steps 5 --warmup 1 --align-input moments > $O/r05_rank_share_proxy_c4_moments.json 2>> $O/bench.err
MI355_BENCH_NO_STANDALONE=1 python bench.py --as-rank 0,3,7 --of 8 --frames 2000 --layout block --window 182 --blend --steps 1 --warmup 1 --align-input moments > $O/r05_rank_share_proxy_c5_blend_moments.json 2>> $O/bench.err
MI355_BENCH_NO_STANDALONE=1 python bench.py --frames 2000 --layout block --blend --window 182 --steps 1 --warmup 1 --align-input moments > $O/r05_bench_c5_blend_n1_moments.json 2>> $O/bench.err
MI355_BENCH_NO_STANDALONE=1 python bench.py --window 182 --steps 2 --warmup 1 --align-input moments > $O/r05_bench_c4_n1_moments.json 2>> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/r05_bench_n1_driver_style.json 2>> $O/bench.err
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" > $O/r05_pytest_gpu_final.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1; cat $O/r05_pytest_gpu_final.txt
