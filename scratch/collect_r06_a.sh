#!/bin/bash
# round 6, first measurement pass: the three bench lines and the corrected rank-share proxies into gpurun_out/r06
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
python bench.py --steps 10 --warmup 3 > $O/r06_bench_n1.json 2>> $O/bench.err
MI355_BENCH_NO_STANDALONE=1 python bench.py --window 182 --steps 2 --warmup 1 > $O/r06_bench_c4_n1.json 2>> $O/bench.err
MI355_BENCH_NO_STANDALONE=1 python bench.py --frames 2000 --layout block --blend --window 182 --steps 1 --warmup 1 > $O/r06_bench_c5_blend_n1.json 2>> $O/bench.err
python bench.py --as-rank 0,3,7 --of 8 --steps 8 --warmup 2 > $O/r06_rank_share_proxy_c3.json 2>> $O/bench.err
python bench.py --as-rank 0,7 --of 8 --window 182 --steps 5 --warmup 1 > $O/r06_rank_share_proxy_c4.json 2>> $O/bench.err
MI355_BENCH_NO_STANDALONE=1 python bench.py --as-rank 0,3,7 --of 8 --frames 2000 --layout block --window 182 --blend --steps 2 --warmup 1 > $O/r06_rank_share_proxy_c5_blend.json 2>> $O/bench.err
MI355_BENCH_NO_STANDALONE=1 python bench.py --as-rank 0,7 --of 8 --frames 2000 --layout block --window 182 --steps 2 --warmup 1 --align-input records > $O/r06_rank_share_proxy_c5_records_everywhere.json 2>> $O/bench.err
tail -5 $O/bench.err
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[1], "unreadable", e); sys.exit()
if "share" in d:
    print(sys.argv[1].split("/")[-1], "one", round(d["one_gpu_ms_per_step"],1), "pred", round(d["predicted_ms_per_step"],1), "x", round(d["predicted_speedup_over_one_gpu"],2), {k:round(v["ms_per_step"],1) for k,v in d["share"].items()})
else:
    print(sys.argv[1].split("/")[-1], round(d["value"],1), round(d["ms_per_step"],1), d["roofline"]["kernel"][:20], round(d["roofline"]["frac"],3), d.get("parity_sample"), d.get("host_frames",{}).get("value"))
PY
done
