cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dist.py tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -2
for R in 7,0 0,7 3,7,0; do python bench.py --as-rank $R --of 8 --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$R', {r:(round(v['ms_per_step'],2), round(sum(v['phase_ms_synchronised'].values()),2)) for r,v in d['share'].items()}, round(d['one_gpu_ms_per_step'],1), round(d['predicted_speedup_over_one_gpu'],2))"; done
MI355_HOST_THREADS=1 python bench.py --as-rank 0,7 --of 8 --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('threads1 0,7', {r:(round(v['ms_per_step'],2), round(sum(v['phase_ms_synchronised'].values()),2)) for r,v in d['share'].items()}, round(d['one_gpu_ms_per_step'],1), round(d['predicted_speedup_over_one_gpu'],2))"
