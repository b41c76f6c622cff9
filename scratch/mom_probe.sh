cd $GRAFT_REPO_ROOT
python -m pytest tests/test_moments.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for M in records moments; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --align-input $M 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$M C3', round(d['value']), d['phase_ms'], d['host_parts_ms'], d['quality']['images_aligned'], d['config']['canvas'])"
MI355_BENCH_NO_STANDALONE=1 python bench.py --window 182 --steps 2 --warmup 1 --no-cpu-baseline --align-input $M 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$M C4', round(d['value']), d['phase_ms'], d['host_parts_ms'], d['quality']['images_aligned'], d['config']['canvas'])"
python bench.py --as-rank 0,7 --of 8 --window 182 --steps 5 --warmup 1 --align-input $M 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$M proxy C4', {r:(round(v['ms_per_step'],2), {k:round(x,2) for k,x in v['phase_ms_synchronised'].items()}) for r,v in d['share'].items()}, round(d['one_gpu_ms_per_step'],1), round(d['predicted_speedup_over_one_gpu'],2))"
done
