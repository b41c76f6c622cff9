#!/bin/bash
# quick A/B of library tunables on the C3 bench (value = image-pairs/s); usage: bench_matrix.sh "ENV1=.. ENV2=.." "..."
for cfg in "$@"; do
  out=$(env $cfg MI355_BENCH_NO_STANDALONE=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$cfg => $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), d["ms_per_step"])')"
done
