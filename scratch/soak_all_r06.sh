#!/bin/bash
# long randomised parity soaks (GPU vs oracle) on round 6's last code commit
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06s; mkdir -p $O
( python scratch/soak_pairs.py 94 600 large; python scratch/soak_pairs.py 95 300; python scratch/soak_blend.py 98 300; python scratch/soak.py 91 420; python scratch/soak.py 97 240 large; python scratch/soak_ransac.py 92 300; python scratch/soak_match.py 93 180; python scratch/soak_mosaic.py 96 240; python scratch/soak_api.py 99 180; python scratch/soak_surf.py 100 240 ) 2>&1 | grep -i "mismatch\|cases\|soak" | grep -v "^+" > $O/r06_soak_totals_long.txt
cat $O/r06_soak_totals_long.txt
