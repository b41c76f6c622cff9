#!/bin/bash
# long randomised parity soaks (GPU vs oracle) on the round's last commit
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s; mkdir -p $O
( python scratch/soak_blend.py 78 300; python scratch/soak.py 71 480; python scratch/soak.py 77 300 large; python scratch/soak_ransac.py 72 360; python scratch/soak_match.py 73 180; python scratch/soak_pairs.py 74 300; python scratch/soak_mosaic.py 75 180; python scratch/soak_api.py 76 180; python scratch/soak_surf.py 79 300 ) 2>&1 | grep -i "mismatch\|cases\|soak" | grep -v "^+" > $O/r05_soak_totals_last_commit.txt
cat $O/r05_soak_totals_last_commit.txt
