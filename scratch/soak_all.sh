#!/bin/bash
# long randomised parity soaks (GPU vs oracle) on the round's last commit
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05s; mkdir -p $O
( python scratch/soak_blend.py 68 180; python scratch/soak.py 61 300; python scratch/soak.py 67 180 large; python scratch/soak_ransac.py 62 240; python scratch/soak_match.py 63 120; python scratch/soak_pairs.py 64 180; python scratch/soak_mosaic.py 65 120; python scratch/soak_api.py 66 120; python scratch/soak_surf.py 69 180 ) 2>&1 | grep -i "mismatch\|cases\|soak" | grep -v "^+" > $O/r05_soak_totals_final.txt
cat $O/r05_soak_totals_final.txt
