#!/bin/bash
for v in "" nowave inl r04ransac ""; do
  if [ -z "$v" ]; then echo "== current (wave inverse out of line)"; python scratch/ransac_time.py 300 2>&1 | grep "pairs "; 
  else echo "== variant $v"; MI355_LIB=$PWD/scratch/variants/lib_$v.so python scratch/ransac_time.py 300 2>&1 | grep "pairs "; fi
done
python scratch/small_batch_time.py 2>&1 | grep -v amdgpu | grep "^ransac_split -1\|^ransac_split  0"
