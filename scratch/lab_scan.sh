#!/bin/bash
cd scratch
for u in 1024 1536 2048 2560 3072 4096; do
  echo "== LAB_UNITS=$u"
  LAB_UNITS=$u timeout 300 ./blur_lab 4000 3000 16 "v4 asm w4" | awk 'NR%3==0' | cut -c1-120
  LAB_UNITS=$u timeout 300 ./blur_lab 4000 3000 16 "v4 asm w3" | awk 'NR%3==0' | grep -E "R 10|R 13" | cut -c1-120
done
