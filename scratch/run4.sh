#!/bin/bash
for q in "" 2 4 8 16; do
  echo "GPU_MAX_HW_QUEUES=$q"; if [ -z "$q" ]; then python scratch/pipe_time.py 96 4000 3000 32 "base:" "slots4:sift_slots=4" 2>&1 | grep "us/frame" | cut -c1-100; else GPU_MAX_HW_QUEUES=$q python scratch/pipe_time.py 96 4000 3000 32 "base:" "slots4:sift_slots=4" 2>&1 | grep "us/frame" | cut -c1-100; fi
done
