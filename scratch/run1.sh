#!/bin/bash
mkdir -p gpurun_out
./scratch/cumask_probe > gpurun_out/cumask.txt 2>&1
python scratch/small_batch_time.py > gpurun_out/small_batch.txt 2>&1
python scratch/pipe_time.py 96 4000 3000 32 \
  "base:" "slots4:sift_slots=4" "split:sift_split=1" "split+prio:sift_split=1,sift_prio=1" \
  "split+one_heavy:sift_split=1,sift_one_heavy=1" "split+one_heavy+prio:sift_split=1,sift_one_heavy=1,sift_prio=1" \
  "split+one_heavy+prio+slots4:sift_split=1,sift_one_heavy=1,sift_prio=1,sift_slots=4" \
  "small3:stream_waves_small=3" "small3big2:stream_waves_small=3,stream_waves_big=2" "small3big2x2:stream_waves_small=3,stream_waves_big=2,xwaves=2" \
  "big2:stream_waves_big=2" "x2:xwaves=2" \
  "split+prio+small3big2:sift_split=1,sift_prio=1,stream_waves_small=3,stream_waves_big=2" \
  "split+one_heavy+prio+small3big2:sift_split=1,sift_one_heavy=1,sift_prio=1,stream_waves_small=3,stream_waves_big=2" \
  "split+one_heavy+prio+small3big2x2:sift_split=1,sift_one_heavy=1,sift_prio=1,stream_waves_small=3,stream_waves_big=2,xwaves=2" \
  "tail4:sift_split=1,tail_cus=4" "tail4x:sift_split=1,tail_cus=4,heavy_excl=1" "tail8:sift_split=1,tail_cus=8" "tail8x:sift_split=1,tail_cus=8,heavy_excl=1" \
  "tail8x+one_heavy:sift_split=1,tail_cus=8,heavy_excl=1,sift_one_heavy=1" "tail12x+one_heavy:sift_split=1,tail_cus=12,heavy_excl=1,sift_one_heavy=1" \
  "serial_heavy:serial_heavy=1" \
  > gpurun_out/pipe_time.txt 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1
tail -15 gpurun_out/pytest_gpu.txt
cat gpurun_out/cumask.txt gpurun_out/small_batch.txt gpurun_out/pipe_time.txt
