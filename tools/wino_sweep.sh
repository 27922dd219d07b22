#!/bin/bash
# Winograd kernel against batch size AND workgroup count (run on the GPU box): is a launch CU-bound or bound by something chip-wide?
#   bash tools/wino_sweep.sh > gpurun_out/r06_wino_sweep.txt
for g in 96 128 160 256; do
  for b in 5 10 15 33; do
    echo "== groups $g  B = $b"
    CLSLAM_WINO_GROUPS=$g BENCH_WGRAD=0 BENCH_LAYERS=0,1,2,3 python tools/bench_conv.py $b 40 2>&1 | grep -v amdgpu
  done
done
