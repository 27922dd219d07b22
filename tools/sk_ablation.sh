#!/bin/bash
# Where the stream-K conv kernel's time goes (profiles/r02_sk_ablation.txt): the kernel's probes
# (CLSLAM_SK_DBG: 1 = no epilogue, 2 = no hand-off, 4 = no MFMA, 8 = no DMA) on the layer shapes of the step, B=5,
# next to the tiled kernel (c-1).  Run on an MI355X from the repo root:  bash tools/sk_ablation.sh
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for dbg in 0 1 3 7 11; do
  echo "== CLSLAM_SK_DBG=$dbg"
  CLSLAM_SK_DBG=$dbg BENCH_WGRAD=0 BENCH_LAYERS=0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,18 timeout 300 python tools/bench_conv.py 5 30,31,32,33 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/sk_ablation.txt
