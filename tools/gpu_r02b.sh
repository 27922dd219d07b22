cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv.py -x -q -m gpu -k stream_k 2>&1 | tail -3
for dbg in 0 1 3; do
  echo "== CLSLAM_SK_DBG=$dbg"
  CLSLAM_SK_DBG=$dbg BENCH_WGRAD=0 BENCH_LAYERS=0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,18 timeout 300 python tools/bench_conv.py 5 30,31,32,33 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r02b_sk_ablation.txt
