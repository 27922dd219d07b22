set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv.py tests/test_replay_lcd.py tests/test_edge_cases.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r02a_tests.log
cat gpurun_out/r02a_tests.log
BENCH_WGRAD=0 timeout 600 python tools/bench_conv.py 5 30,31,32,33 2>&1 | tee gpurun_out/r02a_conv_sk.txt
BENCH_WGRAD=0 timeout 300 python tools/bench_conv.py 1 30,31,32,33 2>&1 | tee gpurun_out/r02a_conv_sk_b1.txt
