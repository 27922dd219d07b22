hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_clock tools/micro/mfma_clock.hip && /tmp/mfma_clock
(timeout 300 python -m pytest tests/test_conv.py -x -q -m gpu -k "winograd" 2>&1 | tail -3)
BENCH_WGRAD=0 BENCH_LAYERS=0,1,2,3,4,5,8 timeout 300 python tools/bench_conv.py 5 40 2>&1 | grep -v amdgpu.ids
BENCH_WGRAD=0 BENCH_LAYERS=0,1,2,3 timeout 300 python tools/bench_conv.py 10 40 2>&1 | grep -v amdgpu.ids
CLSLAM_HIPCC_EXTRA=-DCLSLAM_WINO_TRACE=1 python cl-slam_amd/csrc/build.py > /dev/null 2>&1
python tools/wino_trace.py 5 48 160 64 2>&1 | grep -v "amdgpu.ids\|^wg"
python tools/wino_trace.py 10 12 40 256 2>&1 | grep -v "amdgpu.ids\|^wg"
