cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_slam_usage.py -x -q -m gpu -k mini 2>&1 | grep -E "Error|error|^E " | head -20
rm -rf /tmp/prof_tl
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_tl -o run -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-also > /tmp/prof_tl.log 2>&1)
db=$(ls /tmp/prof_tl/*/*.db /tmp/prof_tl/*.db 2>/dev/null | head -1)
python tools/timeline.py $db -3 > gpurun_out/r02c_timeline.txt 2>&1
tail -12 gpurun_out/r02c_timeline.txt
