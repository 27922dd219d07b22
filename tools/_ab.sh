export TMPDIR=/tmp
run() { CLSLAM_TOOL_LIB=$1 python tools/bench_variant.py --no-cpu-baseline --no-also --steps 50 --blocks 11 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'])"; }
{
timeout 600 python -m pytest tests/test_conv.py -q -m gpu -k winograd 2>&1 | tail -1
for v in act ""; do for g in 96 160; do b=$([ $g = 96 ] && echo 5 || echo 10); echo "== variant '$v' groups $g B=$b"; CLSLAM_TOOL_LIB=$v CLSLAM_WINO_GROUPS=$g BENCH_WGRAD=0 BENCH_LAYERS=0,1,2,3 python tools/bench_conv.py $b 40 2>&1 | grep -v "amdgpu\|tool library"; done; done
for i in 1 2 3; do echo "step act:"; run act; echo "step i32:"; run ""; done
} > gpurun_out/ab_i32.txt 2>&1
echo done
