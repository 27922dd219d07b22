cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r02d_gpu_tests.log; cat gpurun_out/r02d_gpu_tests.log
brief() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('value', 'ms_per_step', 'also')}, d['roofline']['all_conv_launches'])"; }
{
echo "== default";           python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | brief
echo "== --readback";        python bench.py --steps 30 --warmup 5 --no-cpu-baseline --readback 2>&1 | brief
echo "== replay 0";       python bench.py --replay 0 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | brief
echo "== replay 0 no streamk";    CLSLAM_NO_STREAMK=1 python bench.py --replay 0 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | brief
} 2>&1 | tee gpurun_out/r02d_bench_variants.txt
