export TMPDIR=/tmp
R=$PWD
for v in 0 1; do
  echo "== CLSLAM_SK_B_FASTEST=$v"
  CLSLAM_SK_B_FASTEST=$v python bench.py --no-cpu-baseline --no-also 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['roofline']['all_conv_launches'])"
  CLSLAM_SK_B_FASTEST=$v BENCH_WGRAD=0 BENCH_LAYERS=2,3,4,5,8,9,12 python tools/bench_conv.py 5 30,31,32,33 2>/dev/null | grep -v amdgpu | cut -c1-200
  rm -rf /tmp/pmc_$v
  (cd /tmp && CLSLAM_SK_B_FASTEST=$v CLSLAM_SIDE_STREAM=0 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_$v -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > /tmp/pmc_$v.log 2>&1)
  db=$(ls /tmp/pmc_$v/*/*.db /tmp/pmc_$v/*.db 2>/dev/null | head -1)
  python tools/pmc_summary.py $db conv3x3_sk 2>&1 | tail -12
done
