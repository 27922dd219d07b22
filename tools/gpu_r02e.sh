cd $GRAFT_REPO_ROOT
brief() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['all_conv_launches'])"; }
for v in 0 512 256 128 64; do
  echo "== CLSLAM_SK_ALL=$v"; CLSLAM_SK_ALL=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | brief
done
echo "== no streamk"; CLSLAM_NO_STREAMK=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | brief
