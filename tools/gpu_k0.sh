cd $GRAFT_REPO_ROOT
brief() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['also']['ms_per_frame'], d['also']['end_to_end']['ms_per_frame'])"; }
for rep in 1 2 3; do
  echo "== default";     python bench.py --replay 0 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | brief
  echo "== no streamk";  CLSLAM_NO_STREAMK=1 python bench.py --replay 0 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | brief
done
