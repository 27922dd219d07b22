export TMPDIR=/tmp
run() { python bench.py --no-cpu-baseline --no-also --steps 50 --blocks 11 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'])"; }
{ echo default; run; echo "DS_AUX=1"; CLSLAM_DS_AUX=1 run; echo default; run; echo "DS_AUX=1"; CLSLAM_DS_AUX=1 run; } > gpurun_out/c15.txt 2>&1
echo done
