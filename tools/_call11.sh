set -u
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
bash tools/pmc_calib.sh > $O/c11_pmc_calibration.txt 2>&1
rm -rf /tmp/prof_3s
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_3s -o run -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > /tmp/prof_3s.log 2>&1)
python tools/timeline.py $(ls /tmp/prof_3s/*/*.db /tmp/prof_3s/*.db 2>/dev/null | head -1) -5 > $O/c11_timeline.txt 2>&1
echo done
