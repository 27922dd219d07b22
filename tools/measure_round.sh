#!/bin/bash
# Reproduces the round's committed measurements on one MI355X (run from the repo root on the GPU box):
#   bash tools/measure_round.sh <tag>        -> gpurun_out/<tag>_*
# bench line, rocprofv3 kernel summaries (three-stream and serial launch order), FETCH_SIZE / WRITE_SIZE
# PMC passes (separate runs, no other tracing), the other BASELINE configurations, and the micro-benchmarks.
set -u
TAG=${1:-rXX}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
# identity of the kernel sources every file of this set was measured on (bench.py prints the same id and refuses to pair
# its timings with counter files of another build)
python -c "import sys; sys.path.insert(0, 'cl-slam_amd'); from clslam_hip import _lib; print(_lib.build_id())" 2>/dev/null | tail -1 > $OUT/${TAG}_build_id.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${TAG}_smoke.txt 2>&1
# counter passes FIRST: bench.py below then finds a counter file of ITS build and reports roofline.traffic from it
for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$ctr
    (cd /tmp && CLSLAM_SIDE_STREAM=0 rocprofv3 --pmc $ctr -d /tmp/pmc_$ctr -o run -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > /tmp/pmc_$ctr.log 2>&1)
    db=$(ls /tmp/pmc_$ctr/*/*.db /tmp/pmc_$ctr/*.db 2>/dev/null | head -1)
    python tools/pmc_summary.py $db conv > $OUT/${TAG}_pmc_$ctr.txt 2>&1
    eval "DB_$ctr=$db"
done
# MFMA-busy evidence (north_star: "rocprof ... MFMA-busy against gfx950 peak"): two more counter passes of the same build
rm -rf /tmp/pmc_sqA /tmp/pmc_sqB
(cd /tmp && CLSLAM_SIDE_STREAM=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/pmc_sqA -o run -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > /tmp/pmc_sqA.log 2>&1)
(cd /tmp && CLSLAM_SIDE_STREAM=0 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d /tmp/pmc_sqB -o run -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > /tmp/pmc_sqB.log 2>&1)
python tools/pmc_mfma_busy.py $(ls /tmp/pmc_sqA/*/*.db /tmp/pmc_sqA/*.db 2>/dev/null | head -1) $(ls /tmp/pmc_sqB/*/*.db /tmp/pmc_sqB/*.db 2>/dev/null | head -1) conv > $OUT/${TAG}_pmc_mfma_busy.txt 2>&1
# L2-miss traffic per launch of every conv kernel, keyed by kernel name: bench.py's roofline.traffic reads THIS file
python tools/pmc_traffic.py $DB_FETCH_SIZE $DB_WRITE_SIZE $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.log 2>&1
cp $OUT/${TAG}_pmc_traffic.json profiles/pmc_traffic.json
python bench.py --dump-convs > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
grep "^# conv" $OUT/${TAG}_bench.err > $OUT/${TAG}_conv_launches.txt
prof() {   # name, env..., then bench args after --
    local name=$1; shift
    rm -rf /tmp/prof_$name
    (cd /tmp && env "$@" rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o run -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > /tmp/prof_$name.log 2>&1)
    local db=$(ls /tmp/prof_$name/*/*.db /tmp/prof_$name/*.db 2>/dev/null | head -1)
    python tools/rocprof_summary.py $db $OUT/${TAG}_kernel_stats_$name.csv
    if [ "$name" == "3streams" ]; then python tools/timeline.py $db -5 > $OUT/${TAG}_timeline_one_step.txt 2>&1; fi
}
prof 3streams CLSLAM_SIDE_STREAM=1
prof serial CLSLAM_SIDE_STREAM=0
# the RCCL code path on the one GPU (backend nccl, world size 1): which queue the all-reduce kernels run on, what they overlap
rm -rf /tmp/prof_rccl1
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_rccl1 -o run -- python $OLDPWD/tools/rccl1_step.py 6 > /tmp/prof_rccl1.log 2>&1)
{ tail -1 /tmp/prof_rccl1.log; python tools/timeline.py $(ls /tmp/prof_rccl1/*/*.db /tmp/prof_rccl1/*.db 2>/dev/null | head -1) -2; } > $OUT/${TAG}_rccl1_timeline.txt 2>&1
brief() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('value', 'unit', 'ms_per_step', 'also')}, d['config']['workload'][:70])"; }
{
  for r in 0 2 32; do echo "== 192x640 replay $r"; python bench.py --replay $r --steps 20 --warmup 5 --no-cpu-baseline | brief; done
  echo "== 192x640 replay 4, uniform-random images"; python bench.py --random-images --steps 20 --warmup 5 --no-cpu-baseline --no-also | brief
  echo "== 384x1280 replay 8"; python bench.py --height 384 --width 1280 --replay 8 --steps 10 --warmup 3 --no-cpu-baseline | brief
  echo "== 384x1280 replay 8 + loop-closure encoder forward in every frame (BASELINE config 5 on one GPU, ONE number)"; python bench.py --height 384 --width 1280 --replay 8 --lcd --steps 10 --warmup 3 --no-cpu-baseline | brief
  echo "== 192x640 replay 4, opt-in host-output path"; python bench.py --host-outputs --steps 50 --warmup 5 --no-cpu-baseline --no-also | brief
} > $OUT/${TAG}_other_configs.txt 2>&1
BENCH_WGRAD=1 python tools/bench_conv.py 5 30,31,32,33 > $OUT/${TAG}_conv_microbench.txt 2>&1
BENCH_WGRAD=0 python tools/bench_conv.py 1 30,31,32,33 > $OUT/${TAG}_conv_microbench_b1.txt 2>&1
# Winograd F(2x2,3x3) (config 40) against the library's direct pick on the encoder layer shapes, B = 5 / 10 / 33
{ for b in 5 10 33; do echo "== B = $b"; BENCH_WGRAD=0 BENCH_LAYERS=0,1,2,3,8 python tools/bench_conv.py $b 40 2>&1 | grep -v amdgpu; done; } > $OUT/${TAG}_wino_microbench.txt
# round 6: the END-TO-END frame (pinned host minibatch -> H2D inside adapt() -> pose + losses on the host) as a timeline with its copies
rm -rf /tmp/prof_e2e
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_e2e -o run -- python $OLDPWD/tools/e2e_frames.py 12 4 > /tmp/prof_e2e.log 2>&1)
python tools/timeline_e2e.py $(ls /tmp/prof_e2e/*/*.db /tmp/prof_e2e/*.db 2>/dev/null | head -1) -3 > $OUT/${TAG}_timeline_e2e.txt 2>&1
python tools/e2e_frames.py 12 4 2>&1 | grep -v amdgpu > $OUT/${TAG}_e2e_host_stamps.txt
# round 6: the Winograd kernel against batch size and workgroup count, and its per-wave phase times (probe build lib/variants/..._wtrace.so,
# tools/build_variant.py wtrace conv_wino -DCLSLAM_WINO_TRACE=3 -- built in the container, travels with the snapshot)
bash tools/wino_sweep.sh > $OUT/${TAG}_wino_sweep.txt 2>&1
if [ -f cl-slam_amd/lib/variants/libclslam_hip_wtrace.so ]; then
  { for args in "10 48 160 64 160" "5 48 160 64 96" "10 24 80 128 160" "10 6 20 512 160"; do CLSLAM_TOOL_LIB=wtrace python tools/wino_trace_waves.py $args 2>&1 | grep -v amdgpu; done; } > $OUT/${TAG}_wino_waves.txt
fi
# round 6: FETCH_SIZE / WRITE_SIZE against known byte counts in the Winograd kernel's access patterns
bash tools/pmc_calib.sh > $OUT/${TAG}_pmc_calibration.txt 2>&1
# gfx950 calibration the kernel designs rest on: MFMA vs same-wave / partner-wave VALU, LDS-DMA rate per CU
{ for m in mfma_clock mfma_valu_share lds_dma_rate event_gap ext_launch_cost; do echo "== tools/micro/$m.hip"; hipcc --offload-arch=gfx950 -O3 -o /tmp/$m tools/micro/$m.hip 2>/dev/null && /tmp/$m; done; } > $OUT/${TAG}_micro_calibration.txt 2>&1
# one decoder-only step (steps 2..5 of adapt(steps=5)) as a kernel timeline
rm -rf /tmp/prof_s5
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_s5 -o run -- python $OLDPWD/bench.py --adapt-steps 5 --steps 6 --warmup 2 --blocks 1 --no-also --no-cpu-baseline > /tmp/prof_s5.log 2>&1)
python tools/timeline.py $(ls /tmp/prof_s5/*/*.db /tmp/prof_s5/*.db 2>/dev/null | head -1) -14 > $OUT/${TAG}_timeline_reuse_step.txt 2>&1
python tools/bench_small.py > $OUT/${TAG}_small_kernels.txt 2>&1
python tools/bench_reduce.py > $OUT/${TAG}_reduce.txt 2>&1
BENCH_DGRAD=1 BENCH_WGRAD=0 python tools/bench_conv.py 5 20,21,22,26,30,31,32,33 2>&1 | grep -v amdgpu > $OUT/${TAG}_conv_microbench_dgrad.txt
python tools/bench_gpu_bound.py 2>&1 | grep 'K=' > $OUT/${TAG}_gpu_bound.txt
python tools/bench_memo.py 2>&1 | grep -v amdgpu > $OUT/${TAG}_descriptor_memo.txt
bash tools/dp2_gloo.sh > $OUT/${TAG}_dp2_gloo.txt 2>&1
{ python tools/soak.py 2000 4; python tools/soak.py 2000 0; } 2>&1 | grep -v amdgpu > $OUT/${TAG}_soak.txt
# the REAL frame of the SLAM loop: descriptor pass + replay get (reference's host path / GPU ingest) + adapt + read-back
python tools/real_frame.py 40 4 2>&1 | grep -v amdgpu > $OUT/${TAG}_real_frame.txt
# the measured numbers the parity tests print (trajectory envelope through steps=5, backward ladder, configs 4 / 5 at full workload)
python -m pytest tests/test_trajectory.py -q -m gpu -s 2>&1 | grep "^\[hip\|passed\|failed" > $OUT/${TAG}_trajectory.txt
python -m pytest tests/test_teacher_forced_steps.py -q -m gpu -s 2>&1 | grep "^\[hip\|passed\|failed" > $OUT/${TAG}_teacher_forced.txt
python tools/diag_flips.py 2>&1 | grep -v amdgpu > $OUT/${TAG}_flip_attribution.txt
python -m pytest tests/test_backward_parity.py -q -m gpu -s 2>&1 | grep "hip \|    \|passed\|failed" > $OUT/${TAG}_backward_parity.txt
python -m pytest tests/test_configs_4_5.py -q -s 2>&1 | grep "^\[config\|passed\|failed" > $OUT/${TAG}_configs_4_5.txt
echo done
