#!/bin/bash
# Two ranks SHARING the one MI355X of the box (backend gloo on device tensors): step time of the data-parallel path with and
# without the asynchronous tail (gradient reduction -> all-reduce -> Adam on their own stream).  Not a scaling number -- both
# ranks compete for one GPU and gloo stages the 17.9 MB arena through host memory -- but the first run of that code on HIP streams.
#   bash tools/dp2_gloo.sh > profiles/rNN_dp2_gloo.txt
set -u
brief() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('value', 'unit', 'ms_per_step', 'n_gpus')}, d['config']['shards'], d['config']['workload'][:60]); ge = d.get('gradient_exchange'); [print('      exchange', r) for r in (ge['per_rank'] if ge else [])]"; }
run() {  # label, env, bench args
    local label=$1; local envs=$2; shift 2
    echo "== $label ($envs)"
    env $envs python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 \
        bench.py --gpus 2 --backend gloo --no-cpu-baseline --no-also "$@" 2>/tmp/dp2.err | brief || tail -5 /tmp/dp2.err
}
for tail in 1 0; do
    run "192x640, shards 3+2 (BASELINE config 4 shard sizes)" "CLSLAM_ASYNC_TAIL=$tail" --total-replay 4 --steps 20 --warmup 5
    run "192x640, shards 5+5 (weak: every rank the N=1 minibatch)" "CLSLAM_ASYNC_TAIL=$tail" --steps 20 --warmup 5
    run "384x1280, shards 2+1 (BASELINE config 5 shard sizes)" "CLSLAM_ASYNC_TAIL=$tail" --height 384 --width 1280 --total-replay 2 --steps 10 --warmup 3
done
echo "== single process for comparison"
python bench.py --replay 4 --steps 20 --warmup 5 --no-cpu-baseline --no-also | brief
python bench.py --replay 2 --steps 20 --warmup 5 --no-cpu-baseline --no-also | brief
