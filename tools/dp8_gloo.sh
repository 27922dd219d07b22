#!/bin/bash
# EIGHT ranks sharing the one MI355X of the box (backend gloo on device tensors): BASELINE configs 4 and 5 with their real
# shardings -- K=32 at 192x640 as 5,4,4,4,4,4,4,4 and K=8 at 384x1280 as 2,1,1,1,1,1,1,1 -- as a FUNCTIONAL run of the 8-rank
# code path (sharding, global sample weights, rank-0 smoothness terms, one all-reduce of the flat arena per step, async tail).
# Eight processes time-slice one GPU and gloo stages the arena through the host: the times say nothing about scaling.
#   bash tools/dp8_gloo.sh > profiles/rNN_dp8_gloo.txt
set -u
brief() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('value', 'unit', 'ms_per_step', 'n_gpus')}, 'shards', d['config']['shards'], 'loss', round(d['config']['loss'], 6), d['config']['workload'][:60])"; }
run() {
    echo "== $1"; shift
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29641 \
        bench.py --gpus 8 --backend gloo --no-cpu-baseline --no-also "$@" 2>/tmp/dp8.err | brief || tail -5 /tmp/dp8.err
}
run "config 4: 192x640, K=32 over 8 ranks" --total-replay 32 --steps 5 --warmup 2
run "config 5: 384x1280, K=8 over 8 ranks" --height 384 --width 1280 --total-replay 8 --steps 5 --warmup 2
echo "== single process, same minibatches (loss of the last step for comparison)"
python bench.py --replay 32 --steps 5 --warmup 2 --no-cpu-baseline --no-also 2>/dev/null | brief
python bench.py --height 384 --width 1280 --replay 8 --steps 5 --warmup 2 --no-cpu-baseline --no-also 2>/dev/null | brief
