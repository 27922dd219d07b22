#!/bin/bash
# A/B of environment knobs inside ONE gpurun call (same box, same clocks):  bash tools/ab.sh "<bench args>" "VAR=1" "VAR=2 OTHER=x" ...
# prints ms per step (median of the timed blocks), min / max and the 5-step frame per variant; "-" = no extra environment.
ARGS=${1:---steps 20 --warmup 5 --blocks 9 --no-cpu-baseline}
shift
for v in "$@"; do
    if [ "$v" == "-" ]; then e=""; else e="$v"; fi
    env $e python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d.get('also') or {}
print('%-44s step %.3f ms (min %.3f max %.3f)  5-step frame %s  per optimizer step %s' % ('$v', d['ms_per_step'], d.get('ms_per_step_min',0), d.get('ms_per_step_max',0), a.get('ms_per_frame'), a.get('ms_per_optimizer_step')))"
done
