#!/usr/bin/env python
"""profiles/pmc_traffic.json from the two rocprofv3 --pmc passes of tools/measure_round.sh:

    python tools/pmc_traffic.py <FETCH_SIZE.db> <WRITE_SIZE.db> <out.json> [kernel-name filter ...]

Per kernel (name as rocprofv3 prints it, argument list dropped): average FETCH_SIZE / WRITE_SIZE in KB and
traffic_bytes_per_launch = 2 * FETCH + WRITE  (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B;
the TCC_EA counters sit behind the per-XCD L2s and include Infinity-Cache hits: L2-miss traffic, an upper bound on HBM
bytes).  bench.py reads the entry of its dominant kernel by NAME -- no entry, no `traffic`."""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute('pragma table_info(counters_collection)').fetchall()]
    ix = {n: i for i, n in enumerate(cols)}
    agg = defaultdict(list)
    for r in c.execute('select * from counters_collection'):
        if r[ix['counter_name']] != counter:
            continue
        kn = r[ix['kernel_name']] if 'kernel_name' in ix else r[ix['name']]
        kn = re.sub(r'\(.*$', '', kn).replace('void ', '').replace('clslam::', '')
        agg[kn].append(r[ix['value']])
    return agg


def main():
    fetch_db, write_db, out = sys.argv[1:4]
    filters = sys.argv[4:] or ['conv3x3', 'conv_igemm', 'wgrad']
    f, w = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
    res = {}
    for kn in sorted(set(f) & set(w)):
        if not any(s in kn for s in filters):
            continue
        fk, wk = sum(f[kn]) / len(f[kn]), sum(w[kn]) / len(w[kn])
        res[kn] = {'FETCH_SIZE_KB_avg': round(fk, 1), 'WRITE_SIZE_KB_avg': round(wk, 1), 'launches': len(f[kn]),
                   'traffic_bytes_per_launch': int((2 * fk + wk) * 1024),
                   # round 6 calibration (tools/micro/pmc_calib.hip, profiles/r06_pmc_calibration.txt): WRITE_SIZE is exact for
                   # coalesced, 64-B-segment and write-through stores; FETCH_SIZE is HALF the bytes of >= 128-B coalesced reads but
                   # EXACT for 64-B segment reads (the 16-channel patch rows of a 64-channel layer) -- so the true L2-miss bytes of
                   # a kernel that mixes both lie between FETCH + WRITE and 2 FETCH + WRITE
                   'traffic_bytes_per_launch_lower_bound': int((fk + wk) * 1024),
                   'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/measure_round.sh, serial launch '
                           'order); FETCH_SIZE doubled per MI355X_MICROARCH.md; includes Infinity-Cache hits (L2-miss traffic)'}
    # identity of the kernel sources the counters were taken on (bench.py refuses to pair them with another build's timings)
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / 'cl-slam_amd'))
    from clslam_hip import _lib
    res['_build_id'] = _lib.build_id()
    json.dump(res, open(out, 'w'), indent=1)
    print(f'{len(res)} kernels -> {out}')


if __name__ == '__main__':
    main()
