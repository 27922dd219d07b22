#!/usr/bin/env python
"""Print per-kernel averages of the PMC counters in a rocprofv3 rocpd database."""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
filt = sys.argv[2] if len(sys.argv) > 2 else ''
cols = [r[1] for r in c.execute('pragma table_info(counters_collection)').fetchall()]
rows = c.execute('select * from counters_collection').fetchall()
ix = {n: i for i, n in enumerate(cols)}
agg = defaultdict(lambda: defaultdict(list))
for r in rows:
    kn = r[ix['kernel_name']] if 'kernel_name' in ix else r[ix['name']]
    if filt not in kn:
        continue
    agg[kn[:70]][r[ix['counter_name']]].append(r[ix['value']])
for kn, d in agg.items():
    print(kn)
    for cn, vals in sorted(d.items()):
        print(f'   {cn:32s} avg {sum(vals) / len(vals):16.1f}  n={len(vals)}')
