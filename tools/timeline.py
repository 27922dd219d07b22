#!/usr/bin/env python
"""Print the kernel timeline of ONE adapt step from a rocprofv3 --kernel-trace rocpd database
(start offset, duration, queue, short kernel name), plus busy/idle statistics per queue."""
import re
import sqlite3
import sys

db = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2   # which adam_kernel-delimited step
c = sqlite3.connect(db)
rows = c.execute('select name, start, end, queue_id, grid_x, grid_y, grid_z from kernels order by start').fetchall()
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
lo, hi = adam[which - 1] + 1, adam[which] + 1
step = rows[lo:hi]
t0 = step[0][1]
qs = sorted(set(r[3] for r in step))


def short(n):
    n = re.sub(r'\(.*$', '', n).replace('void ', '').replace('clslam::', '')
    return n[:58]


last_end = {q: t0 for q in qs}
for n, s, e, q, gx, gy, gz in step:
    gap = (s - last_end[q]) / 1e3
    print(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q{qs.index(q)}  gap {gap:6.1f}  grid {gx // 256 if gx else 0:6d}x{gy}x{gz}  {short(n)}')
    last_end[q] = e
print('step wall', (step[-1][2] - t0) / 1e3, 'us;  kernels', len(step))
for q in qs:
    busy = sum(e - s for n, s, e, qq, *_ in step if qq == q) / 1e3
    print(f'queue {qs.index(q)}: busy {busy:.1f} us, {sum(1 for r in step if r[3] == q)} kernels')
# union busy time
ev = sorted([(s, 1) for _, s, e, *_ in step] + [(e, -1) for _, s, e, *_ in step])
act = 0; busy = 0; prev = ev[0][0]
for t, d in ev:
    if act > 0:
        busy += t - prev
    act += d; prev = t
print('GPU busy (any kernel running):', busy / 1e3, 'us')
