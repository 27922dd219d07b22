#!/usr/bin/env python
"""Two consecutive END-TO-END frames (tools/e2e_frames.py under rocprofv3 --kernel-trace --memory-copy-trace) as one timeline:
every memory copy (direction, bytes) and, per HIP queue, the kernels condensed into runs -- where the host->device copies of frame
t+1 sit relative to frame t's backward / optimizer step and frame t+1's first kernels.

    python tools/timeline_e2e.py <rocpd .db> [frame index from the end, default -3]"""
import re
import sqlite3
import sys

db = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
c = sqlite3.connect(db)
tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
rows = c.execute('select name, start, end, queue_id from kernels order by start').fetchall()
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
# a frame = (end of the previous optimizer step, end of the one after next]: two frames
lo, hi = adam[which - 1] + 1, adam[which + 1] + 1
seg = rows[lo:hi]
t0, t1 = rows[adam[which - 1]][1], seg[-1][2]
copies = []
ctab = next((t for t in ('memory_copies', 'memory_copy') if t in tables), None)
if ctab is None:
    ctab = next((t for t in tables if 'copy' in t.lower() or 'copies' in t.lower()), None)
if ctab is not None:
    cols = [r[1] for r in c.execute(f'pragma table_info({ctab})')]
    name_c = 'name' if 'name' in cols else cols[0]
    size_c = next((x for x in ('size', 'bytes') if x in cols), None)
    q = f'select {name_c}, start, end, {size_c or 0} from {ctab} where end >= {t0} and start <= {t1} order by start'
    copies = c.execute(q).fetchall()
else:
    print('# no memory-copy table in this database; tables:', tables)


def short(n):
    n = re.sub(r'\(.*$', '', n).replace('void ', '').replace('clslam::', '')
    return re.sub(r'<.*', '', n)[:28]


qs = sorted(set(r[3] for r in seg))
events = []
for n, s, e, q in seg:
    events.append((s, e, f'q{qs.index(q)}', short(n)))
# condense consecutive kernels of one queue separated by < 8 us into runs
runs = []
for q in qs:
    cur = None
    for n, s, e, qq in seg:
        if qq != q:
            continue
        if cur is not None and s - cur[1] < 8000:
            cur[1] = max(cur[1], e); cur[3] += 1; cur[4] = short(n)
        else:
            if cur is not None:
                runs.append(tuple(cur))
            cur = [s, e, f'q{qs.index(q)}', 1, short(n), short(n)]
    if cur is not None:
        runs.append(tuple(cur))
lines = []
for s, e, q, cnt, last, first in runs:
    lines.append((s, f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  {q}  {cnt:3d} kernels  {first} .. {last}'))
for n, s, e, size in copies:
    kind = 'H2D' if 'HOST_TO_DEVICE' in str(n).upper() or 'H2D' in str(n).upper() else 'D2H' if 'DEVICE_TO_HOST' in str(n).upper() or 'D2H' in str(n).upper() else str(n)[-16:]
    lines.append((s, f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  copy {kind:4s} {int(size) / 1e6:8.3f} MB'))
print(f'# two end-to-end frames: {(t1 - t0) / 1e3:.1f} us between the ends of three consecutive adam_kernel launches; marks: adam_kernel ends at')
for i in (which - 1, which, which + 1):
    print(f'#   {(rows[adam[i]][2] - t0) / 1e3:9.1f} us')
for _, ln in sorted(lines):
    print(ln)
h2d = [(s, e, sz) for n, s, e, sz in copies if 'HOST_TO_DEVICE' in str(n).upper() or 'H2D' in str(n).upper()]
if h2d:
    tot = sum(sz for *_, sz in h2d)
    busy = sum(e - s for s, e, _ in h2d)
    print(f'# H2D in the window: {len(h2d)} copies, {tot / 1e6:.2f} MB, {busy / 1e3:.1f} us of copy time, {tot / max(busy, 1):.1f} GB/s while copying')
