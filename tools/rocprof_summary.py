#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as a small CSV:
    python tools/rocprof_summary.py gpurun_out/prof/r01_results.db profiles/r01_kernel_stats.csv
"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r'\(.*$', '', name)              # drop the argument list
    name = name.replace('void ', '').replace('clslam::', '')
    return name if len(name) <= 110 else name[:107] + '...'


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
    agg = {}
    for name, calls, total, avg, pct in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += calls
        a[1] += total
        a[2] += pct
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'percent'])
        for k, (calls, total, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, calls, round(total, 1), round(total / calls, 3), round(pct, 3)])
    print(f'wrote {out}: {len(agg)} kernels, {sum(a[1] for a in agg.values()) / 1e3:.2f} ms total')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
