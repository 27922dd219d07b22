#!/usr/bin/env python
"""Fill the @@PLACEHOLDERS@@ of DESIGN.md section 5 / README.md from the committed measurement set profiles/<tag>_*
(python tools/fill_design.py r03): the numbers in the documents are the numbers in the files."""
import ast
import csv
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
P = ROOT / 'profiles'
d = json.loads((P / f'{tag}_bench.json').read_text().strip().splitlines()[-1])
r = d['roofline']
kern = r['kernel']
rows = list(csv.reader(open(P / f'{tag}_kernel_stats_serial.csv')))
steps = 15
row = next(x for x in rows[1:] if x[0] == kern)
other = {}
name = None
for line in (P / f'{tag}_other_configs.txt').read_text().splitlines():
    if line.startswith('== '):
        name = line[3:]
    elif line.startswith('{') and name:
        other[name] = ast.literal_eval(line[:line.rindex('}') + 1])
k0, k2, k32 = other['192x640 replay 0'], other['192x640 replay 2'], other['192x640 replay 32']
hi = other['384x1280 replay 8']
rnd = other['192x640 replay 4, uniform-random images']
e2e = d['also']['end_to_end']
vals = {
    'VALUE': f"{d['value']:.1f}", 'MS': f"{d['ms_per_step']:.2f}", 'E2E': f"{e2e['ms_per_frame']:.2f}",
    'E2EDEV': f"{e2e['ms_per_frame_with_device_outputs']:.2f}", 'NL': str(r['launches_per_step']), 'ACH': f"{r['achieved']:.1f}",
    'FRAC': f"{r['frac']:.2f}", 'US': f"{r['avg_launch_us']:.1f}", 'CSVN': f"{int(row[1]) / steps:g}", 'CSVUS': f"{float(row[3]):.1f}",
    'ALL': f"{r['all_conv_launches']['achieved']:.1f}", 'ALLMS': f"{r['all_conv_launches']['time_ms_per_step']:.2f}",
    'TRAF': 'null' if r['traffic'] is None else f"{r['traffic'] / 1e6:.1f}", 'ALG': f"{r['algorithmic_bytes_per_launch_avg'] / 1e6:.1f}",
    'RATIO': 'n/a' if r['traffic'] is None else f"{r['traffic'] / r['algorithmic_bytes_per_launch_avg']:.2f}",
    'CPU': f"{d['cpu_baseline']['value']:.1f}" if d.get('cpu_baseline') else 'n/a',
    'K0': f"{k0['ms_per_step']:.2f}", 'K0E': f"{k0['also']['end_to_end']['ms_per_frame']:.2f}", 'K2': f"{k2['ms_per_step']:.2f}",
    'K32': f"{k32['ms_per_step']:.1f}", 'HI': f"{hi['ms_per_step']:.1f}", 'RND': f"{rnd['value']:.1f}",
    'S5': f"{d['also']['ms_per_frame']:.1f}", 'S5K0': f"{k0['also']['ms_per_frame']:.1f}", 'S5K2': f"{k2['also']['ms_per_frame']:.1f}",
    'S5FPS': f"{d['also']['frames_per_s']:.1f}", 'E2EFPS': f"{e2e['frames_per_s']:.1f}", 'BUILD': d.get('build_id', '?'),
}
for doc in ('DESIGN.md', 'README.md'):
    path = ROOT / doc
    text = path.read_text()
    missing = set(re.findall(r'@@([A-Z0-9]+)@@', text)) - set(vals)
    if missing:
        raise SystemExit(f'{doc}: no value for {missing}')
    for k, v in vals.items():
        text = text.replace(f'@@{k}@@', v)
    path.write_text(text)
print(vals)
