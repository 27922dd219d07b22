#!/usr/bin/env python
"""MFMA-busy fraction per kernel from two rocprofv3 --pmc passes (rocpd databases):
    pass A: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES      pass B: GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the share of the launch during which a SIMD's
matrix pipe is executing an MFMA (MI355X_MICROARCH.md: the counter counts cycles, 32 per v_mfma_f32_16x16x4_f32; GRBM_GUI_ACTIVE
is summed over the 8 XCDs).  The same number should come out of  flops / time / peak  at the clock the launch ran at.
    python tools/pmc_mfma_busy.py A.db B.db [filter]"""
import sqlite3
import sys
from collections import defaultdict


def load(path, filt):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute('pragma table_info(counters_collection)').fetchall()]
    ix = {n: i for i, n in enumerate(cols)}
    agg = defaultdict(lambda: defaultdict(list))
    for r in c.execute('select * from counters_collection').fetchall():
        kn = r[ix['kernel_name']] if 'kernel_name' in ix else r[ix['name']]
        if filt in kn:
            agg[kn][r[ix['counter_name']]].append(r[ix['value']])
    return agg


filt = sys.argv[3] if len(sys.argv) > 3 else 'conv'
a, b = load(sys.argv[1], filt), load(sys.argv[2], filt)
print(f'{"kernel":92s} launches  MFMA-busy  MFMA cycles/launch  GRBM_GUI_ACTIVE/8   waves resident per SIMD (SQ_WAVE_CYCLES*4 / active / 1024)')
rows = []
for kn in a:
    if kn not in b or 'SQ_VALU_MFMA_BUSY_CYCLES' not in a[kn] or 'GRBM_GUI_ACTIVE' not in b[kn]:
        continue
    mf = sum(a[kn]['SQ_VALU_MFMA_BUSY_CYCLES']) / len(a[kn]['SQ_VALU_MFMA_BUSY_CYCLES'])
    act = sum(b[kn]['GRBM_GUI_ACTIVE']) / len(b[kn]['GRBM_GUI_ACTIVE']) / 8
    wc = b[kn].get('SQ_WAVE_CYCLES')
    occ = (sum(wc) / len(wc) * 4 / act / 1024) if wc else float('nan')
    rows.append((mf * len(a[kn]['SQ_VALU_MFMA_BUSY_CYCLES']), kn, len(a[kn]['SQ_VALU_MFMA_BUSY_CYCLES']), mf / (act * 1024), mf, act, occ))
for _, kn, n, busy, mf, act, occ in sorted(rows, reverse=True):
    print(f'{kn[:92]:92s} {n:7d}   {busy:8.3f}   {mf:16.0f}   {act:15.0f}   {occ:6.2f}')
tot_mf = sum(r[4] * r[2] for r in rows)
tot_act = sum(r[5] * r[2] for r in rows)
if tot_act:
    print(f'all of the above: MFMA-busy {tot_mf / (tot_act * 1024):.3f} of the cycles they were on the GPU')
