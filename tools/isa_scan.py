#!/usr/bin/env python
"""Scan the gfx950 ISA of every kernel in cl-slam_amd/csrc for memory operations that hipcc's wait-count pass serialised:

    python tools/isa_scan.py            # report
    python tools/isa_scan.py <substr>   # the memory-op skeleton of the kernels whose demangled name contains <substr>

(runs in the GPU-less build container: hipcc cross-compiles).  Round 5 found every output store of the tiled conv, implicit-GEMM
and stem kernels behind an `s_waitcnt vmcnt(0)` (stores count in vmcnt on gfx9): 8 ... 32 serial memory round trips per
workgroup.  Reported here:
  * stores separated by `s_waitcnt vmcnt(0)`        (S W S): an epilogue that stores one element per round trip
  * single loads each followed by `vmcnt(0)`        (L W L): a gather that fetches one operand per round trip
Skeleton letters: L global load, l LDS-DMA load, S global store, A atomic, wN = s_waitcnt vmcnt(N), M<n> = n MFMAs, B barrier,
. branch."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / 'cl-slam_amd' / 'csrc'


def compile_all(out: Path):
    for src in sorted(CSRC.glob('*.hip')):
        subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-DCLSLAM_DEVICE_BUILD=1', '-I', str(CSRC / 'include'),
                        '-S', '--cuda-device-only', str(src), '-o', str(out / (src.stem + '.s'))],
                       check=True, stderr=subprocess.DEVNULL)


def kernels(asm: Path):
    name, seq = None, []
    for ln in asm.read_text().split('\n'):
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            name, seq = m.group(1), []
            continue
        if name is None:
            continue
        if 's_endpgm' in ln:
            yield name, seq
            name = None
        elif re.search(r'\b(global|buffer)_load', ln):
            seq.append('l' if 'lds' in ln else 'L')
        elif 'global_store' in ln or 'buffer_store' in ln:
            seq.append('S')
        elif 'global_atomic' in ln:
            seq.append('A')
        elif 's_waitcnt' in ln and re.search(r'vmcnt\((\d+)\)', ln):
            seq.append('w' + re.search(r'vmcnt\((\d+)\)', ln).group(1))
        elif 'v_mfma' in ln:
            seq.append('M')
        elif 's_barrier' in ln:
            seq.append('B')
        elif 's_cbranch' in ln:
            seq.append('.')


def main():
    want = sys.argv[1:]
    with tempfile.TemporaryDirectory() as d:
        out = Path(d)
        compile_all(out)
        for asm in sorted(out.glob('*.s')):
            for name, seq in kernels(asm):
                dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
                flat = ''.join('W' if t == 'w0' else t[0] for t in seq if t != '.' and not (t[0] == 'w' and t != 'w0'))
                if want:
                    if any(w in dn for w in want):
                        s = re.sub(r'M+', lambda m: f'M{len(m.group(0))}', ''.join(seq))
                        print(dn[:110])
                        print('    ' + s[:600])
                    continue
                stores, serial = flat.count('S'), len(re.findall(r'S[L]*W(?=[LW]*S)', flat))
                gathers = len(re.findall(r'LW(?=L)', flat))
                if stores >= 4 and serial >= 3:
                    print(f'{asm.stem:14s} {stores:3d} stores, {serial:3d} of them behind a vmcnt(0) of their own   {dn[:100]}')
                if gathers >= 3:
                    print(f'{asm.stem:14s} {gathers:3d} single loads each followed by vmcnt(0)                 {dn[:100]}')


if __name__ == '__main__':
    main()
