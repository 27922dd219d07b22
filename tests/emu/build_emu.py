#!/usr/bin/env python
"""Compile the clslam kernel SOURCES for the host against the CPU emulator headers
(tests/emu/include) -> tests/emu/libclslam_emu.so.  Test infrastructure only."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

EMU = Path(__file__).resolve().parent
ROOT = EMU.parents[1]
CSRC = ROOT / 'cl-slam_amd' / 'csrc'
OBJ = EMU / 'build'
CXX = os.environ.get('EMU_CXX', '/opt/rocm/lib/llvm/bin/clang++')
FLAGS = ['-x', 'c++', '-std=c++17', '-O3', '-fPIC', '-DCLSLAM_DEVICE_BUILD=0', '-I', str(EMU / 'include'),
         '-I', str(CSRC / 'include'), '-Wno-unused-value', '-Wno-unknown-attributes', '-Wno-ignored-attributes',
         '-ffp-contract=off']
# The emulated MFMA is 16 fmaf() per lane: without -mfma each one is a libm call (x86-64 baseline has no FMA), 80 % of the
# CPU suite's run time.  Same bits either way (fmaf is correctly rounded); implicit contraction stays off.
try:
    if ' fma ' in Path('/proc/cpuinfo').read_text().split('flags', 1)[1].split('\n', 1)[0] + ' ':
        FLAGS.append('-mfma')
except Exception:    # noqa: BLE001 -- no /proc/cpuinfo: portable build
    pass


def build(force: bool = False, verbose: bool = False) -> Path:
    OBJ.mkdir(exist_ok=True)
    srcs = sorted(CSRC.glob('*.hip')) + [EMU / 'emu_runtime.cpp']
    hdrs = list(CSRC.glob('*.h')) + list((EMU / 'include').rglob('*.h')) + [ROOT / 'include' / 'clslam_hip.h']
    hdr_m = max(h.stat().st_mtime for h in hdrs)
    stamp = OBJ / 'flags.txt'
    if not stamp.exists() or stamp.read_text() != ' '.join(FLAGS):
        force = True             # objects of another flag set (e.g. built with -mfma on another host)
        stamp.write_text(' '.join(FLAGS))
    jobs = []
    for s in srcs:
        o = OBJ / (s.stem + '.o')
        if force or not o.exists() or o.stat().st_mtime < max(s.stat().st_mtime, hdr_m):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        r = subprocess.run([CXX, *FLAGS, '-c', str(s), '-o', str(o)], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'emu compile failed for {s.name}:\n{r.stderr[-6000:]}')
        if verbose:
            print('[emu cc]', s.name, flush=True)

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(cc, jobs))
    lib = EMU / 'libclslam_emu.so'
    if jobs or not lib.exists():
        objs = [str(OBJ / (s.stem + '.o')) for s in srcs]
        r = subprocess.run([CXX, '-shared', '-fPIC', '-o', str(lib), *objs, '-lpthread'], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'emu link failed:\n{r.stderr[-4000:]}')
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
