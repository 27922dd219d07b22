#!/usr/bin/env python
"""Build libclslam_hip.so for gfx950 from the HIP sources in this directory.

    python cl-slam_amd/csrc/build.py [--force]

hipcc cross-compiles without a GPU; objects are cached by source mtime.  The shared library is
written IN-TREE (cl-slam_amd/lib/libclslam_hip.so) so it travels to the GPU box with the
repository snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent
ROOT = CSRC.parents[1]
LIB_DIR = CSRC.parent / 'lib'
OBJ_DIR = CSRC / 'build'
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DCLSLAM_DEVICE_BUILD=1',
         '-I', str(CSRC / 'include'), '-Wno-unused-result'] + os.environ.get('CLSLAM_HIPCC_EXTRA', '').split()
# per-source flags (none at present)
FILE_FLAGS = {}


def source_id() -> str:
    """Identity of the kernel sources (every .hip / .h under csrc/ + the public header, i.e. including the conv pick-config
    table): 16 hex digits.  build() stores it next to the library; bench.py prints it and only pairs its timings with
    rocprofv3 counter files (profiles/pmc_traffic.json) that carry the same id."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(CSRC.glob('*.hip')) + sorted(CSRC.glob('*.h')) + sorted((CSRC / 'include').rglob('*.h')) + [ROOT / 'include' / 'clslam_hip.h']
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    # the flags are part of the identity: a diagnostic build (CLSLAM_HIPCC_EXTRA=-DCLSLAM_EXACT_DIV, probe builds of conv_wino.hip)
    # must not pass for the production library
    h.update(repr((FLAGS, sorted(FILE_FLAGS.items()))).encode().replace(str(CSRC).encode(), b'<csrc>'))
    return h.hexdigest()[:16]


def _deps_mtime() -> float:
    hs = list(CSRC.glob('*.h')) + list((CSRC / 'include').rglob('*.h')) + [ROOT / 'include' / 'clslam_hip.h']
    return max(h.stat().st_mtime for h in hs)


def _embedded_id(lib: Path) -> str:
    """the id a built library carries (clslam_build_id), without loading it: the string sits in .rodata behind a marker"""
    try:
        data = lib.read_bytes()
        i = data.find(b'clslam-build-id:')
        return data[i + 16:i + 32].decode() if i >= 0 else ''
    except OSError:
        return ''


def build(force: bool = False, verbose: bool = True) -> Path:
    LIB_DIR.mkdir(exist_ok=True)
    OBJ_DIR.mkdir(exist_ok=True)
    srcs = sorted(CSRC.glob('*.hip'))
    hdr_m = _deps_mtime()
    sid = source_id()
    lib = LIB_DIR / 'libclslam_hip.so'
    # The id is a statement about the library: it is compiled INTO it (capi.hip, -DCLSLAM_BUILD_ID) whenever anything is
    # rebuilt, and a library whose embedded id is not the id of the sources present is rebuilt whatever the mtimes say
    # (sources restored with older timestamps, an object cache of another checkout).
    if lib.exists() and _embedded_id(lib) != sid:
        force = True
    jobs = []
    for s in srcs:
        o = OBJ_DIR / (s.stem + '.o')
        if force or not o.exists() or o.stat().st_mtime < max(s.stat().st_mtime, hdr_m):
            jobs.append((s, o))
    if jobs and not any(s.stem == 'capi' for s, _ in jobs):
        jobs.append((CSRC / 'capi.hip', OBJ_DIR / 'capi.o'))        # re-stamp

    def cc(job):
        s, o = job
        extra = [f'-DCLSLAM_BUILD_ID="clslam-build-id:{sid}"'] if s.stem == 'capi' else []
        cmd = [HIPCC, *FLAGS, *FILE_FLAGS.get(s.stem, []), *extra, '-c', str(s), '-o', str(o)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {s.name}:\n{r.stderr[-4000:]}')
        if verbose:
            print(f'[hipcc] {s.name}', flush=True)
        return o

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [str(OBJ_DIR / (s.stem + '.o')) for s in srcs]
    if jobs or not lib.exists():
        r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(lib), *objs],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stderr[-4000:]}')
        if verbose:
            print(f'[link] {lib}', flush=True)
        (LIB_DIR / 'libclslam_hip.build_id').write_text(sid + '\n')     # (a readable copy; the library itself is the authority)
    return lib


if __name__ == '__main__':
    build(force='--force' in sys.argv)
