#!/usr/bin/env python
"""Probe / trace builds of the kernel library beside the production one (measurement tooling, never loaded by the product):

    python tools/build_variant.py <tag> <source stem>[,<stem>...] -D... [-D...]

compiles the named sources of cl-slam_amd/csrc with the extra flags, links them with the production objects of every other source
and writes cl-slam_amd/lib/variants/libclslam_hip_<tag>.so (git-ignored; travels to the GPU box with the snapshot).  Tools pick a
variant with CLSLAM_TOOL_LIB=<tag> (tools/_variant.py) -- the packages themselves only ever load lib/libclslam_hip.so."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'cl-slam_amd' / 'csrc'))
import build as hip_build  # noqa: E402


def main():
    tag, stems, extra = sys.argv[1], sys.argv[2].split(','), sys.argv[3:]
    hip_build.build(verbose=False)                      # production objects up to date
    vdir = hip_build.OBJ_DIR / 'variants' / tag
    vdir.mkdir(parents=True, exist_ok=True)
    objs = []
    for s in sorted(hip_build.CSRC.glob('*.hip')):
        if s.stem in stems:
            o = vdir / (s.stem + '.o')
            cmd = [hip_build.HIPCC, *hip_build.FLAGS, *extra, '-c', str(s), '-o', str(o)]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit(f'hipcc failed for {s.name}:\n{r.stderr[-4000:]}')
            objs.append(str(o))
        else:
            objs.append(str(hip_build.OBJ_DIR / (s.stem + '.o')))
    out = hip_build.LIB_DIR / 'variants'
    out.mkdir(exist_ok=True)
    lib = out / f'libclslam_hip_{tag}.so'
    r = subprocess.run([hip_build.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(lib), *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(f'link failed:\n{r.stderr[-4000:]}')
    print(lib)


if __name__ == '__main__':
    main()
