"""cProfile of the host side of 50 eager adapt steps at B=1 (where the step is closest to host-bound)."""
import cProfile, pstats, sys, os, io
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'cl-slam_amd'))
os.environ['CLSLAM_HIPGRAPH'] = '0'
import torch, bench
from clslam_hip import synth
p = bench.build_predictor(192, 640, 1)
batch = {k: v.cuda() for k, v in synth.make_batch(1, 192, 640, seed=0).items()}
for _ in range(5): p.adapt(None, batch, steps=1)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50): p.adapt(None, batch, steps=1)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
