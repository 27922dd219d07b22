import cProfile, pstats, sys, os, io
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/cl-slam_amd')
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
