"""Step time with and without the per-step staging launch (the one kernel outside the step's hipGraph)."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, bench
from clvae_amd.trainer import TrainStep
dev = torch.device('cuda:0')
w = bench.WORKLOADS['cfg3']; B = w['B']
eng, cfg = bench.make_engine(w, dev)
X, Xp, wv = bench.synthetic_windows(w, 4 * B, 1234, dev)
ts = TrainStep(eng, seed=1234, use_graph=True)
def run(k, stage=True):
    for i in range(k):
        j = i % 4
        if stage: ts.stage_batch(X[j*B:(j+1)*B], Xp[j*B:(j+1)*B], wv[j*B:(j+1)*B])
        ts.step()
run(60); torch.cuda.synchronize()
for stage in (True, False, True, False):
    t0 = time.perf_counter(); run(200, stage); torch.cuda.synchronize()
    print("stage_batch per step" if stage else "no staging (graph only)", "%.4f ms" % ((time.perf_counter() - t0) / 200 * 1e3))
