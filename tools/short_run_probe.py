"""Why a 20-step run after 5 warm-up steps is slower than a 200-step one: per-chunk step time and host issue time."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, bench
from clvae_amd.trainer import TrainStep
dev = torch.device('cuda:0')
w = bench.WORKLOADS['cfg3']; B = w['B']
eng, cfg = bench.make_engine(w, dev)
X, Xp, wv = bench.synthetic_windows(w, 4 * B, 1234, dev)
ts = TrainStep(eng, seed=1234, use_graph=True)
def run(k):
    for i in range(k):
        j = i % 4
        ts.stage_batch(X[j*B:(j+1)*B], Xp[j*B:(j+1)*B], wv[j*B:(j+1)*B]); ts.step()
run(5); torch.cuda.synchronize()
out = []
iss = []
for c in range(12):
    t0 = time.perf_counter(); run(20); t1 = time.perf_counter(); torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 20 * 1e3); iss.append((t1 - t0) / 20 * 1e3)
print("chunks of 20, sync between:", ["%.4f" % v for v in out])
print("  host issue per step     :", ["%.4f" % v for v in iss])
t0 = time.perf_counter(); run(240); torch.cuda.synchronize(); print("240 in one go: %.4f" % ((time.perf_counter() - t0) / 240 * 1e3))
# host-only cost of issuing a step
t0 = time.perf_counter(); run(20); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("issue 20 steps: %.3f ms, then wait %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
