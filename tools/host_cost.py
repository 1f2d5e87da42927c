"""Host-side cost of a training step: how long the CPU needs to enqueue K steps (gather launch + hipGraph replay)
against how long the GPU needs to run them.  If the two are close the bench is host-bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as Bn
import clvae_amd  # noqa
from clvae_amd.trainer import TrainStep

def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device('cuda', 0)
    w = Bn.WORKLOADS[wl]
    eng, cfg = Bn.make_engine(w, dev)
    B = w['B']
    X_all, Xp_all, w_all = Bn.synthetic_windows(w, 4 * B, 1234, dev)
    ts = TrainStep(eng, seed=1234, use_graph=True)
    def run(k, stage=True):
        for i in range(k):
            j = i % 4
            if stage:
                ts.stage_batch(X_all[j * B:(j + 1) * B], Xp_all[j * B:(j + 1) * B], w_all[j * B:(j + 1) * B])
            ts.step()
    run(5); torch.cuda.synchronize()
    for rep in range(4):
        for stage in (True, False):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); run(K, stage); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            print("rep %d stage=%d: host enqueue %.4f ms/step, total %.4f ms/step" % (rep, stage, 1e3 * (t1 - t0) / K, 1e3 * (t2 - t0) / K))
    # 20-step run like the driver's
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); run(20); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("20-step run: %.4f ms/step" % (1e3 * (t2 - t0) / 20))
main()
