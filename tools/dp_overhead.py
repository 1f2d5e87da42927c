"""What the data-parallel step's schedule costs on ONE GPU (round-5 verdict, item 6): configuration 3 (256 x 128), ms per step of
  single   : the single-GPU step, one hipGraph
  split0   : the DP schedule (CLV_FORCE_DP_GRAPHS=1) with no-op collectives: two graphs + plain launches around where the
             two all-reduces would be
  split    : ... with the two all-reduces issued for REAL on a one-rank RCCL group (CLV_DP_REAL_COLLECTIVES=1), eager RCCL
             calls on the side stream
  whole    : ... and the whole step, both collectives included, captured as ONE graph (CLV_CAPTURE_COLLECTIVES=1); the note says
             whether this box's RCCL let itself be captured
alternating, --reps rounds of --steps steps; host issue time per step next to it (clock stopped before the synchronize).
Run on the GPU box: python tools/dp_overhead.py > gpurun_out/r06_dp_overhead.txt"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--workload', default='cfg3')
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import bench
    import clvae_amd  # noqa: F401
    from clvae_amd.trainer import TrainStep
    dev = torch.device('cuda:0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29641', rank=0, world_size=1)
    w = bench.WORKLOADS[args.workload]
    B = w['B']
    modes = {'single': {}, 'split0': {'CLV_FORCE_DP_GRAPHS': '1'},
             'split': {'CLV_FORCE_DP_GRAPHS': '1', 'CLV_DP_REAL_COLLECTIVES': '1', 'CLV_CAPTURE_COLLECTIVES': '0'},
             'whole': {'CLV_FORCE_DP_GRAPHS': '1', 'CLV_DP_REAL_COLLECTIVES': '1', 'CLV_CAPTURE_COLLECTIVES': '1'}}
    keys = sorted({k for m in modes.values() for k in m})
    steps = {}
    for name, env in modes.items():
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        eng, cfg = bench.make_engine(w, dev)
        X_all, Xp_all, w_all = bench.synthetic_windows(w, 4 * B, 1234, dev)
        ts = TrainStep(eng, seed=1234, use_graph=True)
        ts.bind_batches(X_all, Xp_all, w_all, idx=None, period=4, stride=B)
        for _ in range(5):
            ts.step()
        torch.cuda.synchronize()
        steps[name] = (ts, eng)
        print("%-7s graphs=%s capture_note=%s" % (name, None if ts._graphs is None else
                                                  [('graph' if g is not None else 'plain') if not isinstance(g, str) else g for g in ts._graphs],
                                                  ts.capture_note), flush=True)
    for k in keys:
        os.environ.pop(k, None)
    for r in range(args.reps):
        line = []
        for name, (ts, eng) in steps.items():
            for _ in range(20):
                ts.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ts.step()
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            line.append("%s %.4f ms (host issue %.1f us)" % (name, 1e3 * dt / args.steps, 1e6 * t_issue / args.steps))
        print("round %d: " % r + " | ".join(line), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
