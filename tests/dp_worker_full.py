"""Worker of tests/test_gpu_bench_dp.py::test_two_ranks_at_the_benchmarks_shape_*: one rank of a data-parallel run at the
shape bench.py times per GPU (256 x 128), both ranks on ONE GPU (gloo moves the CUDA gradient buckets through the host).

The step is the one Model.fit and bench.py replay: TrainStep(use_graph=True) with every default, the mini-batch assembled
inside the captured step from the device-resident data set (bind_batches: stride = the GLOBAL batch, offset = rank * B),
noise drawn in the kernels at GLOBAL row indices, the two gradient buckets on the side stream, the optimizer in two pieces.
argv: out-pattern grid(coarse|fine) B T L C steps nb data-seed
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clvae_amd  # noqa: F401,E402
from clvae_amd.engine import VrnnEngine  # noqa: E402
from clvae_amd.trainer import TrainStep  # noqa: E402
from oracle import clvae_oracle as O  # noqa: E402   (the initial weights only: the checker lives in the test)


def dataset(G, T, C, nb, seed):
    """nb global batches of G windows: (frames t = 1..T, frames t = 0..T-1, one-hot keys); the test draws the same."""
    rng = np.random.default_rng(seed)
    win = rng.random((nb * G, T + 1, 88)) < 0.0443
    keys = np.eye(C)[rng.integers(0, C, nb * G)]
    return win, keys


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out, grid = sys.argv[1], sys.argv[2]
    B, T, L, C, steps, nb, dseed = (int(a) for a in sys.argv[3:10])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    G = world * B
    cfg = dict(O.vrnn_config(latent_dim=L, seq_length=T, n_classes=C, use_x_prev=True), fine_grid=(grid == "fine"))
    p = {k: np.asarray(v, dtype=np.float32) for k, v in O.vrnn_init_params(cfg, seed=5).items()}
    win, keys = dataset(G, T, C, nb, dseed)
    u8 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.uint8), device=dev)
    cur, hist = u8(win[:, 1:].reshape(nb * G, -1)), u8(win[:, :-1].reshape(nb * G, -1))
    wd = torch.as_tensor(np.ascontiguousarray(keys, dtype=np.float32), device=dev)
    eng = VrnnEngine(cfg, B, dev)
    eng.P.set_weights(p)
    ts = TrainStep(eng, seed=4321, rank=rank, world=world)
    assert ts.use_graph and ts.ar is not None and ts.ar.live and ts.split_update and ts.pre_in_tail
    assert eng.fine_grid == (grid == "fine")
    ts.bind_batches(cur, hist, wd, idx=None, period=nb, stride=G, offset=rank * B)
    losses = []
    for _ in range(steps):
        ts.step()
        torch.cuda.synchronize()
        losses.append({k: float(v) for k, v in eng.losses().items()})
    assert ts._graphs is not None and len(ts._graphs) >= 3 and eng.frames_exact_bf16
    assert int(eng.P.iterations.item()) == steps
    np.savez(out % rank, **eng.P.get_weights())
    with open((out % rank) + ".json", "w") as f:
        json.dump(dict(losses=losses, graphs=len([g for g in ts._graphs if g is not None]), capture_note=ts.capture_note), f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
