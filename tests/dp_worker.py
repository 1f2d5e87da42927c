"""Worker of test_gpu_api.test_two_rank_dp_step_matches_single_process: one rank of a 2-process data-parallel run on ONE
GPU (gloo moves the CUDA gradient buckets through the host, so both ranks can share the device)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clvae_amd  # noqa: F401,E402
from clvae_amd.engine import VrnnEngine  # noqa: E402
from clvae_amd.trainer import TrainStep  # noqa: E402
from oracle import clvae_oracle as O  # noqa: E402


def main():
    rank, world, out = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1]
    use_graph = sys.argv[2].startswith("graph")
    dropout = 0.3 if sys.argv[2].endswith("dropout") else 0.0      # LSTM(dropout=p): masks by GLOBAL row, like the noise
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    B, T, L, C = 4, 6, 2, 3
    cfg = O.vrnn_config(latent_dim=L, seq_length=T, n_classes=C, use_x_prev=True)
    p = {k: np.asarray(v, dtype=np.float32) for k, v in O.vrnn_init_params(cfg, seed=11).items()}
    rng = np.random.default_rng(0)
    win = (rng.random((world * B, T + 1, 88)) < 0.05).astype(np.float32)
    wt = np.eye(C, dtype=np.float32)[rng.integers(0, C, world * B)]
    eng = VrnnEngine(dict(cfg, dropout=dropout), B, dev)
    eng.P.set_weights(p)
    ts = TrainStep(eng, seed=5, rank=rank, world=world, use_graph=use_graph)
    assert ts.ar is not None and ts.split_update
    sl = slice(rank * B, (rank + 1) * B)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    for _ in range(4):
        ts.stage_batch(t(win[sl, 1:]), t(win[sl, :-1]), t(wt[sl]))
        ts.step()
    torch.cuda.synchronize()
    np.savez(out % rank, **eng.P.get_weights())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
