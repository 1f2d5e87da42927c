"""Worker of test_gpu_api.test_train_clis_under_two_ranks: one rank of a 2-process run of a train CLI on ONE GPU (gloo
moves the CUDA tensors through the host, so both ranks can share the device)."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clvae_amd  # noqa: F401,E402


def main():
    which, out = sys.argv[1], sys.argv[2]
    argv = sys.argv[3:]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)      # before train(): it keeps an existing group
    if which == 'cl_vae':
        from clvae_amd.cl_vae import train as TR
    else:
        from clvae_amd.cl_vrnn import train as TR
    args = TR.build_parser().parse_args(argv)
    args.seed = 3
    np.random.seed(0)
    model, best = TR.train(args)
    np.savez(out % rank, **model.engine.P.get_weights())
    with open((out % rank) + '.loss', 'w') as f:
        f.write(repr(model.history.history['loss']))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
