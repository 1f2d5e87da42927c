"""Data parallelism for the training step: one process per GPU, torch.distributed (RCCL on
ROCm, gloo in the CPU tests), no collective on the data path except ONE gradient all-reduce
per step over the flat fp32 gradient buffer, issued in two buckets on a side HIP stream: the hW
kernel bucket (87 % of the bytes, produced early by the reordered backward pass) reduces under the
remaining weight-gradient products, the small rest right before Adam.

The reference has no distributed code; this layer is new.  Sharding contract:
  * global batch = world * local batch; rank r owns global rows [r*B, (r+1)*B) of every batch;
  * noise is drawn from the counter-based Philox stream at GLOBAL sample indices, so the
    same global batch sees the same eps at any world size;
  * gradients are averaged (sum / world) => equal to the global-batch mean the reference's
    single-process fit() would compute;
  * parameters and optimizer state are replicated (weight-norm needs whole columns).
"""
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* if world > 1.
    CLV_BENCH_SHARE_GPU=1: every rank on cuda:0 with gloo carrying the gradient buckets through the host (RCCL cannot put
    two ranks on one device) -- how a one-GPU box runs the whole world > 1 schedule; returns local = 0."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if shared_gpu():
        local = 0
        backend = backend or "gloo"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shared_gpu():
    import os
    return os.environ.get("CLV_BENCH_SHARE_GPU") == "1"


def meta_device(dev):
    """Where the small bookkeeping tensors of a collective (times, counts) live: the device under RCCL, the host under gloo."""
    if dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return dev


def shard_rows(global_batch, rank, world):
    """[start, stop) of this rank's rows inside a global batch (must divide evenly, like
    PianoData.adjust_for_batch_size guarantees for batch_size in the reference)."""
    if global_batch % world:
        raise ValueError("global batch %d is not divisible by world size %d" % (global_batch, world))
    b = global_batch // world
    return rank * b, (rank + 1) * b


def eps_first_index(global_row0, per_row):
    """First Philox element index of this rank's slice when every global row owns `per_row` draws."""
    return int(global_row0) * int(per_row)


class GradAllReduce:
    """Bucketed average of a flat gradient tensor: bucket 'main' = everything except the tail
    range, bucket 'tail' = [tail_off, tail_off + tail_n).  On CUDA tensors the collectives run on
    a side stream; `wait()` joins it back into the current stream."""

    def __init__(self, flat_grads, tail_off=0, tail_n=0, group=None, always=False):
        """always: issue the collectives even in a group of one rank (a one-GPU box can then run -- and try to capture --
        the real RCCL calls of the schedule; the average over one rank is the identity)."""
        self.flat = flat_grads
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.live = self.world > 1 or (bool(always) and dist.is_initialized())
        n = flat_grads.numel()
        self.tail = flat_grads[tail_off:tail_off + tail_n] if tail_n else None
        self.main = []
        if tail_n:
            if tail_off > 0:
                self.main.append(flat_grads[:tail_off])
            if tail_off + tail_n < n:
                self.main.append(flat_grads[tail_off + tail_n:])
        else:
            self.main.append(flat_grads)
        self.cuda = flat_grads.is_cuda
        self.side = torch.cuda.Stream(device=flat_grads.device) if self.cuda else None
        self.tail_done = None

    def _reduce(self, t):
        if not self.live:
            return
        if dist.get_backend(self.group) == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.div_(self.world)

    def _on_side(self, tensors):
        if not self.live:
            return
        if self.cuda:
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                for t in tensors:
                    self._reduce(t)
        else:
            for t in tensors:
                self._reduce(t)

    def reduce_main(self):
        self._on_side(self.main)

    def reduce_tail(self):
        if self.tail is not None:
            self._on_side([self.tail])
            if self.cuda and self.live:
                self.tail_done = torch.cuda.Event()
                self.tail_done.record(self.side)

    def wait_tail(self):
        """The current stream waits for the tail bucket only (the main bucket may still be in flight on the side stream)."""
        if self.cuda and self.live and self.tail is not None and self.tail_done is not None:
            torch.cuda.current_stream().wait_event(self.tail_done)

    def wait(self):
        if self.cuda and self.live:
            torch.cuda.current_stream().wait_stream(self.side)
