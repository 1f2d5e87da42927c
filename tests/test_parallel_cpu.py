"""The data-parallel layer on CPU: world_size-2 gloo processes, bucketed gradient averaging,
row sharding and Philox index sharding (the HIP step itself needs a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import clvae_amd  # noqa: F401
from clvae_amd.parallel import GradAllReduce, eps_first_index, shard_rows
from oracle import philox as OP


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, tail_off, tail_n = 1000, 0, 600           # tail bucket first in the flat buffer, like hW/kernel
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)
        ar = GradAllReduce(g, tail_off, tail_n)
        assert ar.world == world and ar.tail.numel() == tail_n and sum(t.numel() for t in ar.main) == n - tail_n
        ar.reduce_main()
        # only the main bucket is averaged so far
        assert torch.allclose(g[tail_n:], torch.arange(tail_n, n, dtype=torch.float32) * 1.5)
        assert torch.allclose(g[:tail_n], torch.arange(tail_n, dtype=torch.float32) * (rank + 1))
        ar.reduce_tail(); ar.wait()
        assert torch.allclose(g, torch.arange(n, dtype=torch.float32) * 1.5)
        # sharding: the union of the ranks' noise equals the single-process draw for the global batch
        B_global, per_row = 8, 9
        r0, r1 = shard_rows(B_global, rank, world)
        mine = OP.normal((r1 - r0) * per_row, 1234, step=3, stream_id=0, first_index=eps_first_index(r0, per_row))
        full = OP.normal(B_global * per_row, 1234, step=3, stream_id=0, first_index=0)
        assert np.array_equal(mine, full[r0 * per_row:r1 * per_row])
        out.put((rank, float(g.sum())))
    finally:
        dist.destroy_process_group()


def test_two_process_gloo_gradient_average_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got[0][1] == got[1][1]


def test_shard_rows_contract():
    assert shard_rows(2048, 3, 8) == (768, 1024)
    with pytest.raises(ValueError):
        shard_rows(10, 0, 4)
    assert eps_first_index(256, 128 * 2) == 65536


def test_bench_launcher_starts_one_rank_per_gpu():
    """`python bench.py --gpus 2` outside a process group starts the two ranks itself (fresh children under
    torch.distributed.run) and rank 0 reports n_gpus = 2; here through the CPU self-test path (gloo, no GPU)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launch"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] == 3.0
    assert sorted(tuple(x) for x in d["ranks"]) == [(0, 0), (1, 1)]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launch"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE 1 != --gpus 2" in (r.stderr + r.stdout)


def test_bench_shared_gpu_mode_puts_every_rank_on_device_0():
    """CLV_BENCH_SHARE_GPU=1 (the one-GPU box's way to run `bench.py --gpus N`, tests/test_gpu_bench_dp.py): every rank gets
    local device 0 and the group is gloo; here through the CPU self-test path."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CLV_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launch"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and sorted(tuple(x) for x in d["ranks"]) == [(0, 0), (1, 0)] and d["backend"] == "gloo"


def test_pmc_passes_become_bytes_per_launch_and_per_step(tmp_path):
    """bench.aggregate_pmc: two rocprofv3 counter CSVs (FETCH_SIZE, WRITE_SIZE; one row per dispatch and counter instance) ->
    HBM bytes with the gfx950 corrections (KB -> bytes, FETCH_SIZE x 2), the dominant kernels' average per launch, the step's
    total from the launch count of the once-per-step kernel."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    head = "Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n"
    fwd, bwd, red = ("void clv::lstm_pair_fwd_kernel<0, true, 1, false>(clv::PairFwdArgs)",
                     "void clv::lstm_pair_bwd_kernel<0, 4, true>(clv::PairBwdArgs)", "clv::splitk_reduce_multi_kernel(clv::ReduceTable)")
    rows_f, rows_w = [], []
    for step in range(3):
        for j, (k, fkb, wkb) in enumerate(((fwd, 100.0, 400.0), (bwd, 300.0, 200.0), (red, 50.0, 1.0))):
            d = 10 * step + j
            rows_f += ['%d,"%s",FETCH_SIZE,%g' % (d, k, fkb / 2)] * 2          # two counter instances per dispatch: summed
            rows_w += ['%d,"%s",WRITE_SIZE,%g' % (d, k, wkb)]
    (tmp_path / "f.csv").write_text(head + "\n".join(rows_f) + "\n")
    (tmp_path / "w.csv").write_text(head + "\n".join(rows_w) + "\n")
    d = bench.aggregate_pmc("cfg3", str(tmp_path / "f.csv"), str(tmp_path / "w.csv"))
    assert d["steps_profiled"] == 3 and len(d["dominant_kernels"]) == 2
    per = {k["kernel"].split("_kernel")[0]: k for k in d["kernels"]}
    assert per["lstm_pair_fwd"]["hbm_read_bytes_corrected"] == 2 * 1024 * 100.0 and per["lstm_pair_fwd"]["hbm_write_bytes"] == 1024 * 400.0
    assert d["dominant_bytes_per_launch"] == ((200 + 400) + (600 + 200)) * 1024 / 2
    assert d["step_bytes"] == ((200 + 400) + (600 + 200) + (100 + 1)) * 1024
