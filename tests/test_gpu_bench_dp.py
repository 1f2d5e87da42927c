"""-m gpu: the world > 1 code of bench.py and of the training step, executed on the ONE GPU of this box.

`python bench.py --gpus N` is what the driver times at N = 1, 2, 4, 8 (BASELINE.json: "timesteps/sec at 1/2/4/8 GPUs"); a
one-GPU box cannot measure that curve, but it can RUN every line of it: with CLV_BENCH_SHARE_GPU=1 every rank opens cuda:0
and gloo carries the two gradient buckets through the host, so `measure_train`'s world > 1 branches (state broadcast,
tune_dp_schedule, the all-gathered block times, allreduce_microbench, the dp_schedule object) execute end to end.
The second half holds the data-parallel step to the single-process step and to the oracle at the shape the benchmark runs
per GPU: 2 ranks x (256 x 128) against one process on 512 x 128 -- the semantics of the reference's one fit() on the global
batch (cl_vrnn/train.py:66-71) -- with the coarse and the fine weight-gradient grid.
"""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import clvae_oracle as O
from oracle import philox as OP

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LR = 1e-3


@pytest.fixture(scope="module")
def dev():
    import clvae_amd  # noqa: F401
    from clvae_amd import _lib
    _lib.require_gpu()
    return torch.device("cuda:0")


def clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CLV_BENCH_SHARE_GPU="1", **extra)
    return env


def last_json(text):
    return json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])


def test_bench_with_two_ranks_runs_its_whole_world_gt_1_path(dev):
    """`python bench.py --gpus 2 --steps 5` outside a process group: the launcher starts two ranks (both on cuda:0), and
    rank 0's line carries what only the world > 1 branches produce."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"],
                       capture_output=True, text=True, timeout=900, env=clean_env())
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["metric"] == "piano-roll timesteps/sec (train)" and d["unit"] == "timesteps/s"
    assert d["config"]["global_batch"] == 512 and d["config"]["parallelism"] == "dp2"
    assert "shared_device" in d and d["backend"] == "gloo"
    # value = the timesteps both ranks processed / the slowest rank's time
    assert d["value"] == pytest.approx(2 * 256 * 128 / (d["ms_per_step"] * 1e-3), rel=1e-3)
    assert len(d["ms_per_step_by_rank"]) == 2 and all(t > 0 for t in d["ms_per_step_by_rank"])
    assert max(d["ms_per_step_by_rank"]) == pytest.approx(d["ms_per_step"], rel=1e-3)
    assert len(d["devices"]) == 2 and d["devices"][0].startswith("rank 0") and d["devices"][1].startswith("rank 1")
    sched = d["dp_schedule"]
    tr = sched["wgrad_grid_trials"]
    assert tr["coarse_ms"] > 0 and tr["fine_ms"] > 0 and tr["chosen"] in ("coarse", "fine")
    assert sched["wgrad_split_scale"] == (2 if tr["chosen"] == "fine" else 1)
    assert sched["collectives_per_step"] == 2 and sched["graphs_per_step"] >= 2
    ar = d["allreduce_alone"]
    assert ar["both_buckets_us"] > 0 and ar["tail_bucket_bytes"] == 4 * (128 * 88 * 88 + 88)
    # every parameter's gradient (1,133,462 floats at this shape; tensors start on 16-byte boundaries) + the scratch
    assert 4 * (1133462 + 88) <= ar["tail_bucket_bytes"] + ar["main_bucket_bytes"] <= 4 * (1133462 + 88) + 16 * 16
    assert d["roofline"] is not None and d["roofline"]["frac"] > 0       # rank 0's kernel-time pass ran next to rank 1
    assert d["cpu_baseline"] is None                                     # N = 1 only
    assert np.isfinite(d["final_loss"]) and "also" not in d


@pytest.mark.parametrize("victim", ["0", "1"])
def test_bench_does_not_hang_when_one_rank_dies(dev, victim):
    """One rank raises between the warm-up and the timed region while the other waits in the barrier: the launcher ends
    every rank and returns non-zero, in bounded time."""
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-roofline"], capture_output=True, text=True, timeout=600,
                       env=clean_env(CLV_BENCH_FAULT_RANK=victim))
    assert r.returncode != 0
    assert "injected failure of rank %s" % victim in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 400


def check_params(got, want, steps, rtol=1e-4, atol=2e-5, frac=1e-4, what=""):
    """tests/test_gpu_timed_step.py's rule: an Adam step moves an entry by +-lr whatever its gradient's size, so an entry
    whose gradient is within rounding of zero may differ by 2 lr per step; everything else agrees to rounding."""
    worst = 0.0
    for k in want:
        d = np.abs(got[k] - want[k])
        off = d > rtol * np.abs(want[k]) + atol
        assert off.mean() <= frac, "%s %s: %.2e of the entries beyond rtol %g / atol %g" % (what, k, off.mean(), rtol, atol)
        assert d.max() <= 2 * LR * steps, "%s %s: max |dw| %.2e" % (what, k, d.max())
        worst = max(worst, float(d.max()))
    return worst


@pytest.mark.parametrize("grid", ["coarse", "fine"])
def test_two_ranks_at_the_benchmarks_shape_equal_one_process_and_the_oracle(dev, tmp_path, grid):
    """2 ranks x (256 x 128) = BASELINE config 4's per-GPU shape, three replayed steps through the bound-batch cursor.
    (1) both ranks end with bit-identical parameters; (2) the mean of their loss terms and their parameters equal ONE
    process training on the 512-row global batches (same Philox noise by global row; summation order differs);
    (3) an oracle loop on the global batches follows every loss term to 1e-3 and the parameters by the Adam rule."""
    from dp_worker_full import dataset
    from clvae_amd.engine import VrnnEngine
    from clvae_amd.trainer import TrainStep
    B, T, L, C, steps, nb, dseed, seed = 256, 128, 2, 10, 3, 2, 77, 4321
    G = 2 * B
    out = str(tmp_path / "r%d.npz")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker_full.py"), out, grid]
                                      + [str(v) for v in (B, T, L, C, steps, nb, dseed)], env=env))
    for pr in procs:
        assert pr.wait(timeout=600) == 0
    w0, w1 = dict(np.load(out % 0)), dict(np.load(out % 1))
    j0, j1 = (json.load(open((out % r) + ".json")) for r in range(2))
    for k in w0:
        np.testing.assert_array_equal(w0[k], w1[k], err_msg=k)
    assert j0["graphs"] >= 2 and j0["graphs"] == j1["graphs"]
    dp_losses = [{k: 0.5 * (a[k] + b[k]) for k in a} for a, b in zip(j0["losses"], j1["losses"])]

    # one process on the global batch
    cfg = O.vrnn_config(latent_dim=L, seq_length=T, n_classes=C, use_x_prev=True)
    p = {k: np.asarray(v, dtype=np.float32).astype(np.float64) for k, v in O.vrnn_init_params(cfg, seed=5).items()}
    win, keys = dataset(G, T, C, nb, dseed)
    u8 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.uint8), device=dev)
    cur, hist = u8(win[:, 1:].reshape(nb * G, -1)), u8(win[:, :-1].reshape(nb * G, -1))
    wd = torch.as_tensor(np.ascontiguousarray(keys, dtype=np.float32), device=dev)
    eng = VrnnEngine(cfg, G, dev)
    eng.P.set_weights(p)
    ts = TrainStep(eng, seed=seed)
    ts.bind_batches(cur, hist, wd, idx=None, period=nb, stride=G)
    st = O.adam_wn_init(p)
    for it in range(steps):
        ts.step()
        torch.cuda.synchronize()
        one = eng.losses()
        sl = slice((it % nb) * G, (it % nb + 1) * G)
        eW = OP.normal(G * (C - 1), seed, step=it, stream_id=0).reshape(G, C - 1).astype(np.float32).astype(np.float64)
        eZ = OP.normal(G * T * L, seed, step=it, stream_id=1).reshape(G, T, L).astype(np.float32).astype(np.float64)
        ref = O.vrnn_loss_and_grads(p, cfg, win[sl, 1:].astype(np.float64), win[sl, :-1].astype(np.float64), keys[sl], eW, eZ)
        O.adam_wn_step(p, ref['grads'], st)
        print("%s grid, step %d: total dp %.6f one process %.6f oracle %.6f" % (grid, it, dp_losses[it]['total'], one['total'],
                                                                              ref['total']))
        for k in ('vae', 'kl_z', 'kl_w', 'w_rec', 'total', 'elbo'):
            assert abs(dp_losses[it][k] - one[k]) <= 2e-5 * max(1.0, abs(one[k])), (it, k, dp_losses[it][k], one[k])
            assert abs(dp_losses[it][k] - ref[k]) <= 1e-3, (it, k, dp_losses[it][k], ref[k])
        assert abs(dp_losses[it]['acc'] - one['acc']) < 1e-6
    single = eng.P.get_weights()
    worst_1 = check_params(w0, single, steps, what="dp vs one process")
    worst_o = check_params(w0, p, steps, what="dp vs oracle")
    print("%s grid: max |dw| after %d steps: vs one process %.2e, vs oracle %.2e" % (grid, steps, worst_1, worst_o))
