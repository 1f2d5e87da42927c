#!/usr/bin/env python
"""bench.py -- piano-roll timesteps/sec (train) of the cl_vrnn hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N>1 launched by
torch.distributed.run, one rank per GPU over RCCL).  A "step" is one full optimisation step
(Philox noise draw, forward, 4 losses, BPTT, gradient all-reduce when N>1, Adam-with-weight-norm)
over one batch of synthetic 88-dim piano-roll windows already resident in HBM.  Rank 0 prints ONE
JSON line.  Workloads (BASELINE.json configs):
  cfg3 (default)  cl_vrnn, 256 windows/GPU x seq_len 128, latent 2, 10 classes   (configs[2]; at N GPUs = configs[3])
  cfg5            cl_vrnn, 1024 windows/GPU x seq_len 256, latent 32, 10 classes  (configs[4])
  cfg2            cl_vae --use_x_prev, batch 512/GPU, latent 4, 2 classes (fp32 path)
Weak scaling: per-GPU work is fixed, the global batch grows with N.

The timed region is exactly K steps between a barrier + torch.cuda.synchronize() on both sides.  When K steps take less
than 50 ms (the driver's K = 20 at 0.4 ms is 8 ms: one clock-state hiccup of the device moves it by 5-10 %), the region is
repeated as 5..9 such blocks and the line carries the MEDIAN block (`timed_blocks`: every block, its min and max).
A default single-GPU run (`python bench.py`, workload cfg3) appends `also`: the other BASELINE workloads -- cfg5, cfg2
fp32 / bf16, generation of 1 and 1024 sequences -- measured right after the headline by a child process (`bench.py
--also-only`; a few seconds each: value, ms per step, dominant-kernel fraction), so that those numbers are witnessed by
whoever runs the line and nothing an extra does can cost the headline (it is on stderr before the child starts).
CLV_BENCH_SHARE_GPU=1 puts every rank of `--gpus N` on cuda:0 with gloo between them: the whole world > 1 code path on a
one-GPU box (tests/test_gpu_bench_dp.py), never a scaling measurement (the line says `shared_device`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    'cfg3': dict(model='cl_vrnn', B=256, T=128, L=2, C=10),
    'cfg5': dict(model='cl_vrnn', B=1024, T=256, L=32, C=10),
    'cfg2': dict(model='cl_vae', B=512, T=1, L=4, C=2),
    # the reference CLIs' own defaults (cl_vae/train.py: batch 100 = BASELINE config 1; cl_vrnn/train.py: batch 200 x seq_length 16)
    'cfg1': dict(model='cl_vae', B=100, T=1, L=4, C=2),
    'cli_vrnn': dict(model='cl_vrnn', B=200, T=16, L=2, C=10),
    # BASELINE config 5, second half: autoregressive generation under hipGraph (frames/s, N seeds at once)
    'gen1024': dict(model='cl_vrnn', B=1024, T=256, L=32, C=10, generate=True),
    'gen1': dict(model='cl_vrnn', B=1, T=256, L=32, C=10, generate=True),
    # cl_vae generation (cl_vae/model.py:9-42): N seeds at once, one persistent kernel (csrc/vae_generate.hip)
    'gen_vae1024': dict(model='cl_vae', B=1024, T=1, L=4, C=2, generate=True),
    'gen_vae1': dict(model='cl_vae', B=1, T=1, L=4, C=2, generate=True),
}
PEAK_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 matrix == fp32 vector peak
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA peak (never the 2:1-sparsity figure)
PROFILE_TAGS = ('r05_f', 'r05_e', 'r05_d', 'r05_c', 'r05_b', 'r05_a', 'r04_f', 'r04_e', 'r04_d', 'r04_c', 'r04_b', 'r04_a', 'r03_f')      # committed profile sets (profiles/<tag>_*), newest first
NOTE_DENSITY = 0.0443        # measured JSB note density


def flop_per_timestep(w):
    """Algorithmic GEMM flops per training timestep (2 flop/MAC, train = 3x forward), SURVEY.md 8(d)."""
    L, C, T = w['L'], w['C'], w['T']
    if w['model'] == 'cl_vae':
        macs = 88 * 88 + 2 * 88 * (C - 1) + (88 + C) * 88 + 2 * 88 * L + (C + 88 + L) * 88 + 88 * 88
    else:
        macs = (88 + C + 88) * 352 + 2 * 88 * L + (88 + L + C + 88) * 352 + 88 * 88 + 88 * 88 + 88 * 2 * (C - 1) / T
    return 6.0 * macs


def lstm_seq_flops(w, B):
    """Algorithmic flops of ONE persistent LSTM sequence launch (recurrent product only): 2*B*T*88*352."""
    return 2.0 * B * w['T'] * 88 * 352


def make_engine(w, dev, bf16=False):
    from clvae_amd.engine import VaeEngine, VrnnEngine
    from clvae_amd.initializers import init_weights
    if w['model'] == 'cl_vrnn':
        cfg = dict(D=88, H=88, L=w['L'], T=w['T'], C=w['C'], use_x_prev=True, class_weight=1.0, kl_weight=1.0,
                   w_kl_weight=1.0, w_log_var_prior=0.0, gate_act='hard_sigmoid')
        eng = VrnnEngine(cfg, w['B'], dev)
    else:
        cfg = dict(D=88, H=88, L=w['L'], Hc=88, C=w['C'], use_x_prev=True, class_weight=1.0, kl_weight=1.0,
                   w_kl_weight=1.0, w_log_var_prior=0.0, bf16=bool(bf16))
        eng = VaeEngine(cfg, w['B'], dev)
    eng.P.set_weights(init_weights(eng.P.logical, cfg, seed=0))
    return eng, cfg


def synthetic_windows(w, n, seed, dev):
    """Bernoulli(0.0443) piano-roll windows + labels, resident on the device the way Model.fit keeps a data set:
    frames as uint8 (the batch gather converts to float), labels as float32."""
    import torch
    rng = np.random.default_rng(seed)
    T = w['T']
    win = (rng.random((n, T + 1, 88)) < NOTE_DENSITY)
    keys = rng.integers(0, w['C'], n)
    onehot = np.eye(w['C'], dtype=np.float32)[keys]
    wt = torch.as_tensor(win.astype(np.uint8), device=dev)
    if w['model'] == 'cl_vae':
        return wt[:, 1].contiguous(), wt[:, 0].contiguous(), torch.as_tensor(onehot, device=dev)
    return wt[:, 1:].contiguous(), wt[:, :-1].contiguous(), torch.as_tensor(onehot, device=dev)


def cpu_model_name():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(w, seconds=20.0):
    """the CPU restatement of the reference math (oracle/torch_cpu.py: fp32, torch-CPU, per-timestep
    LSTM loop like Keras' K.rnn, autograd backward, Adam with weight norm) timed on this box's host cores at the SAME
    batch and sequence length as the GPU run: >= 3 warm-up steps, then >= 10 timed steps (about `seconds` of work),
    median step.  "CPU restatement (Keras-equivalent math), not Keras"."""
    import torch
    from oracle import clvae_oracle as O
    from oracle import torch_cpu as TC
    T, L, C, B = w['T'], w['L'], w['C'], w['B']
    if w['model'] == 'cl_vrnn':
        cfg = O.vrnn_config(latent_dim=L, seq_length=T, n_classes=C, use_x_prev=True)
    else:
        cfg = O.vae_config(latent_dim=L, n_classes=C, use_x_prev=True)
    # torch's default of one thread per hardware thread oversubscribes these small matmuls on a many-core host (the
    # 256-thread GPU box ran 8x slower with 128 threads than an 8-core container): try a few pool sizes on two steps each
    # and time the fastest
    best = None
    for n in sorted({n for n in (8, 16, 32, 64, torch.get_num_threads()) if n <= (os.cpu_count() or 1)}):
        torch.set_num_threads(n)
        probe = TC.time_training_steps(w['model'], cfg, B, T, seconds=0.0, min_steps=2, warmup=1)
        if best is None or probe['timesteps_per_s'] > best[1]:
            best = (n, probe['timesteps_per_s'])
    torch.set_num_threads(best[0])
    r = TC.time_training_steps(w['model'], cfg, B, T, seconds=seconds)
    return dict(value=r['timesteps_per_s'], unit="timesteps/s", cores=int(r['threads']), kind="port",
                sample="%d timed steps (median) of batch %d x seq_len %d after 3 warm-up steps; CPU restatement "
                       "(Keras-equivalent math, torch-CPU fp32, per-timestep LSTM loop), not Keras; %s, os.cpu_count() "
                       "= %d, torch.get_num_threads() = %d"
                       % (r['steps'], B, T, cpu_model_name(), os.cpu_count(), torch.get_num_threads()))


# ---- HBM traffic of the step, measured: rocprofv3 PMC passes over a short eager run of this script -----------------------
DOMINANT_RE = r'lstm_(pair_|mx_)?(fwd|bwd)(_mfma)?_kernel|vae_fused'      # the kernels the roofline object times


def aggregate_pmc(workload, fetch_csv, write_csv, steps_fallback=9):
    """Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> HBM bytes per kernel launch and per step, with the gfx950
    corrections of MI355X_MICROARCH.md: both counters are in KB (x 1024); FETCH_SIZE reports half of the bytes read (x 2)."""
    import collections
    import csv
    import re

    def per_kernel(path, counter):
        disp, name = collections.defaultdict(float), {}
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] != counter:
                continue
            disp[r['Dispatch_Id']] += float(r['Counter_Value'])
            name[r['Dispatch_Id']] = r['Kernel_Name']
        agg = collections.defaultdict(list)
        for d, v in disp.items():
            agg[name[d]].append(v)
        return agg

    f, w = per_kernel(fetch_csv, 'FETCH_SIZE'), per_kernel(write_csv, 'WRITE_SIZE')
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(sum(f.get(k, [0])) * 2 + sum(w.get(k, [0])))):
        fk, wk = f.get(k, [0.0]), w.get(k, [0.0])
        short = re.sub(r'^void ', '', k).replace('clv::', '')
        rows.append(dict(kernel=short[:120], launches=len(fk), fetch_size_kb_avg=sum(fk) / len(fk),
                         write_size_kb_avg=sum(wk) / len(wk), hbm_read_bytes_corrected=2 * 1024 * sum(fk) / len(fk),
                         hbm_write_bytes=1024 * sum(wk) / len(wk)))
    dom = [r for r in rows if re.match(DOMINANT_RE, r['kernel'])]
    per_launch = sum(r['hbm_read_bytes_corrected'] + r['hbm_write_bytes'] for r in dom) / max(len(dom), 1)
    # steps the profiled run executed (set-up + warm-up + timed): the launch count of a kernel that runs once per LSTM pass
    per_step = {'lstm_pair_fwd_kernel': 1, 'lstm_mx_fwd_kernel': 2, 'lstm_fwd_mfma_kernel': 2, 'vae_fused_kernel': 1}
    steps = None
    for r in rows:
        for key, n in per_step.items():
            if steps is None and r['kernel'].startswith(key):
                steps = sum(x['launches'] for x in rows if x['kernel'].startswith(key)) // n
    steps = steps or steps_fallback
    step_bytes = sum((r['hbm_read_bytes_corrected'] + r['hbm_write_bytes']) * r['launches'] for r in rows) / steps
    return dict(note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --no-graph); FETCH_SIZE x2 "
                     "(gfx950 reports half of the bytes read), both KB -> bytes x1024 (MI355X_MICROARCH.md)",
                workload=workload, dominant_kernels=[r['kernel'] for r in dom], dominant_bytes_per_launch=per_launch,
                step_bytes=step_bytes, steps_profiled=steps, kernels=rows)


def aggregate_sq(sq_csv):
    """One rocprofv3 SQ pass -> per kernel (average over its dispatches): mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES / 4
    (four matrix pipes per CU), valu_issue = 2 SQ_INSTS_VALU / (4 SQ_BUSY_CU_CYCLES) (a wave64 vector instruction holds a SIMD-32 for
    two cycles), waves_parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES (tools/pmc_sq.py, MI355X_MICROARCH.md units)."""
    import collections
    import csv
    import re
    per, name = collections.defaultdict(lambda: collections.defaultdict(float)), {}
    for r in csv.DictReader(open(sq_csv)):
        per[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
        name[r['Dispatch_Id']] = r['Kernel_Name']
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d_, cs in per.items():
        k = re.sub(r'\(.*', '', re.sub(r'^void ', '', name[d_]).replace('clv::', ''))[:80]
        for c, v in cs.items():
            acc[k][c].append(v)
    out = {}
    for k, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        busy = m.get('SQ_BUSY_CU_CYCLES')
        if not busy:
            continue
        out[k] = dict(mfma_busy=round(m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / busy / 4, 4),
                      valu_issue=round(2 * m.get('SQ_INSTS_VALU', 0.0) / (4 * busy), 4),
                      waves_parked=round(m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], 4) if m.get('SQ_WAVE_CYCLES') else None)
    return out


def measure_pmc_traffic(workload, bf16=False, timeout_s=(150, 60, 60)):
    """The two PMC passes as CHILD processes of this one -- `rocprofv3 --pmc <counter> --kernel-trace -- python3 bench.py
    --workload W --steps 6 --warmup 3 --no-graph ...` -- started before this process has touched the GPU (nothing is exec'd
    over a GPU-initialised process; the program after `--` is python3 itself; a PMC pass carries no other trace domain).
    Returns aggregate_pmc()'s dict, or {'error': ...}: the caller then falls back to the committed summary.
    timeout_s: per pass; the first one may be the box's first `import torch` (1-2 minutes on a fresh image), the others take
    seconds -- a hung profiler costs the headline at most 4.5 minutes, and the first failure ends the passes."""
    import shutil
    import signal
    import subprocess
    import tempfile
    rp = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(rp):
        return dict(error="rocprofv3 not found")
    tmp = tempfile.mkdtemp(prefix='clv_pmc_')
    try:
        csvs = {}
        # three passes: the two traffic counters each alone (the guide's rule), then the SQ counters of the busy fractions
        for ipass, (counter, pmc_list) in enumerate((('FETCH_SIZE', ['FETCH_SIZE']), ('WRITE_SIZE', ['WRITE_SIZE']),
                                                     ('SQ', ['SQ_BUSY_CU_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_VALU', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY']))):
            limit = timeout_s[ipass]
            out = os.path.join(tmp, counter)
            cmd = [rp, '--pmc'] + pmc_list + ['--kernel-trace', '-d', out, '-o', 't', '--output-format', 'csv', '--',
                   sys.executable, os.path.abspath(__file__), '--workload', workload, '--steps', '6', '--warmup', '3',
                   '--no-cpu-baseline', '--no-roofline', '--no-graph', '--no-also', '--no-pmc-traffic'] + (['--bf16'] if bf16 else [])
            env = dict(os.environ, TMPDIR='/tmp')
            pr = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                  start_new_session=True)
            try:
                so, se = pr.communicate(timeout=limit)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.communicate()
                if counter == 'SQ':
                    break
                return dict(error="the %s pass passed %d s and was killed" % (counter, limit))
            found = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith('counter_collection.csv')]
            if pr.returncode != 0 or not found:
                if counter == 'SQ':      # the busy fractions are an extra: the traffic stands without them
                    break
                return dict(error="the %s pass ended with code %s; stderr tail: %s" % (counter, pr.returncode, se[-200:]))
            csvs[counter] = found[0]
        d = aggregate_pmc(workload, csvs['FETCH_SIZE'], csvs['WRITE_SIZE'])
        if 'SQ' in csvs:
            d['sq'] = aggregate_sq(csvs['SQ'])
        return d
    except Exception as ex:      # noqa: BLE001 -- a measurement extra must never cost the run
        return dict(error=repr(ex)[:200])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def sources_sha16():
    """sha256 over the kernel sources (csrc/*.hip, *.h, include/clvae.h) of THIS tree: a committed PMC summary carries the
    value of the tree it was measured on (tools/round_profile.sh), so a stale one shows."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, 'classifying-vae-lstm_amd', 'csrc', '*.h*')) + [os.path.join(ROOT, 'include', 'clvae.h')]):
        h.update(os.path.basename(path).encode())
        h.update(open(path, 'rb').read())
    return h.hexdigest()[:16]


def bench_generate(args, w, dev, rank, world):
    """Replicas-only sampling benchmark: N seeds per GPU, 16 teacher-forced frames, then free-running frames;
    a "step" is one generated frame for all N sequences (one hipGraph replay)."""
    import torch
    from clvae_amd.engine import VaeEngine, VrnnEngine
    from clvae_amd.initializers import init_weights
    N, L, C = w['B'], w['L'], w['C']
    if w['model'] == 'cl_vae':
        return bench_generate_vae(args, w, dev, rank, world)
    cfg = dict(D=88, H=88, L=L, T=16, C=C, use_x_prev=True, class_weight=1.0, kl_weight=1.0, w_kl_weight=1.0,
               w_log_var_prior=0.0, gate_act='hard_sigmoid')
    eng = VrnnEngine(cfg, 1, dev)
    wts = init_weights(eng.P.logical, cfg, seed=0)
    # random-init weights put every note at p ~ 0.5; a trained model emits piano-roll density (0.0443),
    # which is what the sparse frame handling sees in practice: bias the output head to logit(0.0443)
    wts['X_decoded_mean/bias'] = np.full_like(wts['X_decoded_mean/bias'], float(np.log(NOTE_DENSITY / (1 - NOTE_DENSITY))))
    eng.P.set_weights(wts)
    rng = np.random.default_rng(1234 + rank)
    seeds = torch.as_tensor((rng.random((N, 16, 88)) < NOTE_DENSITY).astype(np.float32), device=dev)
    wv = torch.as_tensor(np.eye(C, dtype=np.float32)[rng.integers(0, C, N)], device=dev)
    persistent = not args.no_persistent
    eng.generate(seeds, wv, max(args.warmup, 2), seed=1, persistent=persistent)
    torch.cuda.synchronize()
    dt = None
    for rep in range(3):                 # best of 3: a run is a few ms and the box shows sporadic ~50 ms stalls
        t0 = time.perf_counter()
        out = eng.generate(seeds, wv, args.steps, seed=2 + rep, persistent=persistent)
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t0
        dt = d1 if dt is None else min(dt, d1)
    frames = world * N * (args.steps + 16)
    if rank == 0:
        return ({"metric": "generated piano-roll frames/sec (sample)", "value": round(frames / dt, 1),
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * dt / (args.steps + 16), 4), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "%s: cl_vrnn generation, %d seeds/GPU, latent %d, 16 seed frames, %s, "
                                                 "output bias = logit(%.4f), best of 3 runs"
                                                 % (args.workload, N, L, "one persistent kernel (workgroup per sequence)"
                                                    if persistent else "hipGraph replay per frame", NOTE_DENSITY),
                                     "parallelism": "replicas%d" % world},
                          "note_density_out": round(float(out.mean().item()), 4)})
    return None


def allreduce_microbench(ts, dev, iters=50):
    """The step's two gradient buckets alone (N > 1): the hW-kernel bucket and the rest, issued back to back on the side
    stream like the step issues them, nothing else on the GPU.  Tells a reader of the first multi-GPU run how much of a
    step the collectives need when nothing hides them."""
    import torch
    import torch.distributed as dist
    ar = ts.ar
    if ar is None:
        return None
    tail_b = 4 * (ar.tail.numel() if ar.tail is not None else 0)
    main_b = 4 * sum(t.numel() for t in ar.main)
    keep = ar.flat.clone()
    for _ in range(5):
        ar.reduce_tail(); ar.reduce_main(); ar.wait()
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        ar.reduce_tail(); ar.reduce_main(); ar.wait()
    torch.cuda.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / iters
    from clvae_amd.parallel import meta_device
    tt = torch.tensor([us], dtype=torch.float64, device=meta_device(dev))
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ar.flat.copy_(keep)
    return dict(tail_bucket_bytes=tail_b, main_bucket_bytes=main_b, both_buckets_us=round(float(tt.item()), 2),
                note="two all-reduces (AVG) per step on a side stream; measured alone, max over ranks, %d iterations" % iters)


def bench_generate_vae(args, w, dev, rank, world):
    """cl_vae sampling: N seed frames per GPU, `steps` generated frames each (cl_vae/model.py:9-42); a "step" is one
    frame for all N sequences."""
    import torch
    from clvae_amd.engine import VaeEngine
    from clvae_amd.initializers import init_weights
    N, L, C = w['B'], w['L'], w['C']
    cfg = dict(D=88, H=88, L=L, Hc=88, C=C, use_x_prev=True, class_weight=1.0, kl_weight=1.0, w_kl_weight=1.0,
               w_log_var_prior=0.0)
    persistent = not args.no_persistent
    eng = VaeEngine(cfg, N if not persistent else 4, dev)
    wts = init_weights(eng.P.logical, cfg, seed=0)
    wts['x_decoded_mean/bias'] = np.full_like(wts['x_decoded_mean/bias'], float(np.log(NOTE_DENSITY / (1 - NOTE_DENSITY))))
    eng.P.set_weights(wts)
    rng = np.random.default_rng(1234 + rank)
    seeds = torch.as_tensor((rng.random((N, 88)) < NOTE_DENSITY).astype(np.float32), device=dev)
    wv = torch.as_tensor(np.eye(C, dtype=np.float32)[rng.integers(0, C, N)], device=dev)
    eng.generate(seeds, wv, max(args.warmup, 2), seed=1, persistent=persistent)
    torch.cuda.synchronize()
    dt = None
    for rep in range(3):
        t0 = time.perf_counter()
        out = eng.generate(seeds, wv, args.steps, seed=2 + rep, persistent=persistent)
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t0
        dt = d1 if dt is None else min(dt, d1)
    if rank == 0:
        return ({"metric": "generated piano-roll frames/sec (sample)", "value": round(world * N * args.steps / dt, 1),
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "%s: cl_vae generation, %d seeds/GPU, latent %d, %s, output bias = "
                                                 "logit(%.4f), best of 3 runs"
                                                 % (args.workload, N, L, "one persistent kernel (workgroup per sequence)"
                                                    if persistent else "hipGraph replay per frame", NOTE_DENSITY),
                                     "parallelism": "replicas%d" % world},
                          "note_density_out": round(float(out.mean().item()), 4)})
    return None


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` outside a process group: start the N ranks (one per GPU) as fresh children under
    torch.distributed.run and return their exit code.  Called before anything has touched the GPU in this process; the
    parent only waits (it never execs)."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC (RCCL across processes)
    env.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(cmd, env=env)


def launch_selftest(args):
    """--selftest-launch: the launcher, the rendezvous and the rank-0 JSON line without a GPU (gloo); used by
    tests/test_parallel_cpu.py.  No timing is reported."""
    import torch
    import torch.distributed as dist
    from clvae_amd.parallel import init_from_env
    rank, local, world = init_from_env(backend='gloo')
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    t = torch.tensor([float(rank + 1)])
    ranks = [None] * world
    if world > 1:
        dist.all_reduce(t)
        dist.all_gather_object(ranks, (rank, local))
    else:
        ranks = [(rank, local)]
    if rank == 0:
        print(json.dumps({"metric": "launch selftest", "value": float(t.item()), "unit": "sum of (rank+1)",
                          "n_gpus": world, "ranks": ranks, "backend": "gloo"}))
    if world > 1:
        dist.destroy_process_group()


def timed_blocks(run, barrier, steps, world, dev):
    """K = `steps` steps between barrier + synchronize, as 1 block, or 5..9 blocks when a block is shorter than 50 ms.
    Returns (seconds of the median block, [ms per step of every block], per-rank ms per step of the median block)."""
    import torch
    import torch.distributed as dist
    from clvae_amd.parallel import meta_device
    mdev = meta_device(dev)

    def one():
        barrier()
        t0 = time.perf_counter()
        run(steps)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            mine = torch.tensor([dt], dtype=torch.float64, device=mdev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            return [float(t.item()) for t in every]
        return [dt]

    blocks = [one()]
    first = max(blocks[0])
    if first < 0.05:
        n = int(min(9, max(5, np.ceil(0.05 / max(first, 1e-6)))))
        if world > 1:                      # every rank must run the same number of blocks
            t = torch.tensor([n], dtype=torch.int32, device=mdev)
            dist.broadcast(t, src=0)
            n = int(t.item())
        blocks += [one() for _ in range(n - 1)]
    worst = [max(b) for b in blocks]                         # a block takes as long as its slowest rank
    mid = int(np.argsort(worst)[len(worst) // 2])
    return worst[mid], [round(1e3 * x / steps, 4) for x in worst], [round(1e3 * x / steps, 4) for x in blocks[mid]]


def kernel_time_pass(eng, ts_args, batch, reps, label_off=False):
    """Per-kernel HIP-event times of `reps` eager steps of the same shapes on the launch stream; parameters and optimizer
    state are put back afterwards.  label_off: the label path's backward as a launch of its own (what the pair backward
    kernel costs WITHOUT that epilogue)."""
    import torch
    from clvae_amd import ops
    from clvae_amd.trainer import TrainStep
    saved = [t_.clone() for t_ in eng.P.state_tensors()]
    keep = getattr(eng, 'label_in_pair', None)
    if label_off:
        eng.label_in_pair = False
    ts_e = TrainStep(eng, use_graph=False, **ts_args)
    ts_e._drop_logits = hasattr(eng, 'keep_logits')      # the launches of the REPLAYED step: it stores no logits (TrainStep._main)
    # fed like the timed step (the bound-batch cursor): the launches -- mini-batch assembly inside the label / fused launch, frames
    # read as bytes -- are then those of the replayed graph, not those of a float batch staged once
    ts_e.bind_batches(batch[0], batch[1], batch[2], idx=None, period=1, stride=int(batch[0].shape[0]))
    ts_e.step(); torch.cuda.synchronize()
    ops.prof_enable(True)
    for _ in range(reps):
        ts_e.step()
        ops.prof_empty_scope()          # what a bracket of the profiler's events costs by itself ("event_pair")
    recs = ops.prof_collect()
    ops.prof_enable(False)
    for t_, sv in zip(eng.P.state_tensors(), saved):
        t_.copy_(sv)
    if label_off:
        eng.label_in_pair = keep
    del ts_e, saved
    return recs


def roofline_object(args, w, wl_name, eng, recs, recs_nolabel, reps, value_per_gpu, bf16, kernel_times=False, pmc=None):
    B = w['B']
    pair_ms = [r[2] / max(r[1], 1) for r in recs if r[0] == 'event_pair']
    event_pair_us = 1e3 * pair_ms[0] if pair_ms else 0.0
    recs = [r for r in recs if r[0] != 'event_pair']
    if kernel_times:
        print("  event_pair (empty bracket)  %.2f us" % event_pair_us, file=sys.stderr)
        tot = sum(r[2] for r in recs)
        for name, n, ms in sorted(recs, key=lambda r: -r[2]):
            print("  %-22s launches/step %5.1f  ms/step %8.4f  %5.1f%%" % (name, n / reps, ms / reps, 100 * ms / tot),
                  file=sys.stderr)

    def table(rs):
        by = {}
        for r in rs:
            if r[0] == 'event_pair':
                continue
            k = "gemm_f32" if r[0].startswith("gemm ") else r[0]
            by[k] = (k, by.get(k, (k, 0, 0.0))[1] + r[1], by.get(k, (k, 0, 0.0))[2] + r[2])
        return by
    by = table(recs)
    if w['model'] == 'cl_vrnn':
        # dominant kernels: the persistent LSTM sequence kernels (one launch = one LSTM pass, or both LSTMs of a pass when
        # the pair kernels run): 4 LSTM passes per step
        names = sorted(k for k in by if k.startswith(('lstm_pair_', 'lstm_seq_', 'lstm_mx_')) and not k.endswith('_pack'))
        n = sum(by[k][1] for k in names)
        ms_raw = sum(by[k][2] for k in names)
        ms = ms_raw - n * event_pair_us * 1e-3         # minus what the brackets themselves cost (see `event_pair_us`)
        avg_s = ms / n * 1e-3
        achieved = 4 * reps * lstm_seq_flops(w, B) / (ms * 1e-3) / 1e12
        kname = '+'.join(names)
    else:
        # cl_vae: the whole step is one fused launch (all 8 Dense layers, forward + backward)
        kname = 'vae_fused_step' if 'vae_fused_step' in by else 'gemm_f32'
        n, ms_raw = by[kname][1], by[kname][2]
        ms = ms_raw - n * event_pair_us * 1e-3
        avg_s = ms / n * 1e-3
        achieved = (flop_per_timestep(w) * B / (n / reps)) / avg_s / 1e12
    # HBM bytes per launch of the dominant kernels and per step: MEASURED in this run when main() made the two PMC passes
    # (measure_pmc_traffic: the default single-GPU run does), else read from the newest committed PMC summary of the SAME
    # workload (tools/round_profile.sh -> profiles/<tag>_pmc_traffic_<workload>.json).  Either way the summary's dominant
    # kernels must be the kernels this run just timed (same names, as many instances), or `traffic` stays null; a
    # committed summary also says whether it was measured on this tree's kernel sources (`traffic_sources_match`).
    import re
    traffic = step_traffic = traffic_source = traffic_match = traffic_problem = None
    measured = bool(pmc is not None and 'error' not in pmc)
    wl_tag = wl_name + ('_bf16' if bf16 else '')

    def names_agree(pm):
        timed = sorted(kname.split('+'))                                         # e.g. lstm_pair_bwd, lstm_pair_fwd
        prof = sorted(set(re.sub(r'_kernel.*', '', k) for k in pm.get('dominant_kernels', [])))
        if w['model'] != 'cl_vrnn':
            return any(k.startswith('vae_fused') for k in pm.get('dominant_kernels', [])) == (kname == 'vae_fused_step')
        return timed == prof

    cands = [('measured in this run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two child passes of bench.py --no-graph)', pmc)] if measured else []
    for tag in PROFILE_TAGS:
        path = os.path.join(ROOT, 'profiles', '%s_pmc_traffic_%s.json' % (tag, wl_tag))
        try:
            cands.append((os.path.relpath(path, ROOT), json.load(open(path))))
        except Exception:
            continue
    for src, pm in cands:
        if pm.get('workload') != wl_name:
            continue
        if not names_agree(pm):
            traffic_problem = "%s: its dominant kernels %s are not the kernels timed here (%s)" % (src, pm.get('dominant_kernels'), kname)
            continue
        if w['model'] == 'cl_vrnn':
            traffic = round(pm.get('dominant_bytes_per_launch', pm.get('lstm_seq_bytes_per_launch')))
        step_traffic = pm.get('step_bytes')
        traffic_source = src
        traffic_match = True if pm is pmc else (pm.get('sources_sha16') == sources_sha16() if 'sources_sha16' in pm else None)
        break
    roofline = dict(bound="mfma", achieved=round(achieved, 3), peak=PEAK_F32_TFLOPS, unit="TFLOP/s",
                    frac=round(achieved / PEAK_F32_TFLOPS, 4), traffic=traffic, kernel=kname,
                    avg_launch_us=round(avg_s * 1e6, 2), avg_launch_us_raw=round(ms_raw / n * 1e3, 2),
                    event_pair_us=round(event_pair_us, 2),
                    timing="HIP events around every launch on the launch stream; avg_launch_us = the bracketed time minus "
                           "event_pair_us, an EMPTY bracket measured in the same pass (the two event records cost that much "
                           "by themselves); achieved / frac use avg_launch_us, the raw figure is avg_launch_us_raw",
                    algorithmic_step_frac=round(value_per_gpu * flop_per_timestep(w) / 1e12 / PEAK_F32_TFLOPS, 4),
                    algorithmic_step_frac_is="whole-step ALGORITHMIC flop rate (SURVEY.md 8d: dense-equivalent GEMM flops, which the "
                                             "sparse input kernels never execute) / the fp32 peak: a throughput figure, not a utilisation",
                    step_traffic=step_traffic, traffic_source=traffic_source,
                    traffic_measured_in_this_run=bool(measured and traffic_source is not None and traffic_source.startswith('measured')),
                    traffic_sources_match=traffic_match, traffic_rejected=traffic_problem,
                    traffic_pass_error=(pmc or {}).get('error') if pmc is not None else None,
                    counters=({kk: v for kk, v in pmc['sq'].items() if re.match(DOMINANT_RE + '|lstm_wgrad_bf16|out_head|vrnn_label_fwd|sparse_proj', kk)}
                              if (measured and pmc.get('sq')) else None),
                    counters_are="SQ counters of the same child passes, per kernel: mfma_busy (of the four matrix pipes), valu_issue "
                                 "(of the SIMDs' vector issue cycles), waves_parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES)",
                    kernel_time_pass="%d eager steps with HIP events around every launch, after the graph capture and before the warm-up" % reps)
    if w['model'] == 'cl_vrnn':
        if getattr(eng, 'use_mx', False):
            roofline['executes_on'] = ("bf16 MFMA (v_mfma_f32_16x16x32_bf16), exact fp32 products from 3 x 3 bf16 pieces, fp32 "
                                       "accumulate; priced against the fp32 peak because the ARITHMETIC is fp32")
        else:
            # one batch row per CU leaves the matrix cores' M dimension empty; the fp32 vector peak equals the fp32 MFMA peak
            roofline['executes_on'] = "fp32 VALU (v_pk_fma_f32): the fp32 vector peak equals the fp32 MFMA peak (157.3 TFLOP/s)"
        if recs_nolabel is not None:
            # honest denominator: the pair backward launch is not only recurrent products -- measured in THIS run by a
            # second kernel-time pass with the label path's backward as a launch of its own
            alt = table(recs_nolabel)
            ep = [r[2] / max(r[1], 1) for r in recs_nolabel if r[0] == 'event_pair']
            ep_us = 1e3 * ep[0] if ep else event_pair_us
            us = lambda t, k: round(1e3 * t[k][2] / t[k][1] - ep_us, 2) if k in t else None
            roofline['kernel_also_runs'] = dict(
                what="lstm_pair_bwd ends with the label path's backward of every batch row (no launch of its own)",
                lstm_pair_bwd_with_it_us=round(1e3 * by['lstm_pair_bwd'][2] / by['lstm_pair_bwd'][1] - event_pair_us, 2),
                lstm_pair_bwd_without_it_us=us(alt, 'lstm_pair_bwd'), label_bwd_own_launch_us=us(alt, 'vrnn_label_bwd'),
                measured="this run: a second pass of %d eager steps with CLV_LABEL_IN_PAIR off" % max(reps // 2, 4))
    wkeys = [k for k in by if k.startswith('lstm_wgrad_bf16')]      # one launch per LSTM, or both in one (..._pair)
    if wkeys:
        # the batched gate GEMM of the north star: every kernel gradient of an LSTM, [x | h | z]^T . dz over B*T rows
        # (csrc/wgrad_bf16.hip), formed on the BF16 matrix cores from bf16 pieces (6 of 9 piece pairs).  Reported against the pipe it
        # runs on: (a) issued bf16 MFMA flops / 2.5 PFLOP/s, (b) the MFMA-busy counter of the committed SQ pass;
        # the algorithmic fp32-equivalent rate is given by name, never as a fraction of a peak it does not use.
        gn, gms = sum(by[k][1] for k in wkeys), sum(by[k][2] for k in wkeys)
        gms -= gn * event_pair_us * 1e-3
        rows = (88 + 88) + (88 + w['L'] + 88)
        alg = 2.0 * rows * 352 * B * w['T'] * reps                     # algorithmic flops of the products, both LSTMs
        exact = bool(getattr(eng, 'frames_exact_bf16', False))
        issued = 0.0
        for nz in (0, w['L']):                                         # encoder, decoder
            wide = 88 + nz > 96 or nz > 8
            h_tiles, x_tiles, col_tiles = (8 if wide else 6), 6, 24   # 16-row tiles of [h | z] and x; 2 x 12 column tiles
            mfmas = (x_tiles * (1 if exact else 3) * 3 + h_tiles * 6) * col_tiles * (B * w['T'] // 32)
            issued += mfmas * 2.0 * 16 * 16 * 32
        issued *= reps
        sq = None
        sq_run = (pmc or {}).get('sq') if (pmc is not None and 'error' not in pmc) else None
        if sq_run:        # measured in this run (the third PMC child pass)
            k = [v for kk, v in sq_run.items() if 'lstm_wgrad_bf16' in kk]
            if k:
                sq = dict(mfma_busy=k[0]['mfma_busy'], source='measured in this run (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES ...)')
        for tag in PROFILE_TAGS if sq is None else ():
            try:
                sj = json.load(open(os.path.join(ROOT, 'profiles', '%s_sq_%s.json' % (tag, wl_name))))
                k = [k for k in sj['kernels'] if 'lstm_wgrad_bf16' in k['kernel']]
                if k:
                    sq = dict(mfma_busy=k[0].get('mfma_busy'), source='profiles/%s_sq_%s.json' % (tag, wl_name))
                    break
            except Exception:
                continue
        roofline['gate_gemm'] = dict(
            kernel='+'.join(sorted(wkeys)), avg_launch_us=round(gms / gn * 1e3, 2), launches_per_step=round(gn / reps, 2),
            us_per_step=round(gms / reps * 1e3, 2), bound="mfma", pipe="bf16 MFMA (v_mfma_f32_16x16x32_bf16)",
            peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
            achieved=round(issued / (gms * 1e-3) / 1e12, 1), frac=round(issued / (gms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            achieved_is="issued bf16 MFMA flops (padded tiles, 6 piece pairs per fp32 product, 3 for byte-valued frames) / launch time",
            mfma_busy_counter=sq,
            fp32_equivalent_tflops=round(alg / (gms * 1e-3) / 1e12, 2),
            arithmetic="fp32 operands as 3 bf16 pieces each; 6 of the 9 piece pairs are multiplied (the 3 dropped ones are below "
                       "2^-25 of the product: fp32-rounding accuracy, not exact; byte-valued frame rows: 1 piece, all pairs), fp32 "
                       "accumulate; tests/test_gpu_ops.py bounds the error against float64 on cancelling inputs")
    return roofline


def measure_train(args, wl_name, dev, rank, world, steps, warmup, bf16=False, reps=40, want_roofline=True, pmc=None):
    """One training workload: engine, synthetic windows resident in HBM, set-up (graph capture + the kernel-time pass),
    warm-up, the timed blocks.  Returns the pieces of the JSON line."""
    import torch
    import torch.distributed as dist
    from clvae_amd.trainer import TrainStep
    w = WORKLOADS[wl_name]
    B, T = w['B'], w['T']
    eng, cfg = make_engine(w, dev, bf16=bf16)
    if world > 1:      # replicas start from rank 0's weights and optimizer state
        for t_ in eng.P.state_tensors():
            dist.broadcast(t_, src=0)
    X_all, Xp_all, w_all = synthetic_windows(w, 4 * B, 1234 + rank, dev)
    ts = TrainStep(eng, seed=1234, rank=rank, world=world, use_graph=not args.no_graph)
    if args.no_graph:      # (profiling passes run eagerly: the same launches as the replayed step, which stores no logits)
        ts._drop_logits = hasattr(eng, 'keep_logits') and 'keep_logits' not in eng.cfg
    nb = X_all.shape[0] // B

    # the step assembles its own batch from the HBM-resident windows (batch = device step counter mod nb), inside its graph:
    # one graph launch per step, nothing staged from the host (TrainStep.bind_batches; Model.fit does the same)
    ts.bind_batches(X_all, Xp_all, w_all, idx=None, period=nb, stride=B)

    def run(k):
        for i in range(k):
            ts.step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, before the W warm-up steps: (1) the step's hipGraph is captured (an eager step, the capture, a first
    # replay); (2) on rank 0 the roofline object's per-kernel HIP-event timing: eager steps of the same shapes on the same
    # stream -- its steps are rank 0's alone, so parameters and optimizer state are put back afterwards.  The warm-up and
    # the timed region follow at once, on a device that has been running this workload.
    run(3)
    barrier()
    if world > 1:          # coarse or fine weight-gradient grid: measured here, on this node, next to the real collectives
        ts.tune_dp_schedule()
        barrier()
    recs = recs_nolabel = None
    if rank == 0 and want_roofline:
        batch = (X_all[:B], Xp_all[:B], w_all[:B])
        recs = kernel_time_pass(eng, dict(seed=1234, rank=rank, world=1), batch, reps)
        if w['model'] == 'cl_vae':
            # a launch-bound step (three launches of 5-40 us, issued eagerly here): when the host falls behind, the interval
            # between a bracket's two events contains the wait for the launch itself (seen once: 70.8 us where rocprofv3 and every
            # other pass say 40-42).  Three passes, the one with the shortest fused-step bracket counts.
            tot = lambda rs: sum(r[2] for r in rs if r[0] in ('vae_fused_step', 'gemm_f32'))
            for _ in range(2):
                again = kernel_time_pass(eng, dict(seed=1234, rank=rank, world=1), batch, reps)
                if tot(again) < tot(recs):
                    recs = again
        if getattr(eng, 'fuse_pair', False) and getattr(eng, 'label_in_pair', False):
            recs_nolabel = kernel_time_pass(eng, dict(seed=1234, rank=rank, world=1), batch, max(reps // 2, 4), label_off=True)
    barrier()
    run(warmup)
    if os.environ.get('CLV_BENCH_FAULT_RANK') == str(rank):        # tests: one rank dies between warm-up and the timed region
        raise RuntimeError("CLV_BENCH_FAULT_RANK: injected failure of rank %d" % rank)
    dt, block_ms, rank_ms = timed_blocks(run, barrier, steps, world, dev)
    allreduce = allreduce_microbench(ts, dev) if world > 1 else None
    # what the host spends per step issuing it (staging launch + graph replays / plain launches / collectives), device idle
    # or not: 50 steps issued back to back, clock stopped BEFORE the synchronize
    barrier()
    t0 = time.perf_counter()
    run(50)
    host_us = 1e6 * (time.perf_counter() - t0) / 50
    barrier()
    loss = eng.losses()
    value = world * B * T * steps / dt
    roofline = None
    if recs is not None:
        roofline = roofline_object(args, w, wl_name, eng, recs, recs_nolabel, reps, value / world, bf16,
                                   kernel_times=args.kernel_times, pmc=pmc)
    return dict(w=w, eng=eng, ts=ts, value=value, dt=dt, block_ms=block_ms, rank_ms=rank_ms, allreduce=allreduce, loss=loss,
                roofline=roofline, host_us=round(host_us, 1))


def also_in_child(timeout_s=600):
    """The `also` list from a FRESH child process (`bench.py --also-only`), so that nothing an extra workload does -- a GPU
    fault, an abort inside a C-ABI call, an out-of-memory kill, a hang -- can cost the headline this process has already
    measured (and printed to stderr).  The child is started, never exec'd over this process; on a timeout exactly the
    process group started here is killed."""
    import signal
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--also-only']
    try:
        pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    except OSError as ex:
        return [{"error": "could not start the child: %r" % (ex,)}]
    try:
        so, se = pr.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(pr.pid, signal.SIGKILL)
        except OSError:
            pass
        pr.communicate()
        return [{"error": "the child process of the extra workloads passed %d s and was killed" % timeout_s}]
    for line in reversed(so.splitlines()):
        if line.startswith('['):
            try:
                return json.loads(line)
            except ValueError:
                break
    return [{"error": "child exit code %s, no list on stdout; stderr tail: %s" % (pr.returncode, se[-300:])}]


def also_list(args, dev):
    """The other BASELINE workloads on one GPU.  Runs inside `bench.py --also-only`: the child process a default run starts
    right after its headline (also_in_child)."""
    import copy
    import gc
    import torch
    out = []

    def drop():
        gc.collect()
        torch.cuda.empty_cache()

    for name, bf16, steps, warmup, reps in (('cfg5', False, 40, 8, 8), ('cfg2', False, 400, 40, 20), ('cfg2', True, 400, 40, 20)):
        try:
            m = measure_train(args, name, dev, 0, 1, steps, warmup, bf16=bf16, reps=reps)
            r = m['roofline'] or {}
            e = {"workload": name + ('_bf16' if bf16 else ''), "metric": "piano-roll timesteps/sec (train)",
                 "value": round(m['value'], 1), "unit": "timesteps/s", "ms_per_step": round(1e3 * m['dt'] / steps, 4),
                 "steps": steps, "warmup": warmup, "timed_blocks_ms_per_step": m['block_ms'],
                 "dtype": "bf16" if bf16 else "f32", "final_loss": round(float(m['loss']['total']), 4),
                 "config": "%s batch %d x seq_len %d, latent %d" % (m['w']['model'], m['w']['B'], m['w']['T'], m['w']['L']),
                 "roofline": {k: r.get(k) for k in ('kernel', 'frac', 'achieved', 'peak', 'unit', 'avg_launch_us', 'executes_on',
                                                    'algorithmic_step_frac')}}
            if 'gate_gemm' in r:
                e['roofline']['gate_gemm'] = {k: r['gate_gemm'].get(k) for k in ('avg_launch_us', 'frac', 'fp32_equivalent_tflops')}
            out.append(e)
            del m
        except Exception as ex:      # an extra must never cost the headline
            out.append({"workload": name, "error": repr(ex)[:200]})
        drop()
    for name, steps in (('gen1', 240), ('gen1024', 240), ('gen_vae1', 240), ('gen_vae1024', 240)):
        try:
            a2 = copy.copy(args)
            a2.steps, a2.warmup, a2.workload = steps, 4, name
            g = bench_generate(a2, WORKLOADS[name], dev, 0, 1)
            out.append({"workload": name, "metric": g["metric"], "value": g["value"], "unit": g["unit"],
                        "ms_per_step": g["ms_per_step"], "steps": steps, "config": g["config"]["workload"]})
        except Exception as ex:
            out.append({"workload": name, "error": repr(ex)[:200]})
        drop()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='cfg3', choices=sorted(WORKLOADS))
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--bf16', action='store_true',
                    help='cfg2: the Dense products of the fused cl_vae step on the bf16 matrix cores (fp32 accumulate)')
    ap.add_argument('--no-persistent', action='store_true', help='generation: per-frame hipGraph replay instead of the persistent kernel')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-also', action='store_true', help='skip the `also` list (the other workloads after a default cfg3 run)')
    ap.add_argument('--kernel-times', action='store_true', help='print per-kernel event times to stderr')
    ap.add_argument('--selftest-launch', action='store_true', help='launcher / rendezvous check on CPU (gloo), no timing')
    ap.add_argument('--no-pmc-traffic', action='store_true',
                    help='do not measure roofline.traffic with two rocprofv3 PMC child passes (a default single-GPU training run does); '
                         'the committed summary under profiles/ is read instead')
    ap.add_argument('--also-only', action='store_true', help='print only the `also` list (what a default run starts as its child)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        sys.exit(self_launch(args.gpus))
    if args.selftest_launch:
        return launch_selftest(args)

    # roofline.traffic, measured: two PMC passes over a short eager run, as child processes, BEFORE this process touches the GPU
    pmc = None
    w0 = WORKLOADS[args.workload]
    profiled = 'rocprof' in os.environ.get('LD_PRELOAD', '').lower() or any(k.startswith('ROCPROF') for k in os.environ)
    if (args.gpus == 1 and int(os.environ.get('WORLD_SIZE', '1')) == 1 and not args.no_pmc_traffic and not args.no_roofline
            and not args.also_only and not w0.get('generate') and not args.no_graph and not profiled):
        t0 = time.perf_counter()
        pmc = measure_pmc_traffic(args.workload, bf16=args.bf16)
        print("pmc traffic passes: %.1f s%s" % (time.perf_counter() - t0, "; " + pmc['error'] if 'error' in pmc else ""), file=sys.stderr)

    import torch
    import torch.distributed as dist
    import clvae_amd  # noqa: F401
    from clvae_amd import _lib
    from clvae_amd.parallel import init_from_env, shared_gpu

    rank, local, world = init_from_env()
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d (launch with torch.distributed.run --nproc-per-node %d, or run "
                         "`python bench.py --gpus %d` outside a process group and it starts the ranks itself)"
                         % (world, args.gpus, args.gpus, args.gpus))
    _lib.require_gpu()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    w = WORKLOADS[args.workload]
    B, T = w['B'], w['T']

    if args.also_only:
        print(json.dumps(also_list(args, dev)))
        return
    if w.get('generate'):
        g = bench_generate(args, w, dev, rank, world)
        if rank == 0:
            print(json.dumps(g))
        return
    m = measure_train(args, args.workload, dev, rank, world, args.steps, args.warmup, bf16=args.bf16,
                      want_roofline=not args.no_roofline, pmc=pmc)
    eng, ts = m['eng'], m['ts']

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(w)

    devices = [torch.cuda.get_device_name(local)]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, "rank %d: cuda:%d %s" % (rank, local, torch.cuda.get_device_name(local)))
    if rank == 0:
        try:
            rccl = '.'.join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            rccl = None
        out = {
            "metric": "piano-roll timesteps/sec (train)", "value": round(m['value'], 1), "unit": "timesteps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * m['dt'] / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if (args.bf16 and w['model'] == 'cl_vae') else "f32", "data": "synthetic",
            "config": {"workload": "%s: %s batch %d/GPU x seq_len %d, latent %d, %d classes, 88-dim piano-roll, "
                                   "Adam-WN, hipGraph=%s" % (args.workload, w['model'], B, T, w['L'], w['C'],
                                                             not args.no_graph),
                       "global_batch": world * B, "seq_len": T, "parallelism": "dp%d" % world},
            "timed_blocks": {"ms_per_step": m['block_ms'], "min": min(m['block_ms']), "max": max(m['block_ms']),
                             "reported": "median block" if len(m['block_ms']) > 1 else "the one block",
                             "rule": "every block = exactly --steps steps between barrier + synchronize; more than one block "
                                     "only when a block is shorter than 50 ms"},
            "final_loss": round(float(m['loss']['total']), 4),
            "roofline": m['roofline'], "cpu_baseline": cpu,
            "devices": devices, "rccl": rccl if world > 1 else None,
            "ms_per_step_by_rank": m['rank_ms'], "allreduce_alone": m['allreduce'],
            "host_issue_us_per_step": m['host_us'],
            "dp_schedule": None if world == 1 else {
                "graphs_per_step": (1 if ts._graphs and ts._graphs[0] == 'whole' else
                                    len([g for g in ts._graphs if g is not None]) if ts._graphs else 0),
                "collectives_per_step": 2, "collectives_captured": ts.capture_note,
                "wgrad_grid_trials": ts.dp_trials,
                "wgrad_split_scale": 2 if getattr(eng, 'fine_grid', False) else 1,
                "optimizer": "hW kernel updated (two launches, its sum g.V averaged with its gradient bucket) under the "
                             "main bucket's all-reduce, the rest after it"},
        }
        if shared_gpu():
            out["shared_device"] = ("CLV_BENCH_SHARE_GPU=1: all %d ranks on cuda:0, gloo carries the gradient buckets through "
                                    "the host -- a run of the world > 1 code path on a one-GPU box, NOT a scaling measurement"
                                    % world)
            out["backend"] = dist.get_backend() if world > 1 else None
        if world == 1 and args.workload == 'cfg3' and not args.bf16 and not args.no_also and not args.no_graph:
            # the headline is out (stderr, flushed) before any extra workload runs, and the extras run in a child process
            print("headline (repeated on stdout with `also`): " + json.dumps(out), file=sys.stderr, flush=True)
            del m, eng, ts
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["also"] = also_in_child()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
