"""-m gpu: `python bench.py` exactly as the driver runs it (no flags but a short --steps / --warmup): ONE JSON line on stdout with
the contract's keys, `roofline` (traffic measured in the run), `cpu_baseline`, and the `also` list from its child process --
every piece of the default path, so that a slip in any of them (round 6: a NameError behind the headline, in the `also` child's
timeout) fails here and not in the driver's run."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def test_default_bench_line_is_whole():
    pytest.importorskip("torch")
    import clvae_amd  # noqa: F401
    from clvae_amd import _lib
    _lib.require_gpu()
    env = dict(os.environ)
    env.pop('CLV_LIB', None)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '10', '--warmup', '3'], cwd=ROOT, env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [l for l in pr.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, pr.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 10 and d['warmup'] == 3 and d['dtype'] == 'f32' and d['vs_baseline'] is None
    assert abs(d['value'] - 256 * 128 / (d['ms_per_step'] * 1e-3)) < 1e-3 * d['value']
    r = d['roofline']
    assert r['bound'] == 'mfma' and 0.05 < r['frac'] < 1.0 and r['peak'] == 157.3 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert r['traffic'] is None or r['traffic'] > 1e6
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['value'] > 0 and c['cores'] >= 1
    also = d.get('also')
    assert isinstance(also, list) and len(also) >= 6, also
    assert not [a for a in also if 'error' in a], also
