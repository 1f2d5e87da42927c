import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a `-m gpu` run on a box without a GPU must fail loudly, never silently skip
    pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _poisoned_device_memory(request):
    """-m gpu tests run on POISONED device memory (round 6): the engines' buffers are torch.empty by design (every kernel writes
    what a later one reads), and on a fresh box "empty" reads as zeros -- a read-before-write then passes every test and fails
    behind any process that left other bytes in the HBM (found: the dropout chain's padded decoder inputs, after bench.py had run
    on the same box).  Before each GPU test the blocks the caching allocator will hand out next are filled with NaN, so such a
    read shows up as a NaN here.  CLV_TEST_POISON=0 switches it off."""
    if request.node.get_closest_marker('gpu') is None or os.environ.get('CLV_TEST_POISON', '1') == '0':
        yield
        return
    try:
        import torch
    except ImportError:
        yield
        return
    if torch.cuda.is_available():
        nan = float('nan')
        blocks = [torch.full((64 << 20,), nan, dtype=torch.float32, device='cuda') for _ in range(3)]      # large pool: 3 x 256 MB
        blocks += [torch.full((n,), nan, dtype=torch.float32, device='cuda') for n in (1 << 22, 1 << 20, 1 << 18) for _ in range(8)]
        blocks += [torch.full((100000,), nan, dtype=torch.float32, device='cuda') for _ in range(64)]       # small pool (< 1 MB)
        blocks += [torch.full((n,), nan, dtype=torch.float32, device='cuda') for n in (16384, 2048, 256, 16) for _ in range(256)]
        torch.cuda.synchronize()
        del blocks
    yield
