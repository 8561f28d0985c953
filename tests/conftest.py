import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the CPU oracle's op sizes do not scale past ~32 threads; with every core of a 2-socket GPU host (256 threads) the same
    # step is ~25x slower (measured: 375 s instead of 14 s)
    import torch
    torch.set_num_threads(min(os.cpu_count() or 1, 32))


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load


_REFERENCE_PINNED_FILES = ('test_oracle_golden.py', 'test_modules_gpu.py', 'test_pipeline_golden.py', 'test_boundary_cpu.py', 'test_mmcv_semantics_cpu.py')


def _evidence_rank(item):
    """0: compares with a fixture the reference wrote (``golden``) — 1: compares with the CPU oracle — 2: everything else (workload runs,
    product-vs-product self comparisons).  The driver runs ``-x``: a flaky self comparison must never hide a reference-fixture test again."""
    import inspect
    if 'golden' in getattr(item, 'fixturenames', ()) or os.path.basename(str(item.fspath)) in _REFERENCE_PINNED_FILES:
        return 0
    try:
        src = inspect.getsource(item.function)
    except (OSError, TypeError, AttributeError):
        return 2
    if 'golden(' in src or 'GOLDEN' in src or '.npz' in src:
        return 0
    if 'O.' in src or 'oracle' in src or 'f64ref' in src:
        return 1
    return 2


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=_evidence_rank)          # stable: file / definition order is kept inside each rank
