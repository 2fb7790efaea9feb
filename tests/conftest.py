import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'saturates: the test provokes fp16 saturation on purpose (the counter may be non-zero)')


# Collection order of the GPU suite: kernel parity first, then the models / loop body, then the rows either side of the
# hot path, and the multi-process data-parallel tests LAST — under `-x` a red late test must never hide the parity tests.
_ORDER = ['test_gpu_ops', 'test_gpu_split', 'test_gpu_models', 'test_gpu_step_batching', 'test_gpu_determinism',
          'test_augment', 'test_eval_stats', 'test_data_path', 'test_checkpoint', 'test_gpu_dp']


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(mod) if mod in _ORDER else len(_ORDER) - 1.5   # unknown modules: just before test_gpu_dp
    items.sort(key=rank)                                                   # stable: order inside a module is kept


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + '.npz'))
        return cache[name]
    return load


@pytest.fixture(autouse=True)
def _no_silent_saturation(request):
    """Every GPU test must leave the library's saturation counter at zero (include/rick_hip.h: rick_saturation_count): an
    operand far above its block's sampled maximum, or a value above a split-image producer's bound, yields finite but wrong
    products — never silently.  Tests that provoke the event on purpose carry @pytest.mark.saturates."""
    if request.node.get_closest_marker('gpu') is None:
        yield
        return
    import ctypes

    import torch
    from rick_amd._lib import check, lib
    c = ctypes.c_uint(0)
    check(lib.rick_saturation_count(ctypes.byref(c), 1), 'rick_saturation_count')
    yield
    if request.node.get_closest_marker('saturates') is None:
        torch.cuda.synchronize()
        check(lib.rick_saturation_count(ctypes.byref(c), 1), 'rick_saturation_count')
        assert c.value == 0, f'{c.value} saturation events during {request.node.name}'
