"""Where the bucketed and the blocking data-parallel runs of tests/test_gpu_dp.py differ (parameter, magnitude)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_dp import _run  # noqa: E402

if __name__ == '__main__':
    from rick_amd.models import Discriminator, Generator
    from rick_amd.train import FlatParams, d_optim_filter, g_optim_filter
    a, b = _run('bucketed'), _run('blocking')
    g, d = Generator(32, 512, 8), Discriminator(32)
    for net, mod, flt in ((1, g, g_optim_filter), (2, d, d_optim_filter)):
        fp = FlatParams(mod.named_parameters(), flt)
        x, y = a[0][net], b[0][net]
        print('net', net, 'equal', np.array_equal(x, y), 'ranks equal', np.array_equal(a[0][net], a[1][net]))
        for n in fp.names:
            lo, hi = fp.segment(n)
            dd = np.abs(x[lo:hi] - y[lo:hi]).max()
            if dd > 0:
                print(f'  {n:44s} max abs diff {dd:.3e}  (max |p| {np.abs(y[lo:hi]).max():.3e})')
