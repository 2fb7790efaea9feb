"""Where the bucketed and the blocking data-parallel runs of tests/test_gpu_dp.py differ (parameter, count, magnitude) —
and whether each mode agrees with ITSELF (mode dependence vs run-to-run non-determinism)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_dp import _run, describe_difference  # noqa: E402

if __name__ == '__main__':
    a, a2, b, b2 = _run('bucketed'), _run('bucketed'), _run('blocking'), _run('blocking')
    for label, x, y in (('bucketed vs bucketed', a, a2), ('blocking vs blocking', b, b2), ('bucketed vs blocking', a, b)):
        for net in (1, 2):
            print(label, 'net', net, 'equal', np.array_equal(x[0][net], y[0][net]),
                  'ranks equal', np.array_equal(x[0][net], x[1][net]))
            print(describe_difference(net, x[0][net], y[0][net]))
