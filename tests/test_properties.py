"""Property tests (CPU): upfirdn2d output-size formula, C oracle vs PyTorch oracle on random
configurations incl. negative pads (SURVEY.md §4 'property' row)."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from oracle import c_ref
from oracle.ops_ref import upfirdn2d_out_size, upfirdn2d_ref


@settings(max_examples=60, deadline=None)
@given(up=st.integers(1, 3), down=st.integers(1, 3), p0=st.integers(-2, 5), p1=st.integers(-2, 5),
       h=st.integers(4, 14), w=st.integers(4, 14), k=st.integers(1, 5), seed=st.integers(0, 2 ** 16))
def test_upfirdn2d_c_oracle_matches_torch_oracle(up, down, p0, p1, h, w, k, seed):
    oh = upfirdn2d_out_size(h, up, down, p0, p1, k)          # op/upfirdn2d.py:103-104
    ow = upfirdn2d_out_size(w, up, down, p0, p1, k)
    if oh <= 0 or ow <= 0 or h * up + p0 + p1 < k or w * up + p0 + p1 < k:
        return
    if h * up + min(p0, 0) + min(p1, 0) <= 0 or w * up + min(p0, 0) + min(p1, 0) <= 0:
        return
    rs = np.random.RandomState(seed)
    x = torch.from_numpy(rs.randn(1, 2, h, w).astype(np.float32))
    kern = torch.from_numpy(rs.rand(k, k).astype(np.float32))
    y = upfirdn2d_ref(x.double(), kern.double(), up, down, (p0, p1))
    assert y.shape == (1, 2, oh, ow)                           # == op/upfirdn2d_kernel.cu:237-240
    yc = c_ref.upfirdn2d_c(x.numpy(), kern.numpy(), (up, up), (down, down), (p0, p1, p0, p1))
    assert yc.shape == (1, 2, oh, ow)
    assert np.allclose(yc, y.numpy(), rtol=1e-5, atol=1e-5)
