"""Oracle (test infrastructure): ctypes access to oracle/_build/liboracle.so (plain-C scalar
restatement of the reference kernels' index math; see oracle/csrc/oracle_ops.c)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle.so')
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(['make', '-C', _HERE], stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(_SO)
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if a is not None else None


def upfirdn2d_c(x, k, up=(1, 1), down=(1, 1), pad=(0, 0, 0, 0)):
    """x: float32 [N,C,H,W]; k: float32 [kh,kw]; pad = (x0, x1, y0, y1)."""
    x = np.ascontiguousarray(x, np.float32)
    k = np.ascontiguousarray(k, np.float32)
    n, c, h, w = x.shape
    kh, kw = k.shape
    oh = (h * up[1] + pad[2] + pad[3] - kh) // down[1] + 1
    ow = (w * up[0] + pad[0] + pad[1] - kw) // down[0] + 1
    out = np.empty((n, c, oh, ow), np.float32)
    rc = lib().oracle_upfirdn2d_f32(_fp(x), _fp(k), _fp(out), ctypes.c_int64(n * c), h, w, kh, kw,
                                    up[0], up[1], down[0], down[1], pad[0], pad[1], pad[2], pad[3])
    assert rc == 0
    return out


def bias_act_c(x, b, ref, act, grad, alpha, scale):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    step_b = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    b_ = np.ascontiguousarray(b, np.float32) if b is not None and b.size else None
    r_ = np.ascontiguousarray(ref, np.float32) if ref is not None and ref.size else None
    lib().oracle_bias_act_f32(_fp(x), _fp(b_), _fp(r_), _fp(out), ctypes.c_int64(x.size),
                              ctypes.c_int64(step_b), ctypes.c_int64(b_.size if b_ is not None else 1),
                              act, grad, ctypes.c_float(alpha), ctypes.c_float(scale))
    return out
