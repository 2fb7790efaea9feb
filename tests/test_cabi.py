"""CPU-only: the C-ABI library loads and exports every symbol include/rick_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'rick_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rick_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from rick_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_lib.lib, n), f'{n} declared in rick_hip.h but not exported'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature in rick_amd/_lib.py'
    assert _lib.lib.rick_abi_version() == 1


def test_geom_struct_matches_header():
    from rick_amd._lib import ConvGeom, MAX_TAPS
    # 15 ints + 3*16 ints + int + float
    assert ctypes.sizeof(ConvGeom) == 4 * (15 + 3 * MAX_TAPS + 2)


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from rick_amd import op
    with pytest.raises(RuntimeError):
        op.fused_leaky_relu(torch.zeros(2, 4), torch.zeros(4))
    with pytest.raises(RuntimeError):
        op.conv2d(torch.zeros(1, 4, 4, 4), torch.zeros(4, 4, 3, 3), 1, 1)


def test_conv_tuning_is_a_host_side_switch():
    """rick_conv_tuning (include/rick_hip.h): returns the previous value, -1 for an unknown key; the shipped defaults select the
    forms that measured fastest (eight-wave stride-1 igemm off, eight-wave stride-2 igemm on, in-launch split-K fix-up off, 8 x 8 FIR tile) — no GPU involved."""
    from rick_amd._lib import lib
    defaults = {0: 0, 1: 192, 2: 0, 3: 0, 4: 2}
    for key, dflt in defaults.items():
        prev = lib.rick_conv_tuning(key, 7)
        assert prev == dflt, (key, prev)
        assert lib.rick_conv_tuning(key, prev) == 7
        assert lib.rick_conv_tuning(key, prev) == prev
    assert lib.rick_conv_tuning(99, 1) == -1 and lib.rick_conv_tuning(-1, 1) == -1
