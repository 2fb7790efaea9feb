"""LDS bank-conflict checks of the MFMA kernels' patch reads against the gfx950 lane-group / bank model
(tools/lds_sim.py; MI355X_MICROARCH.md 'LDS').  CPU only: the layouts are host-side decisions (the position map of
rick_convt2_posmap, the swizzle key bit of the weight-gradient patch)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import lds_sim  # noqa: E402

# the transposed-conv launches of the 256-px networks (G upsampling layers at batch 4 / 2, D data gradients at N = 8 / 4)
CT2_SHAPES = [(4, 4, 4, 512, 512), (4, 8, 8, 512, 512), (4, 16, 16, 512, 512), (4, 32, 32, 512, 512), (4, 64, 64, 512, 256),
              (4, 128, 128, 256, 128), (8, 128, 128, 256, 128), (8, 64, 64, 512, 256), (8, 16, 16, 512, 512), (2, 64, 64, 512, 256),
              (1, 128, 128, 256, 128), (3, 9, 6, 64, 32), (2, 5, 7, 36, 20)]


@pytest.mark.parametrize('shape', CT2_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_convt2_position_map_is_conflict_free(shape):
    """Every ds_read_b128 of convt2_kernel's patch takes the minimum 4 LDS cycles with the plan's position map (and the
    map is a permutation of the tile's positions: asserted inside the simulator); the identity map of round 2 was 2-way."""
    assert lds_sim.ct2(*shape, use_map=True, verbose=False) == 4.0
    assert lds_sim.ct2(*shape, use_map=False, verbose=False) > 4.0 or shape[1] * shape[2] < 16


def test_wgrad_patch_swizzle_key_bit():
    """conv_wgrad_kernel's transposing patch reads: 16-wide stride-1 position tiles are conflict-free with the slot key on
    bit 3 of the pixel index (2 cycles per ds_read_b64_tr_b16) and 2-way with the igemm's bit 2."""
    assert lds_sim.wgrad_patch_reads(4, 2, 18, 1, 3, verbose=False) == 2.0
    assert lds_sim.wgrad_patch_reads(4, 2, 18, 1, 2, verbose=False) == 4.0


def test_stride2_patch_columns_deinterleaved():
    """Stride-2 layers (conv_tiling.h cv_patch_col): with row-major patch rows every transposing read of the weight-gradient
    kernel is a 2-way conflict whatever single key bit is used; with the even / odd columns of a row de-interleaved the
    gathered pixels are neighbours and the stride-1 key (bit 3, 16-wide tile) is conflict-free."""
    assert min(lds_sim.wgrad_patch_reads(4, 2, 33, 2, kb, verbose=False) for kb in range(7)) == 4.0
    assert lds_sim.wgrad_patch_reads(4, 2, 33, 2, 3, verbose=False, deint=True) == 2.0
    assert lds_sim.wgrad_patch_reads(2, 2, 9, 2, 2, verbose=False, deint=True) == 2.0
    assert lds_sim.wgrad_patch_reads(3, 3, 17, 2, 5, verbose=False, deint=True) < 2.25


CT2_LAYERS = [(n, ih, ci, co) for n in (1, 2, 3, 4, 8, 25)
              for ih, ci, co in ((4, 512, 512), (8, 512, 512), (16, 512, 512), (32, 512, 512), (64, 512, 256), (128, 256, 128))]


@pytest.mark.parametrize('n,ih,ci,co', CT2_LAYERS)
def test_convt2_plan_covers_every_tile_once_and_fills_rounds(n, ih, ci, co):
    """rick_convt2_plan (host code): the grid of the one-pass transposed conv is whole-tile blocks in full rounds of 256 plus
    the work items of the last, partly filled round as 2 / 4 blocks of 4 / 2 fragment columns (csrc/convt2.hip, DESIGN
    §3.4) — every work item is covered exactly once, the sub-blocks fit one round, tiles hold <= 128 positions."""
    import ctypes
    from rick_amd._lib import lib
    plan = (ctypes.c_int * 8)()
    oh = 2 * ih + 1
    assert lib.rick_convt2_plan(n, ih, ih, ci, co, oh, oh, plan) == 0
    tw, th, nb, tiles, nsplit, cps, nfull, subq = list(plan)
    assert 1 <= tw * th * nb <= 128 and nb <= n
    g = ih + 1                                            # positions per image side
    assert tiles == -(-g // tw) * -(-g // th) * -(-n // nb) * -(-co // 128)
    nchunks = -(-ci // 32)
    assert 1 <= nsplit <= 16 and nsplit == -(-nchunks // cps)
    items = tiles * nsplit
    assert subq in (1, 2, 4) and 0 <= nfull <= items
    if subq == 1:
        assert nfull == items
    else:
        assert nfull % 256 == 0 and nfull == items // 256 * 256 and 0 < (items - nfull) * subq <= 256
    ws = lib.rick_convt2_workspace_bytes(n, ih, ih, ci, co, oh, oh)
    assert ws == (nsplit * n * oh * oh * co * 4 if nsplit > 1 else 0)
