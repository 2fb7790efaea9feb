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
