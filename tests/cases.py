"""Shared parity-case tables (used by tools/make_golden.py in the build container and by
the tests everywhere).  Inputs are regenerated from rick_amd.synth by key name."""
import torch

from rick_amd.synth import synth_tensor

UPFIRDN_CASES = [
    # (tag, up, down, pad0, pad1, N, C, H, W, ksize)   live call sites first
    ('g_blur_8',      1, 1, 1, 1, 2, 3, 17, 17, 4),
    ('g_blur_odd',    1, 1, 1, 1, 1, 2, 9, 9, 4),
    ('d_blur22_16',   1, 1, 2, 2, 2, 3, 16, 16, 4),
    ('d_blur11_16',   1, 1, 1, 1, 2, 3, 16, 16, 4),
    ('rgb_up2_8',     2, 1, 2, 1, 2, 3, 8, 8, 4),
    ('rgb_up2_4',     2, 1, 2, 1, 1, 3, 4, 4, 4),
    ('down2_pad11',   1, 2, 1, 1, 2, 3, 16, 16, 4),
    ('down2_odd',     1, 2, 1, 1, 1, 1, 7, 7, 4),
    ('ada_up2_k12',   2, 1, 6, 5, 1, 3, 9, 9, 12),
    ('ada_down2_k12', 1, 2, 5, 5, 1, 3, 33, 33, 12),
    ('negpad',        1, 1, -1, -2, 1, 2, 17, 17, 4),
    ('up2_down2',     2, 2, 1, 1, 1, 2, 8, 8, 4),
    ('rect',          1, 1, 2, 1, 1, 2, 7, 12, 4),
    ('k3',            1, 1, 1, 1, 1, 2, 8, 8, 3),
    ('up3_k6',        3, 1, 3, 2, 1, 1, 5, 5, 6),
]


def upfirdn_kernel(ksize, up):
    if ksize == 4:
        k = torch.tensor([1., 3., 3., 1.])
        k = torch.outer(k, k)
        k = k / k.sum()
        return k * (up ** 2)
    return synth_tensor(f'upfirdn/k{ksize}', (ksize, ksize)).abs() / ksize ** 2


