"""LDS bank-conflict simulator for the patch-image reads of the MFMA conv kernels (gfx950 lane groups and bank rule from
/opt/skills/guides/MI355X_MICROARCH.md, 'LDS'):  cycles per ds_read_b128 / ds_read_b64_tr_b16 wave-instruction for a given
patch geometry, swizzle and tap offset.  CPU only.

    python tools/lds_sim.py ct2 N IH IW Ci Co          # convt2_kernel's B reads for the planner's tile
    python tools/lds_sim.py wgrad TW TH PW is          # conv_wgrad_kernel's transposing patch reads
"""
import ctypes
import itertools
import sys

G128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128 = G128 + [[l + 32 for l in g] for g in G128]
G64 = [list(range(32)), list(range(32, 64))]


def cycles(addrs, groups, width):
    """addrs[lane] = byte address; width bytes per lane; returns LDS cycles (1 per group when conflict-free)."""
    tot = 0
    for g in groups:
        bank_addrs = {}
        for l in g:
            a = addrs[l]
            for d in range(width // 4):
                bank_addrs.setdefault(((a // 4) + d) % 64, set()).add((a + 4 * d) // 4)
        tot += max(len(v) for v in bank_addrs.values())
    return tot


def swz_default(kg, row):
    return kg ^ (((row >> 2) & 1) << 1)


def b128_patch_read(pp_of_l15, swz=swz_default, row_bytes=64):
    """ds_read_b128 of the B operand: lane = kg*16 + l15 reads 16 B at pp*row_bytes + swz(kg, pp)*16."""
    addrs = [0] * 64
    for lane in range(64):
        l15, kg = lane & 15, lane >> 4
        pp = pp_of_l15[l15]
        addrs[lane] = pp * row_bytes + swz(kg, pp) * 16
    return cycles(addrs, G128, 16)


def ct2(N, IH, IW, Ci, Co, swz=swz_default, use_map=True, verbose=True):
    """Average LDS cycles per ds_read_b128 of convt2_kernel's patch reads for the planner's tile, with the plan's
    position map (rick_convt2_posmap) or the identity map (round-2 behaviour: slot j*16 + l15 = position)."""
    sys.path.insert(0, '.')
    from rick_amd._lib import lib
    out = (ctypes.c_int * 8)()
    assert lib.rick_convt2_plan(N, IH, IW, Ci, Co, 2 * IH + 1, 2 * IW + 1, out) == 0
    TW, TH, NB = out[0], out[1], out[2]
    pm = (ctypes.c_ubyte * 128)()
    pitch = ctypes.c_int()
    assert lib.rick_convt2_posmap(N, IH, IW, Ci, Co, 2 * IH + 1, 2 * IW + 1, pm, ctypes.byref(pitch)) == 0
    PW, PH = (pitch.value if use_map else TW + 1), TH + 1
    seen = set()
    tot = n = 0
    for toff_y, toff_x in itertools.product((0, -1), (0, -1)):
        for j in range(8):
            pps = []
            for l15 in range(16):
                ent = pm[j * 16 + l15] if use_map else min(j * 16 + l15, TW * TH * NB - 1)
                pos = ent & 127
                if use_map and not (ent & 128):
                    seen.add(pos)
                nbi, rem = divmod(pos, TW * TH)
                ty, tx = divmod(rem, TW)
                nbi = min(nbi, NB - 1)
                pps.append((nbi * PH + ty + 1 + toff_y) * PW + tx + 1 + toff_x)
            tot += b128_patch_read(pps, swz)
            n += 1
    if use_map:
        assert seen == set(range(TW * TH * NB)), 'position map is not a permutation of the tile'
    if verbose:
        print(f'ct2 N{N} {IH}x{IW} {Ci}->{Co}: tile {TW}x{TH}x{NB}, pitch {PW}, {"position map" if use_map else "identity"}: '
              f'{tot / n:.2f} cycles per ds_read_b128 (4 = conflict-free)')
    return tot / n


if __name__ == '__main__':
    if sys.argv[1] == 'ct2':
        a = [int(v) for v in sys.argv[2:7]]
        ct2(*a, use_map=False)
        ct2(*a, use_map=True)


def wgrad_patch_reads(tw_log2, th_log2, PW, is_, key_bit, nbe=1, PH=None, verbose=True, deint=False):
    """conv_wgrad_kernel's B-operand reads: ds_read_b64_tr_b16, lane (G = lane>>4, q = (lane>>2)&3, p = lane&3) reads 8 B of
    patch row pbase(r) + toff, r = kk*32 + G*8 + h*4 + q, at slot (b_kg ^ key(pp)) * 16 + (p&1) * 8, b_kg = wn*2 + (p>>1).
    deint (stride 2, round 4): patch rows stored [even columns | odd columns] (conv_tiling.h cv_patch_col) — the gathered
    pixels are neighbours and key bit 3 is conflict-free for the 16-wide tile (4.0 -> 2.0 cycles)."""
    tw, th = 1 << tw_log2, 1 << th_log2
    tot = n = 0
    half = (PW + 1) // 2
    col = (lambda x: (x >> 1) + (x & 1) * half) if deint else (lambda x: x)
    toffs = ([dy * PW + col(dx) for dy in range(3) for dx in range(3)] if deint else list(range(0, 3 * PW, max(PW // 2, 1))))
    for kk in range(2):
        for h in range(2):
            for wn in range(2):
                for toff in toffs:
                    addrs = []
                    for lane in range(64):
                        G, q, p = lane >> 4, (lane >> 2) & 3, lane & 3
                        r = kk * 32 + G * 8 + h * 4 + q
                        px, py, nbi = r & (tw - 1), (r >> tw_log2) & (th - 1), r >> (tw_log2 + th_log2)
                        nbi = min(nbi, nbe - 1)
                        pp = (nbi * (PH or ((th - 1) * is_ + 3)) + py * is_) * PW + (px if deint else px * is_) + toff
                        b_kg = wn * 2 + (p >> 1)
                        key = ((pp >> key_bit) & 1) << 1
                        addrs.append(pp * 64 + (b_kg ^ key) * 16 + (p & 1) * 8)
                    tot += cycles(addrs, G64, 8)
                    n += 1
    if verbose:
        print(f'wgrad tile {tw}x{th} PW {PW} stride {is_} key bit {key_bit}{" de-interleaved" if deint else ""}: {tot / n:.2f} cycles per ds_read_b64_tr_b16 (2 = conflict-free)')
    return tot / n
