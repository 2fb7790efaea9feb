"""Convolution family on the MFMA implicit-GEMM kernels (rick_amd/csrc/conv.hip).

Replaces the F.conv2d / F.conv_transpose2d calls of the reference model
(model_probe_tune.py:122,265,274,280) and their autograd.  Three bilinear primitives that
are closed under differentiation, so R1 / path-length second-order terms need nothing else:

    conv  (x, w)       y[n,o,p]      = sum_{i,k} w[o,i,k] x[n,i,p*s + k - pad]
    convT (x, w)       y[n,o,q]     += w[o,i,k] x[n,i,p]      with q = p*s + k - pad
    wgrad (a, b)       gw[o,i,k]     = sum_{n,p} a[n,o,p] b[n,i,p*s + k - pad]

    d conv /dx = convT(g, w^T)     d conv /dw = wgrad(g, x)
    d convT/dx = conv (g, w^T)     d convT/dw = wgrad(x, g)^T
    d wgrad/da = conv (b, gg)      d wgrad/db = convT(a, gg^T)

(w^T swaps the two channel axes; no spatial flips are needed because the tap geometry is
explicit.)  All activations are channels-last; weights are any [O, I, kh, kw] view whose last
two dims are jointly contiguous.  `set_precision()` selects fp16x3 (default: fp16 hi/lo split with a
per-block power-of-two operand exponent, 2^-22 per product, fp32 accumulate) or plain fp16 MFMA.
"""
import ctypes
import os
import threading

import numpy as np
import torch
from torch.autograd import Function

from .._lib import MAX_TAPS, ConvEpilogue, ConvGeom, check, lib, ptr, require_cuda_f32, stream_ptr

_SPLIT = 2          # 2: fp16 hi/lo split, 3 MFMAs per product (fp32-grade); 1: plain fp16
_weights_epoch = 0  # bumped by optimisers that update parameters through raw pointers


_tls = threading.local()   # per-thread switch of the gradient sink (DataParallel-style callers run the ops from threads)
_prof = None               # bench.py's launch profiler: process-wide on purpose — backward launches are issued by the autograd
                           # engine's device thread, not by the thread that entered the context


class launch_profiler:
    """Context manager used by bench.py: brackets every conv-family launch with HIP events on
    the launch stream (torch's current stream) and records its algorithmic FLOPs."""

    def __enter__(self):
        global _prof
        _prof = []
        return _prof

    def __exit__(self, *a):
        global _prof
        _prof = None


def _launch(kind, flops, fn, *args, tag='', abytes=0.0):
    if _prof is None:
        return fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    _prof.append((kind, flops, e0, e1, tag, abytes))
    return rc


def hbm_launch(name, nbytes, fn, *args):
    """An HBM-bound launch under bench.py's launch profiler: `nbytes` = the bytes it has to move once (its inputs + outputs)."""
    if _prof is None:
        return fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    _prof.append(('hbm', 0.0, e0, e1, name, float(nbytes)))
    return rc


def set_precision(name):
    """'fp16x3' (default; parity-grade, ~2^-22 relative per product) or 'fp16' (single pass, ~3x the MFMA rate,
    ~2^-12 relative).

    PROCESS-WIDE on purpose, like the packed-weight caches and PackGroup tables: a backward pass is issued by the autograd
    engine's device thread, not by the thread that ran the forward, so a thread-local setting would let the two halves of
    one step disagree.  The C ABI below is re-entrant (rick_hip.h); this Python layer supports ONE training thread per
    process — the deployment model is one process per GPU (rick_amd/dist.py), not nn.DataParallel's worker threads.  Only
    the switches that are read in an op's forward (`second_order`, `grad_sink`) are thread-local."""
    global _SPLIT
    _SPLIT = {'fp16x3': 2, 'fp16': 1}[name]


def get_precision():
    return 'fp16x3' if _SPLIT == 2 else 'fp16'


def weights_stamp(param):
    """What a cached function of `param`'s values is valid for (the same triple the packed-weight caches compare)."""
    grp = getattr(param, '_rick_group', None)
    return (param._version, _weights_epoch, grp.epoch if grp is not None else -1, param.data_ptr())


def bump_weights_epoch(params=None):
    """Invalidate cached packed weights (call after any raw-pointer parameter update).  With `params`
    (the parameters that were updated) only their PackGroups go stale; parameters outside any group, or
    no argument, invalidate everything."""
    global _weights_epoch
    if params is not None:
        groups = [getattr(p, '_rick_group', None) for p in params]
        if all(g is not None for g in groups):
            for g in {id(g): g for g in groups}.values():
                g.epoch += 1
            return
    _weights_epoch += 1


def _nhwc(x):
    return x.contiguous(memory_format=torch.channels_last)


def _empty_nhwc(n, c, h, w, like):
    return torch.empty((n, c, h, w), device=like.device, dtype=like.dtype, memory_format=torch.channels_last)


def _w_strides(w):
    """(tensor, s_o, s_i, s_t) with element (o, i, ky*kw+kx) at s_o*o + s_i*i + s_t*(ky*kw+kx)."""
    kh, kw = w.shape[2], w.shape[3]
    st = w.stride()
    if kh * kw > 1 and not (st[2] == kw * st[3]):
        w = w.contiguous()
        st = w.stride()
    s_t = st[3] if kh * kw > 1 else 1
    return w, st[0], st[1], s_t


# rick_pack_desc (include/rick_hip.h), 64 bytes
_DESC = np.dtype([('w', '<u8'), ('s_co', '<i8'), ('s_ci', '<i8'), ('s_t', '<i8'), ('packed', '<u8'), ('Co', '<i4'),
                  ('Ci', '<i4'), ('nslices', '<i4'), ('scale', '<f4'), ('blk_begin', '<i4'), ('reserved', '<i4')])


class PackGroup:
    """All packed conv weights of one network, refreshed by ONE launch.

    The optimiser updates every parameter of a network together, so the first stale lookup after a step
    repacks every (parameter view, orientation) the network has asked for so far — rick_conv_pack_weights_multi —
    instead of one pack launch + one allocation per layer and orientation.  Destination buffers and the device
    descriptor table persist across steps.

    The pack launch may be captured into a hipGraph (RickTrainer._run), which bakes in the table pointer, the entry
    count and the block count.  The table is therefore APPEND-ONLY at fixed capacity: a request registered after a
    capture lands behind the entries the captured launch reads (its prefix, and the prefix's block ranges, never
    change), and a table that has to be rebuilt (capacity exceeded, a parameter's storage moved) is retired, not
    freed — replays keep reading valid memory."""

    CAPACITY = 256          # descriptors per table (one per parameter view and orientation; a network needs < 100)

    def __init__(self):
        self.reqs = {}          # request key -> dict(param, buf, desc fields, stamp); insertion order == table order
        self.table = None       # device copy of the descriptor array (CAPACITY entries)
        self.host = None        # host mirror
        self.n = 0              # entries in use
        self.total_blocks = 0
        self.retired = []       # tables a captured graph may still reference
        self.epoch = 0          # bumped when this network's parameters were updated through raw pointers
        self.after_repack = []  # callables run behind the pack launch (weight-only side products: modconv.DemodBank's wsq)

    def _append(self, req):
        """Write the request's descriptor behind the existing ones (host mirror + the one device entry)."""
        dev = req['buf'].device
        if self.table is None or self.n >= self.host.shape[0]:
            cap = max(self.CAPACITY, 2 * self.n)
            host = np.zeros(cap, dtype=_DESC)
            if self.table is not None:
                host[:self.n] = self.host[:self.n]
                self.retired.append(self.table)
            self.host = host
            self.table = torch.from_numpy(host.view(np.uint8).copy()).to(dev)
        w, s_o, s_i, s_t, packed, O, I, ns, sc = req['desc']
        self.host[self.n] = (w, s_o, s_i, s_t, packed, O, I, ns, sc, self.total_blocks, 0)
        sz = _DESC.itemsize
        self.table[self.n * sz:(self.n + 1) * sz].copy_(torch.from_numpy(self.host[self.n:self.n + 1].view(np.uint8).copy()))
        self.total_blocks += lib.rick_conv_pack_blocks(O, I)
        self.n += 1

    def _rebuild(self):
        """Some parameter's storage moved: a fresh table from the surviving requests (the old one is retired)."""
        if self.table is not None:
            self.retired.append(self.table)
        self.table, self.host, self.n, self.total_blocks = None, None, 0, 0
        for r in self.reqs.values():
            self._append(r)

    def lookup(self, w, scale, key):
        param, tag = key
        rk = (id(param), tag, float(scale), tuple(w.shape), w.stride(), w.data_ptr())
        req = self.reqs.get(rk)
        if req is None:
            w2, s_o, s_i, s_t = _w_strides(w)
            if w2.data_ptr() != w.data_ptr():
                return None                      # needed a temporary copy: not a stable view, pack individually
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('PackGroup: a packed-weight request was registered during hipGraph capture; run the '
                                   'step eagerly once before capturing it')
            O, I, kh, kw = w.shape
            buf = torch.empty(lib.rick_conv_packed_bytes(O, I, kh * kw), device=w.device, dtype=torch.uint8)
            req = dict(param=param, off=w.data_ptr() - param.data_ptr(), buf=buf, stamp=None,
                       desc=(w.data_ptr(), s_o, s_i, s_t, buf.data_ptr(), O, I, kh * kw, float(scale)))
            self.reqs[rk] = req
            self._append(req)
            # first use: pack just this one (the group launch takes over from the next refresh on)
            check(lib.rick_conv_pack_weight(w.data_ptr(), s_o, s_i, s_t, O, I, kh * kw, float(scale), _SPLIT,
                                            buf.data_ptr(), stream_ptr()), 'rick_conv_pack_weight')
            req['stamp'] = (param._version, _weights_epoch, self.epoch, _SPLIT)
        if req['stamp'] != (param._version, _weights_epoch, self.epoch, _SPLIT):
            self._repack()
        return req['buf']

    def refresh(self, skip=None):
        """Repack now if any registered request is stale (one launch).  RickTrainer calls this on the host before it
        captures or replays a step graph, so the graphs themselves contain no pack launches and a network is repacked
        once per update of its weights instead of once per step graph that uses it.  `skip`: a FlatParams whose
        optimiser step is still pending (data-parallel pipelining) — a group that holds its parameters is left alone."""
        if skip is not None and self.reqs:
            first = next(iter(self.reqs.values()))['param']
            if any(first is p for p in skip.params):
                return False
        for r in self.reqs.values():
            if r['stamp'] != (r['param']._version, _weights_epoch, self.epoch, _SPLIT):
                self._repack()
                return True
        return False

    def _repack(self):
        # requests whose parameter storage moved (e.g. .to(), re-flattening) are dropped; they re-register on use
        stale = [k for k, r in self.reqs.items() if r['param'].data_ptr() + r['off'] != r['desc'][0]]
        if stale:
            for k in stale:
                del self.reqs[k]
            self._rebuild()
        for cb in self.after_repack:
            cb()
        if not self.reqs:
            return
        check(lib.rick_conv_pack_weights_multi(ptr(self.table), self.n, self.total_blocks, _SPLIT, stream_ptr()),
              'rick_conv_pack_weights_multi')
        for r in self.reqs.values():
            r['stamp'] = (r['param']._version, _weights_epoch, self.epoch, _SPLIT)


def register_pack_group(*modules):
    """Put every parameter of `modules` (one network) into one PackGroup; returns the group."""
    grp = PackGroup()
    for m in modules:
        for p in m.parameters():
            p._rick_group = grp
    return grp


def _pack(w, scale, key=None):
    """Pack w[O, I, kh, kw] (any strides) * scale for the igemm A operand.  `key` = (param, tag)
    enables caching across calls until the parameter changes; parameters that belong to a PackGroup are
    refreshed together with the rest of their network."""
    O, I, kh, kw = w.shape
    if key is not None:
        grp = getattr(key[0], '_rick_group', None)
        if grp is not None:
            buf = grp.lookup(w, scale, key)
            if buf is not None:
                return buf
        ent = getattr(key[0], '_rick_packed', None)
        sig = (key[1], key[0]._version, _weights_epoch, _SPLIT, float(scale), tuple(w.shape), w.stride(), w.data_ptr())
        if ent is not None and sig in ent:
            return ent[sig]
    w, s_o, s_i, s_t = _w_strides(w)
    nbytes = lib.rick_conv_packed_bytes(O, I, kh * kw)
    buf = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    check(lib.rick_conv_pack_weight(ptr(w), s_o, s_i, s_t, O, I, kh * kw, float(scale), _SPLIT, ptr(buf), stream_ptr()),
          'rick_conv_pack_weight')
    if key is not None:
        if ent is None:
            ent = {}
            key[0]._rick_packed = ent      # lives and dies with the parameter object
        # keep only entries of the current parameter version
        for k in [k for k in ent if k[1] != sig[1] or k[2] != sig[2]]:
            del ent[k]
        ent[sig] = buf
    return buf


def _geom(N, IH, IW, Ci, OH, OW, Co, GH, GW, is_, os_, oy0, ox0, taps, nslices, alpha=1.0):
    g = ConvGeom()
    g.N, g.IH, g.IW, g.Ci, g.OH, g.OW, g.Co = N, IH, IW, Ci, OH, OW, Co
    g.GH, g.GW, g.is_, g.os, g.oy0, g.ox0 = GH, GW, is_, os_, oy0, ox0
    if len(taps) > MAX_TAPS:
        raise RuntimeError(f'conv: at most {MAX_TAPS} taps per launch')
    g.ntaps, g.nslices = len(taps), nslices
    for t, (dy, dx, wt) in enumerate(taps):
        g.dy[t], g.dx[t], g.wt[t] = dy, dx, wt
    g.split, g.alpha = _SPLIT, alpha
    return g


_geom_cache = {}
_EDGE_STRIPS = False   # interior + 1-wide strips for (8k+1)x(16k+1) class grids: measured neutral on MI355X, kept off


def _igemm_ws(g, like):
    """Split-K workspace for launches with few output tiles (None when the kernel does not need one)."""
    nbytes = lib.rick_conv_igemm_workspace_bytes(ctypes.byref(g))
    if nbytes < 0:
        raise RuntimeError('rick_conv_igemm_workspace_bytes: invalid geometry')
    return torch.empty(nbytes, device=like.device, dtype=torch.uint8) if nbytes else None


def conv_out_size(i, k, s, p):
    return (i + 2 * p - k) // s + 1


# ------------------------------------------------------------------------------ raw launches
def _epilogue(bias, noise, noise_w, slope, gain):
    """rick_conv_epilogue for `gain * lrelu(conv + bias + noise_w * noise)`; noise is [1 or N, 1, OH, OW]."""
    e = ConvEpilogue()
    e.bias = ptr(bias)
    e.noise = ptr(noise)
    e.noise_w = ptr(noise_w) if noise is not None else None
    e.noise_nb = noise.shape[0] if noise is not None else 1
    e.act, e.slope, e.gain = 1, float(slope), float(gain)
    return e


def _conv_launch(x, wp, O, kh, kw, s, p, iscale=None, oscale=None, alpha=1.0, epi=None, x_split=None):
    """`x_split`: the input as a split image (op/split.py; its producer has folded `iscale` in) — `x` may then be None."""
    x = _nhwc(x) if x_split is None else x_split.data
    N, I, IH, IW = x.shape
    key = ('c', N, I, IH, IW, O, kh, kw, s, p, alpha, _SPLIT)
    ent = _geom_cache.get(key)
    if ent is None:
        OH, OW = conv_out_size(IH, kh, s, p), conv_out_size(IW, kw, s, p)
        taps = [(ky - p, kx - p, ky * kw + kx) for ky in range(kh) for kx in range(kw)]
        g = _geom(N, IH, IW, I, OH, OW, O, OH, OW, s, 1, 0, 0, taps, kh * kw, alpha)
        nbytes = lib.rick_conv_igemm_workspace_bytes(ctypes.byref(g))
        if nbytes < 0:
            raise RuntimeError('rick_conv_igemm_workspace_bytes: invalid geometry')
        ent = (g, ctypes.byref(g), nbytes, OH, OW, 2.0 * N * OH * OW * O * I * kh * kw,
               f'conv {I}->{O} k{kh} s{s} N{N} {IH}x{IW}', 4.0 * (N * IH * IW * I + N * OH * OW * O + O * I * kh * kw))
        _geom_cache[key] = ent
    g, gref, nbytes, OH, OW, flops, tag, abytes = ent
    y = _empty_nhwc(N, O, OH, OW, x)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8) if nbytes else None
    if x_split is not None:
        check(_launch('igemm', flops, lib.rick_conv_igemm_split_f32, ptr(x), ptr(x_split.hdr), ptr(wp), ptr(y), ptr(oscale), gref,
                      ctypes.byref(epi) if epi is not None else None, ptr(ws), stream_ptr(), tag=tag + ' [split]', abytes=abytes),
              'rick_conv_igemm_split_f32')
        return y
    if epi is not None:
        check(_launch('igemm', flops, lib.rick_conv_igemm_act_f32, ptr(x), ptr(wp), ptr(y), ptr(iscale), ptr(oscale), gref,
                      ctypes.byref(epi), ptr(ws), stream_ptr(), tag=tag, abytes=abytes), 'rick_conv_igemm_act_f32')
        return y
    check(_launch('igemm', flops, lib.rick_conv_igemm_f32, ptr(x), ptr(wp), ptr(y), ptr(iscale), ptr(oscale), gref,
                  ptr(ws), stream_ptr(), tag=tag, abytes=abytes), 'rick_conv_igemm_f32')
    return y


_USE_CT2 = True     # tools/bench_conv.py switches the dedicated stride-2 kernel off to time the generic multi-class launch


def _convT2_launch(x, wp, O, out_hw, iscale, oscale, alpha, x_split=None, amax=None):
    """3x3 stride-2 padding-0 transposed convolution on the single-staging kernel (csrc/convt2.hip)."""
    N, I, IH, IW = x.shape
    OH, OW = out_hw
    key = ('t2', N, I, IH, IW, O, OH, OW)
    ent = _geom_cache.get(key)
    if ent is None:
        nbytes = lib.rick_convt2_workspace_bytes(N, IH, IW, I, O, OH, OW)
        if nbytes < 0:               # tensor of >= 2^31 elements, or no tile whose patch + scale table fits the LDS
            _geom_cache[key] = (None, 0.0, '')
            return None
        # algorithmic FLOPs: every (input pixel, tap) pair whose output pixel exists
        ny = [sum(1 for iy in range(IH) if 2 * iy + k < OH) for k in range(3)]
        nx = [sum(1 for ix in range(IW) if 2 * ix + k < OW) for k in range(3)]
        ent = (nbytes, 2.0 * N * O * I * sum(ny) * sum(nx), f'convT {I}->{O} k3 s2 N{N} {IH}x{IW}',
               4.0 * (N * IH * IW * I + N * OH * OW * O + O * I * 9))
        _geom_cache[key] = ent
    nbytes, flops, tag, abytes = ent[0], ent[1], ent[2], (ent[3] if len(ent) > 3 else 0.0)
    if nbytes is None:
        return None
    y = _empty_nhwc(N, O, OH, OW, x)
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8) if nbytes else None
    if x_split is not None:
        check(_launch('igemm', flops, lib.rick_convt2_split_f32, ptr(x), ptr(x_split.hdr), ptr(wp), ptr(y), ptr(oscale), N, IH, IW,
                      I, O, OH, OW, alpha, ptr(amax), ptr(ws), stream_ptr(), tag=tag + ' [split]', abytes=abytes), 'rick_convt2_split_f32')
        return y
    check(_launch('igemm', flops, lib.rick_convt2_f32, ptr(x), ptr(wp), ptr(y), ptr(iscale), ptr(oscale), N, IH, IW, I, O,
                  OH, OW, _SPLIT, alpha, ptr(ws), stream_ptr(), tag=tag, abytes=abytes), 'rick_convt2_f32')
    return y


def _convT_launch(x, wp, O, kh, kw, s, p, out_hw, iscale=None, oscale=None, alpha=1.0, x_split=None, amax=None):
    """y[q] += w[k] x[pos], q = pos*s + k - p: the output parity classes (s*s of them) run as one launch.
    `x_split`: the input as a split image (stride 1, or the 3x3 stride-2 single-staging kernel)."""
    x = _nhwc(x) if x_split is None else x_split.data
    N, I, IH, IW = x.shape
    OH, OW = out_hw
    if (_USE_CT2 and kh == 3 and kw == 3 and s == 2 and p == 0 and I % 4 == 0 and O % 4 == 0
            and OH in (2 * IH, 2 * IH + 1) and OW in (2 * IW, 2 * IW + 1)):
        y = _convT2_launch(x, wp, O, out_hw, iscale, oscale, alpha, x_split=x_split, amax=amax)
        if y is not None:            # (None: the single-staging kernel has no plan for this size -> generic launch below)
            return y
    if x_split is not None and s != 1:
        raise RuntimeError('convT: no split-image form for this geometry')
    key = ('t', N, I, IH, IW, O, kh, kw, s, p, OH, OW, alpha, _SPLIT)
    ent = _geom_cache.get(key)
    if ent is None:
        classes = []
        for py in range(s):
            for px in range(s):
                taps = [((py + p - ky) // s, (px + p - kx) // s, ky * kw + kx)
                        for ky in range(kh) for kx in range(kw)
                        if (py + p - ky) % s == 0 and (px + p - kx) % s == 0]
                GH, GW = (OH - py + s - 1) // s, (OW - px + s - 1) // s
                if GH > 0 and GW > 0:
                    classes.append((py, px, GH, GW, taps))
        full = all(len(c[4]) > 0 for c in classes)
        live = [c for c in classes if c[4]]
        # A class grid of (8k+1) x (16k+1) positions (stride-2 transposed conv of an 8k x 16k input) would
        # pad almost a whole extra tile row / column: cut it into an aligned interior plus 1-wide edge strips.
        rects = []
        for (py, px, GH, GW, taps) in live:
            ys = [(0, GH - 1), (GH - 1, GH)] if (_EDGE_STRIPS and GH > 16 and GH % 8 == 1) else [(0, GH)]
            xs = [(0, GW - 1), (GW - 1, GW)] if (_EDGE_STRIPS and GW > 16 and GW % 16 == 1) else [(0, GW)]
            parts = [(ys[0], xs[0])]
            if len(ys) > 1:
                parts.append((ys[1], (0, GW)))
            if len(xs) > 1:
                parts.append((ys[0], xs[1]))
            for (y0, y1), (x0, x1) in parts:
                rects.append((py + y0 * s, px + x0 * s, y1 - y0, x1 - x0,
                              [(dy + y0, dx + x0, wt) for dy, dx, wt in taps]))
        if len(rects) > 8:
            rects = [(py, px, GH, GW, taps) for (py, px, GH, GW, taps) in live]
        geoms = (ConvGeom * len(rects))()
        flops = 0.0
        for i, (oy0, ox0, GH, GW, taps) in enumerate(rects):
            geoms[i] = _geom(N, IH, IW, I, OH, OW, O, GH, GW, 1, s, oy0, ox0, taps, kh * kw, alpha)
            flops += 2.0 * N * GH * GW * O * I * len(taps)
        live = rects
        nbytes = lib.rick_conv_igemm_multi_workspace_bytes(geoms, len(live))
        if nbytes < 0:
            raise RuntimeError('rick_conv_igemm_multi_workspace_bytes: invalid geometry')
        ent = (geoms, len(live), nbytes, full, flops, f'convT {I}->{O} k{kh} s{s} N{N} {IH}x{IW}',
               4.0 * (N * IH * IW * I + N * OH * OW * O + O * I * kh * kw))
        _geom_cache[key] = ent
    geoms, ngeom, nbytes, full, flops, tag, abytes = ent
    y = _empty_nhwc(N, O, OH, OW, x)
    if not full:
        y.zero_()
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8) if nbytes else None
    if x_split is not None:          # stride 1: one class
        check(_launch('igemm', flops, lib.rick_conv_igemm_split_f32, ptr(x), ptr(x_split.hdr), ptr(wp), ptr(y), ptr(oscale),
                      ctypes.byref(geoms[0]), None, ptr(ws), stream_ptr(), tag=tag + ' [split]', abytes=abytes), 'rick_conv_igemm_split_f32')
        return y
    check(_launch('igemm', flops, lib.rick_conv_igemm_multi_f32, ptr(x), ptr(wp), ptr(y), ptr(iscale), ptr(oscale),
                  geoms, ngeom, ptr(ws), stream_ptr(), tag=tag, abytes=abytes), 'rick_conv_igemm_multi_f32')
    return y


class grad_sink:
    """Inside this context the first-order conv / modulated-conv backward ADDS a weight gradient straight into the
    parameter's existing ``.grad`` (the trainer's flat gradient buffer) from the wgrad kernel's second stage and reports
    no gradient to autograd — instead of materialising it, routing it through the select / view nodes of the parameter and
    an AccumulateGrad add (three extra passes over a 9.4 MB tensor per 512x512 layer).  Only valid around a plain
    ``loss.backward()`` on leaf parameters whose ``.grad`` is allocated (RickTrainer's D / G steps); ``autograd.grad`` users
    (Fisher sweep, tests) never enable it.  The switch is per thread and is read in the FORWARD of the fused ops (the
    backward runs on the autograd engine's device thread, where a thread-local would not be visible): wrap forward and
    backward."""

    def __enter__(self):
        self.prev, _tls.grad_sink = getattr(_tls, 'grad_sink', False), True

    def __exit__(self, *a):
        _tls.grad_sink = self.prev


def grad_sink_enabled():
    return getattr(_tls, 'grad_sink', False)


_param_grads_off = False     # process-wide on purpose: read by backward passes on the autograd engine's thread


class no_param_grads:
    """Around an ``autograd.grad(outputs, inputs=<activations / latents>, create_graph=True)`` call (the inner gradient of
    the R1 and path-length terms, train_dynamic_update_prune.py:89-96, 104-118): the backward passes of the conv /
    transposed-conv / fused-activation ops skip the gradients of operands that are leaf parameters (or views of one).
    torch's own conv backward is told by the engine which outputs the call needs; a custom Function is not — its
    ``needs_input_grad`` only mirrors ``requires_grad`` — so without this switch every conv layer runs a full weight-
    gradient kernel whose result the inner call discards (a third of the weight-gradient time of a path-length step).
    Gradients of non-leaf operands (anything that may depend on the differentiation inputs) are always produced."""

    def __enter__(self):
        global _param_grads_off
        self.prev, _param_grads_off = _param_grads_off, True

    def __exit__(self, *a):
        global _param_grads_off
        _param_grads_off = self.prev


def param_like(t):
    """Leaf tensor or a view of one (``Parameter[0]``, ``.view``) — evaluated in an op's forward."""
    return t is not None and (t.is_leaf or (t._is_view() and t._base is not None and t._base.is_leaf))


def skip_param_grad(is_param_like):
    return _param_grads_off and is_param_like


def _sink_target(key, shape, enabled):
    """The [O, I, kh, kw] view of the parameter's .grad a weight gradient may be added into, or None.  `enabled` is the
    switch as the op's forward saw it."""
    if not enabled or key is None:
        return None
    p = key[0]
    if not (p.is_leaf and p.requires_grad and p.grad is not None and p.grad.is_contiguous() and p.numel() == int(np.prod(shape))):
        return None
    return p.grad.view(shape)


# ---- weight gradients off the critical path -----------------------------------------------------------------------------
# A layer's data gradient feeds the next layer's backward; its weight gradient feeds nothing until the optimiser.  Inside
# `wgrad_overlap()` every SUNK weight-gradient launch (op.grad_sink(): the result is added into the parameter's .grad, no tensor
# goes back to autograd) is issued on a second HIP stream: the hardware then fills the CUs a data-gradient launch leaves idle —
# the tail of its last round of blocks, the 4^2...32^2 layers whose grids cover a quarter of the chip — with weight-gradient
# blocks, in eager issue and (fork / join captured) in the replayed step graphs.  Values cannot change: every launch reads and
# writes what it did before, each .grad is written by ONE stream between fork and join (the DemodBank's accumulate joins
# first), and the kernels sum in a fixed order.
# MEASURED (round 5, tools/ab_overlap.sh, two alternating same-box pairs): 158.3 images/s with the side stream against 165.3
# without (D step 10.70 vs 10.38 ms, G step 10.21 vs 9.92): the MFMA launches already fill the chip (a weight-gradient wave
# takes a whole SIMD's 512 registers, so its blocks only land on CUs that have drained), two co-running kernels evict each
# other's L2 lines, and the replayed graphs gain fork / join nodes.  OFF by default (RICK_WGRAD_OVERLAP=1 switches it on);
# tests/test_gpu_step_batching.py keeps the value-equality test.
_side = {'on': False, 'stream': None, 'pending': False}
_OVERLAP_OFF = not os.environ.get('RICK_WGRAD_OVERLAP')


class wgrad_overlap:
    """with op.grad_sink(), op.wgrad_overlap(): loss.backward()   — joins the side stream on exit."""

    def __enter__(self):
        self.prev = _side['on']
        if not _OVERLAP_OFF and torch.cuda.is_available():
            if _side['stream'] is None or _side['stream'].device.index != torch.cuda.current_device():
                if torch.cuda.is_current_stream_capturing():
                    return self               # the side stream must exist before a capture (run the step eagerly once)
                _side['stream'] = torch.cuda.Stream()
            _side['on'] = True
        return self

    def __exit__(self, *exc):
        join_side()
        _side['on'] = self.prev


def join_side():
    """Make the current stream wait for the weight gradients issued on the side stream so far."""
    if _side['pending']:
        torch.cuda.current_stream().wait_stream(_side['stream'])
        _side['pending'] = False


def _wgrad_launch(a, b, kh, kw, s, p, alpha=1.0, ascale=None, bscale=None, out=None, transposed=False, a_split=None,
                  b_split=None):
    """_wgrad_launch_now — on the side stream when the result is sunk and `wgrad_overlap()` is open (not under bench.py's
    launch profiler: its per-kernel event timings want the kernels one at a time)."""
    if out is None or not _side['on'] or _prof is not None:
        return _wgrad_launch_now(a, b, kh, kw, s, p, alpha, ascale, bscale, out, transposed, a_split, b_split)
    side = _side['stream']
    side.wait_stream(torch.cuda.current_stream())          # the operands' producers (and the zeroing of .grad) come first
    with torch.cuda.stream(side):
        r = _wgrad_launch_now(a, b, kh, kw, s, p, alpha, ascale, bscale, out, transposed, a_split, b_split)
    # autograd frees the operands as soon as the node returns: the allocator must not hand their memory to a main-stream
    # launch before the side stream is done with it
    for t in (a, b, ascale, bscale, getattr(a_split, 'data', None), getattr(a_split, 'hdr', None),
              getattr(b_split, 'data', None), getattr(b_split, 'hdr', None)):
        if t is not None:
            t.record_stream(side)
    _side['pending'] = True
    return r


def _wgrad_launch_now(a, b, kh, kw, s, p, alpha=1.0, ascale=None, bscale=None, out=None, transposed=False, a_split=None,
                      b_split=None):
    """gw[o,i,ky,kx] = alpha * sum a[n,o,pos] b[n,i,pos*s + k - p]  -> contiguous [O, I, kh, kw].
    `out`: ADD the result into this contiguous tensor instead ([O, I, kh, kw], or [I, O, kh, kw] when `transposed` — the
    parameter layout of a transposed convolution's weight gradient)."""
    if a_split is not None:
        a = a_split.data                 # (shape carrier; `a` / `b` may be None when only the image exists)
    if b_split is not None:
        b = b_split.data
    a, b = _nhwc(a), _nhwc(b)
    N, O, AH, AW = a.shape
    _, I, BH, BW = b.shape
    key = ('w', N, O, AH, AW, I, BH, BW, kh, kw, s, p, alpha, _SPLIT)
    ent = _geom_cache.get(key)
    if ent is None:
        taps = [(ky - p, kx - p, ky * kw + kx) for ky in range(kh) for kx in range(kw)]
        if len(taps) > 9:
            raise RuntimeError('wgrad: kernels larger than 3x3 are not supported')
        g = _geom(N, BH, BW, I, AH, AW, O, AH, AW, s, 1, 0, 0, taps, kh * kw, alpha)
        nbytes = lib.rick_conv_wgrad_workspace_bytes(ctypes.byref(g))
        if nbytes < 0:
            raise RuntimeError('rick_conv_wgrad_workspace_bytes: invalid geometry')
        ent = (g, ctypes.byref(g), max(nbytes, 16), 2.0 * N * AH * AW * O * I * kh * kw,
               f'wgrad {I}x{O} k{kh} s{s} N{N} a{AH}x{AW} b{BH}x{BW}', 4.0 * (N * AH * AW * O + N * BH * BW * I + O * I * kh * kw))
        _geom_cache[key] = ent
    g, gref, nbytes, flops, tag, abytes = ent
    ws = torch.empty(nbytes, device=a.device, dtype=torch.uint8)
    K = kh * kw
    if a_split is not None or b_split is not None:
        # split-image operands (op/split.py): no fp32 -> fp16 conversion inside the kernel
        if not lib.rick_conv_wgrad_split_supported(gref):
            raise RuntimeError('wgrad: geometry has no split-image form')
        if out is not None:
            s_co, s_ci, acc, dst = (K, O * K, 1, out) if transposed else (I * K, K, 1, out)
        else:
            dst = torch.empty((O, I, kh, kw), device=a.device, dtype=torch.float32)
            s_co, s_ci, acc = I * K, K, 0
        check(_launch('wgrad', flops, lib.rick_conv_wgrad_split_f32, ptr(b_split.data if b_split is not None else b),
                      ptr(b_split.hdr) if b_split is not None else None, ptr(a_split.data if a_split is not None else a),
                      ptr(a_split.hdr) if a_split is not None else None, ptr(dst), s_co, s_ci, 1, gref, acc, ptr(ws),
                      stream_ptr(), tag=tag + ' [split]', abytes=abytes), 'rick_conv_wgrad_split_f32')
        return None if out is not None else dst
    if out is not None:
        s_co, s_ci = (K, O * K) if transposed else (I * K, K)
        check(_launch('wgrad', flops, lib.rick_conv_wgrad_f32, ptr(b), ptr(a), ptr(out), s_co, s_ci, 1,
                      ptr(ascale), ptr(bscale), gref, 1, ptr(ws), stream_ptr(), tag=tag, abytes=abytes), 'rick_conv_wgrad_f32')
        return None
    gw = torch.empty((O, I, kh, kw), device=a.device, dtype=a.dtype)
    check(_launch('wgrad', flops, lib.rick_conv_wgrad_f32, ptr(b), ptr(a), ptr(gw), I * K, K, 1,
                  ptr(ascale), ptr(bscale), gref, 0, ptr(ws), stream_ptr(), tag=tag, abytes=abytes), 'rick_conv_wgrad_f32')
    return gw


# ------------------------------------------------------------------------ autograd primitives
class _Conv(Function):
    @staticmethod
    def forward(ctx, x, w, s, p, wscale, key):
        O, I, kh, kw = w.shape
        if x.shape[1] != I:
            raise RuntimeError(f'conv: input has {x.shape[1]} channels, weight expects {I}')
        ctx.save_for_backward(x, w)
        ctx.cfg = (s, p, wscale, key)
        ctx.w_param = param_like(w)
        return _conv_launch(x, _pack(w, wscale, key and (key[0], key[1] + '/conv')), O, kh, kw, s, p)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        s, p, wscale, key = ctx.cfg
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _ConvT.apply(g, w.transpose(0, 1), s, p, wscale, (x.shape[2], x.shape[3]),
                              key and (key[0], key[1] + '/T'))
        if ctx.needs_input_grad[1] and not skip_param_grad(ctx.w_param):
            gw = _WGrad.apply(g, x, w.shape[2], w.shape[3], s, p, wscale)
        return gx, gw, None, None, None, None


class _ConvT(Function):
    @staticmethod
    def forward(ctx, x, w, s, p, wscale, out_hw, key):
        O, I, kh, kw = w.shape
        if x.shape[1] != I:
            raise RuntimeError(f'convT: input has {x.shape[1]} channels, weight expects {I}')
        ctx.save_for_backward(x, w)
        ctx.cfg = (s, p, wscale, key)
        ctx.w_param = param_like(w)
        return _convT_launch(x, _pack(w, wscale, key and (key[0], key[1] + '/convT')), O, kh, kw, s, p, out_hw)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        s, p, wscale, key = ctx.cfg
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _Conv.apply(g, w.transpose(0, 1), s, p, wscale, key and (key[0], key[1] + '/T'))
        if ctx.needs_input_grad[1] and not skip_param_grad(ctx.w_param):
            # wgrad(a = x [I ch], b = g [O ch]) -> [I, O, kh, kw]; transpose back to w's [O, I, ..]
            gw = _WGrad.apply(x, g, w.shape[2], w.shape[3], s, p, wscale).transpose(0, 1)
        return gx, gw, None, None, None, None, None


class _WGrad(Function):
    @staticmethod
    def forward(ctx, a, b, kh, kw, s, p, alpha):
        ctx.save_for_backward(a, b)
        ctx.cfg = (kh, kw, s, p, alpha)
        return _wgrad_launch(a, b, kh, kw, s, p, alpha)

    @staticmethod
    def backward(ctx, gg):
        a, b = ctx.saved_tensors
        kh, kw, s, p, alpha = ctx.cfg
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = _Conv.apply(b, gg, s, p, alpha, None)
            if ga.shape[2:] != a.shape[2:]:
                ga = ga[:, :, :a.shape[2], :a.shape[3]]
        if ctx.needs_input_grad[1]:
            gb = _ConvT.apply(a, gg.transpose(0, 1), s, p, alpha, (b.shape[2], b.shape[3]), None)
        return ga, gb, None, None, None, None, None


class _ConvBiasAct(Function):
    """EqualConv2d followed by FusedLeakyReLU (every ConvLayer of D, model_probe_tune.py:595-641) with the bias +
    LeakyReLU tail applied in the convolution's epilogue: the separate activation pass (one read + one write of the
    feature map) disappears from the forward; values are bit-identical to conv -> fused_leaky_relu.  The hand-written
    backward is first order; under create_graph=True the backward differentiates the composed form (op/_twice.py)."""

    @staticmethod
    def forward(ctx, x, w, bias, s, p, wscale, key, slope, gain):
        O, I, kh, kw = w.shape
        if x.shape[1] != I:
            raise RuntimeError(f'conv: input has {x.shape[1]} channels, weight expects {I}')
        ctx.bias = bias                             # the Parameter itself (gradient sink target)
        ctx.plike = (False, param_like(w), param_like(bias))
        bias = bias.contiguous()
        y = _conv_launch(x, _pack(w, wscale, key and (key[0], key[1] + '/conv')), O, kh, kw, s, p,
                         epi=_epilogue(bias, None, None, slope, gain))
        ctx.save_for_backward(x, w, y)
        ctx.cfg = (s, p, wscale, key, slope, gain)
        ctx.sink = grad_sink_enabled()
        return y

    @staticmethod
    def backward(ctx, g):
        from .fused_act import _ActAdjoint, fused_leaky_relu, param_sink
        x, w, y = ctx.saved_tensors
        s, p, wscale, key, slope, gain = ctx.cfg
        O, I, kh, kw = w.shape
        if torch.is_grad_enabled():     # create_graph=True (R1): differentiate the composed, twice-differentiable form
            from ._twice import second_order_backward
            bias = ctx.bias
            gx, gw, gb = second_order_backward(
                lambda x_, w_, b_: fused_leaky_relu(_Conv.apply(x_, w_, s, p, wscale, key), b_, slope, gain),
                (x, w, bias), ctx.needs_input_grad[:3], g, ctx.plike)
            return gx, gw, gb, None, None, None, None, None, None
        want_b = ctx.needs_input_grad[2]
        gz, gb, _ = _ActAdjoint.apply(g, y, None, slope, gain, want_b, False, param_sink(ctx.bias, O, ctx.sink and want_b))
        gx = gw = None
        if ctx.needs_input_grad[0]:
            wpT = _pack(w.transpose(0, 1), wscale, key and (key[0], key[1] + '/T/convT'))
            gx = _convT_launch(gz, wpT, I, kh, kw, s, p, (x.shape[2], x.shape[3]))
        if ctx.needs_input_grad[1]:
            gw = _wgrad_launch(gz, x, kh, kw, s, p, wscale, out=_sink_target(key, w.shape, ctx.sink))
        return gx, gw, gb, None, None, None, None, None, None


def conv2d_bias_act(x, w, bias, stride=1, padding=0, wscale=1.0, key=None, negative_slope=0.2, gain=2 ** 0.5):
    """gain * leaky_relu(conv2d(x, w * wscale) + bias) in one launch (needs Co % 4 == 0; first-order autograd)."""
    require_cuda_f32(x, w, bias)
    return _ConvBiasAct.apply(x, w, bias, stride, padding, wscale, key, negative_slope, gain)


def conv2d(x, w, stride=1, padding=0, wscale=1.0, key=None):
    """y = conv2d(x, w * wscale) (cross-correlation, like F.conv2d).  w: [O, I, kh, kw]."""
    require_cuda_f32(x, w)
    return _Conv.apply(x, w, int(stride), int(padding), float(wscale), key)


def conv_transpose2d(x, w, stride=2, padding=0, wscale=1.0, key=None):
    """y = F.conv_transpose2d(x, (w * wscale) given as [O, I, kh, kw]) — note torch stores the
    transposed-conv weight as [I, O, kh, kw]; pass ``w_torch.transpose(0, 1)``."""
    require_cuda_f32(x, w)
    kh, kw = w.shape[2], w.shape[3]
    oh = (x.shape[2] - 1) * stride - 2 * padding + kh
    ow = (x.shape[3] - 1) * stride - 2 * padding + kw
    return _ConvT.apply(x, w, int(stride), int(padding), float(wscale), (oh, ow), key)
