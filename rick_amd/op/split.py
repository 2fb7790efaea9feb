"""Split images: activations / gradients stored pre-split (fp16 {hi x 4 | lo x 4} in the 16 bytes of every 4 channels, one
power-of-two exponent per tensor) for the MFMA convolution kernels — see rick_amd/csrc/conv_common.h and include/rick_hip.h.

A producer that is handed a guaranteed bound on |v| writes the image in its epilogue (next to, or instead of, the fp32
tensor); consumers (igemm, convt2, wgrad) copy 16-byte granules into LDS with no conversion work.  The reference has no
counterpart (its convolutions are cuDNN's, model_probe_tune.py:122,265,274,280)."""
import threading

import torch

from .._lib import check, lib, ptr, stream_ptr


def capture_id():
    """Identity of the hipGraph capture the current stream records into (0: eager issue)."""
    import ctypes
    v = ctypes.c_ulonglong(0)
    check(lib.rick_stream_capture_id(stream_ptr(), ctypes.byref(v)), 'rick_stream_capture_id')
    return v.value


class _Arena(threading.local):
    """Zero-initialised device words handed out in slices (running maxima, headers): one fill launch per 128 KB
    instead of one per tensor.  A chunk belongs to ONE capture (or to eager issue): the words a captured launch accumulates
    into must be re-zeroed by a fill that is part of the same graph — two graphs captured back to back (the D step, then the
    G step) used to share a chunk whose fill only the first one replayed.  The key is the stream's capture
    id, so it holds on the autograd engine's thread (its own thread-local arena) without any hand-over."""

    def __init__(self):
        self.chunk, self.pos, self.cap = None, 0, 0

    def take(self, n, device):
        cap = capture_id()
        n4 = (n + 3) // 4 * 4                      # 16-byte aligned slices
        if self.chunk is None or self.pos + n4 > self.chunk.numel() or cap != self.cap or self.chunk.device != device:
            self.chunk, self.pos, self.cap = torch.zeros(32768, device=device, dtype=torch.float32), 0, cap
        out = self.chunk[self.pos:self.pos + n]
        self.pos += n4
        return out


_arena = _Arena()
AMAX_FLOATS = 512        # RICK_AMAX_FLOATS: 16 slots, one per 128-byte line (include/rick_hip.h)


def new_words(n, device):
    """n zeroed float words (16-byte aligned)."""
    return _arena.take(n, torch.device(device))


def new_amax(device):
    """A zeroed running-maximum word (its value is the maximum over its slots: `amax_value`)."""
    return _arena.take(AMAX_FLOATS, torch.device(device))


def amax_value(word):
    return word.max()


class SplitImage:
    """data: the image (a float32 channels-last tensor of the activation's shape used as a byte container),
    hdr: float32[4] on the device = {2^e, 2^-e, bound, 0},
    bound: (amax word, amax word or None, coef) the producer derived its exponent from — |v| <= coef * (a0 + a1); a kernel
    whose result is bounded by this tensor's values (a FIR with non-negative taps of sum 1) reuses it."""
    __slots__ = ('data', 'hdr', 'bound', 'scale_of')

    def __init__(self, data, hdr, bound=None):
        self.data, self.hdr, self.bound = data, hdr, bound
        self.scale_of = None        # the per-(image, channel) scale tensor a producer folded in (identity-checked by consumers)

    @property
    def shape(self):
        return self.data.shape


def hand(t, name, value):
    """Attach a hand-over attribute (`_rick_split`, `_rick_amax`, `_rick_bound`) to tensor `t` together with what it
    describes: the tensor's version counter and address at this moment."""
    if t.is_inference():        # inference tensors track no version counter (torch.inference_mode()): nothing is handed over,
        return                  # the consumer measures / packs again
    setattr(t, name, (value, t._version, t.data_ptr()))


def taken(t, name):
    """The attribute `hand` attached — or None when the tensor was modified in place since (x.mul_(), an in-place gradient
    accumulation, a hook): a stale image or a maximum that is too small would give finite but wrong products;
    the consumer then measures / packs again."""
    ent = getattr(t, name, None)
    if ent is None:
        return None
    value, version, address = ent
    if t.is_inference():
        return None
    return value if (t._version == version and t.data_ptr() == address) else None


def rehand(src, dst, names=('_rick_split', '_rick_amax', '_rick_bound')):
    """Carry still-valid hand-over attributes of `src` over to `dst`, an alias of the same values (a view_as output)."""
    for a in names:
        v = taken(src, a)
        if v is not None:
            hand(dst, a, v)


def amax(x, word=None):
    """max |x| folded into a device word (atomic max; a fresh zeroed word by default)."""
    if word is None:
        word = new_amax(x.device)
    xc = x if x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last) else x.contiguous()
    check(lib.rick_amax_f32(ptr(xc), xc.numel(), ptr(word), stream_ptr()), 'rick_amax_f32')
    return word


def supported(x):
    return x.dim() == 4 and x.shape[1] % 4 == 0 and x.dtype == torch.float32 and x.is_cuda


def split_pack(x, amax0=None, amax1=None, coef=1.0):
    """Stand-alone fp32 -> split image pass (layers whose producer is not fused; tests).  Without `amax0` the exact maximum
    is measured first (one extra read of x)."""
    if not supported(x):
        raise RuntimeError('split_pack: needs a CUDA float32 [N, C, H, W] tensor with C % 4 == 0')
    x = x.contiguous(memory_format=torch.channels_last)
    if amax0 is None:
        amax0 = amax(x)
    n, c, h, w = x.shape
    data = torch.empty_like(x)
    hdr = new_words(4, x.device)
    check(lib.rick_split_pack_f32(ptr(x), ptr(data), ptr(hdr), ptr(amax0), ptr(amax1), float(coef), n * h * w, c, stream_ptr()),
          'rick_split_pack_f32')
    return SplitImage(data, hdr, (amax0, amax1, float(coef)))


def split_unpack(si):
    """(hi + lo) * 2^-e as an fp32 channels-last tensor (tests)."""
    n, c, h, w = si.data.shape
    out = torch.empty_like(si.data)
    check(lib.rick_split_unpack_f32(ptr(si.data), ptr(si.hdr), ptr(out), n * h * w, c, stream_ptr()), 'rick_split_unpack_f32')
    return out
