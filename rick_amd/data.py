"""Data path of the RICK loop: LMDB / PNG -> device-resident dataset -> normalised batches.

The reference reads PNG-encoded images from an LMDB environment written by ``prepare_data.py`` (keys
``str(index).zfill(6)`` plus ``length``, prepare_data.py:42-64, dataset.py:8-40), decodes them with PIL in eight
DataLoader workers and applies ``Resize -> CenterCrop -> RandomHorizontalFlip -> ToTensor -> Normalize(0.5, 0.5)``
(train_dynamic_update_prune.py:789-843).  Neither ``lmdb`` nor PIL / torchvision exist in this image, and on an MI355X a
10-shot training set (1.9 MB as uint8) or even the 5 000-image test set (983 MB) is a rounding error of 288 GB of HBM, so
the design is:

  * ``LmdbReader``   — read-only B+tree walk over ``data.mdb`` (LMDB's on-disk format: two meta pages, branch / leaf /
                       overflow pages; default byte-wise key order), enough for ``get(key)`` and ordered iteration;
  * ``decode_png``   — PNG (8-bit gray / RGB / RGBA / palette, non-interlaced): zlib inflate + the native scanline
                       reconstruction ``rick_png_unfilter`` (librick_hip.so, host code);
  * ``load_images``  — an LMDB directory, a folder of ``.png`` files or a raw ``.u8`` tensor file (see ``save_raw``) ->
                       uint8 ``[N, H, W, 3]``; stored images must already be ``size x size`` (what prepare_data.py writes;
                       PIL's resampling filters are not reproduced — larger images are centre-cropped, smaller ones are an error);
  * ``DeviceDataset``— the uint8 tensor resident on the GPU + ``rick_image_batch_f32``: one launch gathers a batch, flips
                       and normalises it into ``[B, 3, H, W]`` fp32;
  * ``train_batches`` / ``test_batches`` — the reference's loader ORDER: ``RandomSampler`` / ``SequentialSampler`` ->
                       ``BatchSampler(drop_last=True)`` -> endless ``sample_data`` loop, and the horizontal flips drawn
                       from per-worker generators seeded ``base_seed + worker_id`` exactly as ``torch.utils.data``'s worker
                       loop seeds them (batch k is produced by worker k % num_workers; ``torch.rand(1) < 0.5`` per sample).
                       With the same ``torch.manual_seed`` the index order and the flip pattern equal the reference's.
"""
import ctypes
import os
import struct
import zlib

import numpy as np
import torch
from torch.utils import data as tdata

from ._lib import check, lib, ptr, stream_ptr

# ----------------------------------------------------------------------------------------------- LMDB (read-only)
_P_BRANCH, _P_LEAF, _P_OVERFLOW, _P_META, _P_LEAF2 = 0x01, 0x02, 0x04, 0x08, 0x20
_F_BIGDATA, _F_SUBDATA, _F_DUPDATA = 0x01, 0x02, 0x04
_MDB_MAGIC = 0xBEEFC0DE
_PAGEHDR = 16
_INVALID = (1 << 64) - 1


class LmdbReader:
    """Minimal reader of an LMDB environment's main database (64-bit little-endian layout, as written on x86-64 by
    liblmdb 0.9.x / py-lmdb, environment.yml:68).  ``path`` is the environment directory (``data.mdb`` inside) or the
    file itself.  Sub-databases and duplicate-sorted databases are not supported (the dataset uses neither)."""

    def __init__(self, path):
        if os.path.isdir(path):
            path = os.path.join(path, 'data.mdb')
        self.buf = np.memmap(path, dtype=np.uint8, mode='r')
        metas = []
        psize = 4096
        for pg in range(2):
            off = pg * psize
            if off + _PAGEHDR + 136 > self.buf.size:
                break
            flags = struct.unpack_from('<H', self.buf, off + 10)[0]
            magic, version = struct.unpack_from('<II', self.buf, off + _PAGEHDR)
            if not (flags & _P_META) or magic != _MDB_MAGIC:
                if pg == 0:
                    raise IOError(f'{path}: not an LMDB data file')
                continue
            m = _PAGEHDR + 8 + 16                                   # magic, version, address, mapsize
            free_pad = struct.unpack_from('<I', self.buf, off + m)[0]   # mm_dbs[0].md_pad doubles as the page size
            main = struct.unpack_from('<IHHQQQQQ', self.buf, off + m + 48)
            last_pg, txnid = struct.unpack_from('<QQ', self.buf, off + m + 96)
            metas.append(dict(psize=free_pad, depth=main[2], entries=main[6], root=main[7], txnid=txnid, version=version))
            if pg == 0:
                psize = free_pad or 4096
        if not metas:
            raise IOError(f'{path}: no valid meta page')
        meta = max(metas, key=lambda d: d['txnid'])
        self.psize, self.root, self.entries, self.depth = meta['psize'] or 4096, meta['root'], meta['entries'], meta['depth']

    # -- page helpers
    def _page(self, pgno):
        off = pgno * self.psize
        flags, lower, upper = struct.unpack_from('<HHH', self.buf, off + 10)
        return off, flags, (lower - _PAGEHDR) // 2

    def _node(self, off, i):
        p = off + struct.unpack_from('<H', self.buf, off + _PAGEHDR + 2 * i)[0]
        lo, hi, flags, ksize = struct.unpack_from('<HHHH', self.buf, p)
        return p, lo, hi, flags, bytes(self.buf[p + 8:p + 8 + ksize])

    def _value(self, p, lo, hi, flags, ksize):
        size = lo | (hi << 16)
        if flags & (_F_SUBDATA | _F_DUPDATA):
            raise IOError('LmdbReader: sub-databases / duplicate keys are not supported')
        d = p + 8 + ksize
        if flags & _F_BIGDATA:
            ovf = struct.unpack_from('<Q', self.buf, d)[0] * self.psize
            return bytes(self.buf[ovf + _PAGEHDR:ovf + _PAGEHDR + size])
        return bytes(self.buf[d:d + size])

    def get(self, key):
        """Value of `key` (bytes) or None."""
        if self.root == _INVALID:
            return None
        pgno = self.root
        while True:
            off, flags, n = self._page(pgno)
            if flags & _P_LEAF2:
                raise IOError('LmdbReader: LEAF2 pages (fixed-size dup keys) are not supported')
            if flags & _P_BRANCH:
                lo_i, hi_i = 0, n - 1                                # last node whose key <= key; node 0 is -infinity
                while lo_i < hi_i:
                    mid = (lo_i + hi_i + 1) // 2
                    if self._node(off, mid)[4] <= key:
                        lo_i = mid
                    else:
                        hi_i = mid - 1
                _, lo, hi, fl, _ = self._node(off, lo_i)
                pgno = lo | (hi << 16) | (fl << 32)
                continue
            if not flags & _P_LEAF:
                raise IOError('LmdbReader: unexpected page type')
            lo_i, hi_i = 0, n - 1
            while lo_i <= hi_i:
                mid = (lo_i + hi_i) // 2
                p, lo, hi, fl, k = self._node(off, mid)
                if k == key:
                    return self._value(p, lo, hi, fl, len(k))
                if k < key:
                    lo_i = mid + 1
                else:
                    hi_i = mid - 1
            return None

    def items(self):
        """(key, value) pairs in key order."""
        if self.root == _INVALID:
            return

        def walk(pgno):
            off, flags, n = self._page(pgno)
            for i in range(n):
                p, lo, hi, fl, k = self._node(off, i)
                if flags & _P_BRANCH:
                    yield from walk(lo | (hi << 16) | (fl << 32))
                else:
                    yield k, self._value(p, lo, hi, fl, len(k))
        yield from walk(self.root)


# ------------------------------------------------------------------------------------------------------------ PNG
_PNG_SIG = b'\x89PNG\r\n\x1a\n'


def decode_png(blob):
    """PNG bytes -> uint8 [H, W, 3].  8-bit gray, RGB, RGBA (alpha dropped, like ``Image.convert('RGB')`` in
    prepare_data.py:36) and palette images; non-interlaced."""
    if blob[:8] != _PNG_SIG:
        raise ValueError('not a PNG stream')
    pos, idat, plte, hdr = 8, [], None, None
    while pos + 8 <= len(blob):
        n, typ = struct.unpack_from('>I4s', blob, pos)
        body = blob[pos + 8:pos + 8 + n]
        if typ == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', body)
        elif typ == b'PLTE':
            plte = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif typ == b'IDAT':
            idat.append(body)
        elif typ == b'IEND':
            break
        pos += 12 + n
    if hdr is None:
        raise ValueError('PNG without IHDR')
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 3, 6):
        raise ValueError(f'unsupported PNG (bit depth {depth}, colour type {ctype}, interlace {interlace})')
    ch = {0: 1, 2: 3, 3: 1, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b''.join(idat)), dtype=np.uint8).copy()
    stride = w * ch
    if raw.size != h * (stride + 1):
        raise ValueError('PNG: unexpected amount of image data')
    check(lib.rick_png_unfilter(raw.ctypes.data, h, stride, ch), 'rick_png_unfilter')
    px = raw.reshape(h, stride + 1)[:, 1:].reshape(h, w, ch)
    if ctype == 3:
        if plte is None:
            raise ValueError('palette PNG without PLTE')
        return plte[px[:, :, 0]]
    if ctype == 0:
        return np.repeat(px, 3, axis=2)
    return np.ascontiguousarray(px[:, :, :3])


def encode_png(img, filter_type=0):
    """uint8 [H, W, 3] -> PNG bytes (one filter type for every scanline; tools and tests)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, c = img.shape
    assert c == 3
    rows = img.reshape(h, w * 3).astype(np.int16)
    out = np.zeros((h, w * 3 + 1), dtype=np.uint8)
    out[:, 0] = filter_type
    left = np.zeros_like(rows)
    left[:, 3:] = rows[:, :-3]
    up = np.zeros_like(rows)
    up[1:] = rows[:-1]
    ul = np.zeros_like(rows)
    ul[1:, 3:] = rows[:-1, :-3]
    if filter_type == 0:
        f = rows
    elif filter_type == 1:
        f = rows - left
    elif filter_type == 2:
        f = rows - up
    elif filter_type == 3:
        f = rows - ((left + up) >> 1)
    else:
        p = left + up - ul
        pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - ul)
        pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
        f = rows - pred
    out[:, 1:] = (f & 255).astype(np.uint8)

    def chunk(typ, body):
        return struct.pack('>I', len(body)) + typ + body + struct.pack('>I', zlib.crc32(typ + body) & 0xffffffff)
    return (_PNG_SIG + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) + chunk(b'IDAT', zlib.compress(out.tobytes(), 6))
            + chunk(b'IEND', b''))


# ---------------------------------------------------------------------------------------------- dataset sources
_RAW_MAGIC = b'RICKU8v1'


def save_raw(path, images):
    """uint8 [N, H, W, 3] -> raw tensor file: 8-byte magic, int64 N, H, W, 3, then the pixels (memory-mappable)."""
    images = np.ascontiguousarray(images, dtype=np.uint8)
    with open(path, 'wb') as f:
        f.write(_RAW_MAGIC + struct.pack('<qqqq', *images.shape))
        f.write(images.tobytes())


def load_raw(path):
    with open(path, 'rb') as f:
        head = f.read(40)
    if head[:8] != _RAW_MAGIC:
        raise IOError(f'{path}: not a raw image tensor file')
    shape = struct.unpack('<qqqq', head[8:])
    return np.memmap(path, dtype=np.uint8, mode='r', offset=40, shape=shape)


def _fit(img, size):
    h, w, _ = img.shape
    if h < size or w < size:
        raise ValueError(f'image {h}x{w} is smaller than {size}: resize it first (prepare_data.py does; PIL resampling is not reproduced here)')
    y0, x0 = (h - size) // 2, (w - size) // 2          # torchvision CenterCrop: int(round((h - size) / 2.0)) == (h-size)//2 for even differences
    if (h - size) % 2 or (w - size) % 2:
        y0, x0 = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return img[y0:y0 + size, x0:x0 + size]


def load_images(path, size=256, limit=None):
    """LMDB environment (dataset.py:8-40 key layout), folder of PNGs (sorted by name, like ImageFolder's file list in
    prepare_data.py:45) or raw tensor file -> uint8 [N, size, size, 3]."""
    if os.path.isfile(path) and not path.endswith('.mdb'):
        arr = load_raw(path)
        return np.ascontiguousarray(arr[:limit] if limit else arr)
    if os.path.isdir(path) and not os.path.exists(os.path.join(path, 'data.mdb')):
        files = sorted(f for f in os.listdir(path) if f.lower().endswith('.png'))[:limit]
        return np.stack([_fit(decode_png(open(os.path.join(path, f), 'rb').read()), size) for f in files])
    env = LmdbReader(path)
    length = env.get(b'length')
    if length is None:
        raise IOError('LMDB dataset without a "length" key')
    n = int(length.decode('utf-8'))
    n = min(n, limit) if limit else n
    out = np.empty((n, size, size, 3), dtype=np.uint8)
    for i in range(n):
        blob = env.get(str(i).zfill(6).encode('utf-8'))          # dataset.py:33
        if blob is None:
            raise IOError(f'LMDB dataset: missing key {str(i).zfill(6)}')
        out[i] = _fit(decode_png(blob), size)
    return out


# ---------------------------------------------------------------------------------------------------- device side
class DeviceDataset:
    """uint8 [N, H, W, 3] resident in device memory; ``batch(index, flip)`` -> [B, 3, H, W] fp32 in [-1, 1]."""

    def __init__(self, images, device='cuda'):
        images = torch.as_tensor(np.ascontiguousarray(images))
        if images.dtype != torch.uint8 or images.ndim != 4 or images.shape[3] != 3:
            raise ValueError('DeviceDataset expects uint8 [N, H, W, 3]')
        self.images = images.to(device)
        self.n, self.h, self.w = images.shape[:3]

    def __len__(self):
        return self.n

    def batch(self, index, flip=None, out=None):
        index = torch.as_tensor(index, dtype=torch.int64)
        if int(index.min()) < 0 or int(index.max()) >= self.n:
            raise IndexError('DeviceDataset.batch: index out of range')
        B = index.numel()
        flip = torch.zeros(B, dtype=torch.uint8) if flip is None else torch.as_tensor(flip).to(torch.uint8)
        dev = self.images.device
        idx_d, flip_d = index.to(dev), flip.to(dev)
        if out is None:
            out = torch.empty((B, 3, self.h, self.w), device=dev, dtype=torch.float32)
        check(lib.rick_image_batch_f32(ptr(self.images), ptr(idx_d), ptr(flip_d), ptr(out), self.n, self.h, self.w, B, stream_ptr()),
              'rick_image_batch_f32')
        return out


class _Len:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


def loader_schedule(n, batch, shuffle, num_workers=8, flip_p=0.5, generator=None):
    """Endless generator of (index list, flip list) per batch in the reference loader's order
    (train_dynamic_update_prune.py:822-843 + sample_data :76-79): torch's own RandomSampler / SequentialSampler and
    BatchSampler(drop_last=True) produce the indices; an epoch's flips come from per-worker generators seeded
    base_seed + worker_id, batch k of the epoch going to worker k % num_workers (torch.utils.data._utils.worker)."""
    ds = _Len(n)
    sampler = tdata.RandomSampler(ds, generator=generator) if shuffle else tdata.SequentialSampler(ds)
    batches = tdata.BatchSampler(sampler, batch, drop_last=True)
    while True:                                               # sample_data: `while True: for batch in loader`
        # DataLoader.__iter__ draws the workers' base seed first, the sampler its own seed on first use
        base_seed = int(torch.empty((), dtype=torch.int64).random_(generator=generator).item())
        gens = [torch.Generator().manual_seed(base_seed + w) for w in range(num_workers)]
        for k, idx in enumerate(batches):
            # num_workers == 0: the main process's own generator draws the flips, in fetch order
            gw = gens[k % num_workers] if num_workers else generator
            yield idx, [bool(torch.rand(1, generator=gw) < flip_p) for _ in idx]


def train_batches(dataset, batch, num_workers=8, generator=None):
    """Endless stream of normalised training batches [B, 3, H, W] on the device, reference order and flips."""
    for idx, flip in loader_schedule(len(dataset), batch, True, num_workers, 0.5, generator):
        yield dataset.batch(idx, flip)


def test_batches(dataset, batch, num_workers=8, generator=None):
    for idx, flip in loader_schedule(len(dataset), batch, False, num_workers, 0.5, generator):
        yield dataset.batch(idx, flip)
