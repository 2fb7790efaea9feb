"""SURVEY §8f row 4 — the data path: PNG decode, LMDB read, the reference loader's order / flips, and the device batch
kernel (dataset.py:8-40, prepare_data.py:42-64, train_dynamic_update_prune.py:789-843) on a 10-image synthetic set."""
import numpy as np
import pytest
import torch
from torch.utils import data as tdata

from rick_amd import data as rd
from tests.lmdb_fixture import write_lmdb


def synth_images(n=10, size=32, seed=3):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:size, 0:size]
    imgs = []
    for i in range(n):     # smooth gradients + noise: exercises every PNG predictor
        base = np.stack([(xx * (i + 1) + yy * 3) % 256, (yy * (i + 2)) % 256, (xx + yy + 17 * i) % 256], -1)
        imgs.append(((base + rs.randint(0, 32, base.shape)) % 256).astype(np.uint8))
    return np.stack(imgs)


@pytest.mark.parametrize('ft', [0, 1, 2, 3, 4])
def test_png_roundtrip_all_filters(ft):
    img = synth_images(1, 37)[0][:, :29]                         # odd sizes
    blob = rd.encode_png(img, ft)
    assert np.array_equal(rd.decode_png(blob), img)


def test_png_rejects_garbage():
    with pytest.raises(ValueError):
        rd.decode_png(b'not a png at all')


def test_lmdb_reader_on_dataset_layout(tmp_path):
    """prepare_data.py's layout: keys str(i).zfill(6) -> PNG bytes (overflow pages), 'length' -> count; plus enough small
    keys to force a branch page above several leaves."""
    imgs = synth_images(10, 32)
    items = {str(i).zfill(6).encode(): rd.encode_png(imgs[i], i % 5) for i in range(10)}
    items[b'length'] = b'10'
    items.update({f'k{j:05d}'.encode(): (f'value-{j}' * (1 + j % 7)).encode() for j in range(400)})
    write_lmdb(str(tmp_path / 'db'), items)
    env = rd.LmdbReader(str(tmp_path / 'db'))
    assert env.depth == 2 and env.entries == len(items)
    for k, v in items.items():
        assert env.get(k) == v, k
    assert env.get(b'000010') is None and env.get(b'zzz') is None and env.get(b'') is None
    assert list(env.items()) == sorted(items.items())
    got = rd.load_images(str(tmp_path / 'db'), size=32)
    assert got.dtype == np.uint8 and np.array_equal(got, imgs)
    # raw tensor file and PNG folder sources give the same tensor
    rd.save_raw(str(tmp_path / 'set.u8'), imgs)
    assert np.array_equal(rd.load_images(str(tmp_path / 'set.u8')), imgs)
    (tmp_path / 'png').mkdir()
    for i in range(10):
        (tmp_path / 'png' / f'{i:03d}.png').write_bytes(rd.encode_png(imgs[i], 4))
    assert np.array_equal(rd.load_images(str(tmp_path / 'png'), size=32), imgs)
    big = np.zeros((40, 36, 3), np.uint8)
    big[4:36, 2:34] = imgs[0]
    assert np.array_equal(rd._fit(big, 32), imgs[0])               # CenterCrop of a larger stored image
    with pytest.raises(ValueError):
        rd._fit(imgs[0], 64)


class _FlipProbe(tdata.Dataset):
    """What the reference dataset does per item as far as RNG goes: one torch.rand(1) < 0.5 draw (RandomHorizontalFlip)."""

    def __len__(self):
        return 10

    def __getitem__(self, i):
        return i, bool(torch.rand(1) < 0.5)


@pytest.mark.parametrize('workers', [0, 2])
def test_loader_schedule_equals_torch_dataloader(workers):
    """Index order (RandomSampler, drop_last, endless re-iteration) and flip draws (per-worker generators seeded
    base_seed + worker_id) equal what torch.utils.data.DataLoader itself produces for the same torch.manual_seed."""
    def reference(n_batches):
        torch.manual_seed(1)                                         # train_dynamic_update_prune.py:760
        ds = _FlipProbe()
        loader = tdata.DataLoader(ds, batch_size=4, sampler=tdata.RandomSampler(ds), num_workers=workers, drop_last=True)
        out = []
        while len(out) < n_batches:                                  # sample_data (:76-79)
            for idx, flip in loader:
                out.append((idx.tolist(), [bool(f) for f in flip]))
        return out[:n_batches]
    ref = reference(7)
    torch.manual_seed(1)
    sched = rd.loader_schedule(10, 4, True, num_workers=workers)
    got = [next(sched) for _ in range(7)]
    assert [g[0] for g in got] == [r[0] for r in ref]
    assert [g[1] for g in got] == [r[1] for r in ref]


@pytest.mark.gpu
def test_device_batch_bit_exact():
    imgs = synth_images(10, 64)
    ds = rd.DeviceDataset(imgs, 'cuda')
    idx, flip = [7, 0, 3, 3, 9], [True, False, False, True, True]
    got = ds.batch(idx, flip).cpu()
    t = torch.from_numpy(imgs[idx]).permute(0, 3, 1, 2).float().div(255)     # ToTensor
    t = torch.stack([ti.flip(-1) if f else ti for ti, f in zip(t, flip)])    # RandomHorizontalFlip (before ToTensor: same pixels)
    ref = t.sub(0.5).div(0.5)                                                # Normalize((0.5,)*3, (0.5,)*3)
    assert got.shape == (5, 3, 64, 64) and torch.equal(got, ref)
    with pytest.raises(IndexError):
        ds.batch([10])
    torch.manual_seed(1)
    stream = rd.train_batches(ds, 4, num_workers=8)
    b0, b1 = next(stream), next(stream)
    assert b0.shape == (4, 3, 64, 64) and float(b0.abs().max()) <= 1.0 and not torch.equal(b0, b1)


# ------------------------------------------------------------- independent producer / consumer: PIL (Pillow 12.2)
@pytest.mark.parametrize('mode', ['RGB', 'L', 'RGBA', 'P'])
@pytest.mark.parametrize('size', [(29, 37), (64, 64)])
def test_png_decoder_reads_pil_written_files(mode, size):
    """decode_png against PNG files written by an INDEPENDENT encoder (Pillow — the library prepare_data.py:14-21 uses to
    write the LMDB values): RGB, grayscale, RGBA and palette images; PIL picks its own per-row filters and zlib level."""
    import io
    Image = pytest.importorskip('PIL.Image')
    w, h = size
    rgb = synth_images(1, max(w, h), seed=7)[0][:h, :w]
    if mode == 'RGB':
        im = Image.fromarray(rgb, 'RGB')
    elif mode == 'L':
        im = Image.fromarray(rgb[..., 0], 'L')
    elif mode == 'RGBA':
        a = ((rgb[..., :1].astype(np.int32) * 3 + 11) % 256).astype(np.uint8)
        im = Image.fromarray(np.concatenate([rgb, a], -1), 'RGBA')
    else:
        im = Image.fromarray(rgb, 'RGB').quantize(colors=64)
    buf = io.BytesIO()
    im.save(buf, format='PNG', optimize=(mode == 'L'))
    got = rd.decode_png(buf.getvalue())
    want = np.asarray(im.convert('RGB'))           # the loader hands RGB to the network (dataset.py:33-38)
    assert got.shape == want.shape and got.dtype == np.uint8
    assert np.array_equal(got, want)


@pytest.mark.parametrize('ft', [0, 1, 2, 3, 4])
def test_pil_reads_png_encoder_output(ft):
    """encode_png's files opened by an independent decoder (Pillow)."""
    import io
    Image = pytest.importorskip('PIL.Image')
    img = synth_images(1, 41, seed=9)[0][:, :33]
    back = np.asarray(Image.open(io.BytesIO(rd.encode_png(img, ft))).convert('RGB'))
    assert np.array_equal(back, img)
