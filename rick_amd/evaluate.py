"""Evaluation-time sampling (BASELINE config 4).

The reference's ``Evaluator.compute_inception_score`` (gan_training/eval.py:31-46) draws
``n_sample_store`` latents at a time, runs ``g_ema([z])`` and moves every image to the host as
NumPy until ``n_sample_test`` images exist; FID is then computed by a third-party Inception
network (weights are a download: out of scope here).  This module keeps the
sampling loop on the device: the same batches, images written into one preallocated tensor, no
host round trips.  A feature extractor can be plugged in through ``feature_fn`` (called per
batch on device tensors) so a metric never needs the images on the host either.
"""
import torch


@torch.no_grad()
def sample_images(g_ema, n_sample_test, n_sample_store=25, latent=512, generator=None, feature_fn=None,
                  out=None, latents=None):
    """Generate ``n_sample_test`` images in batches of ``n_sample_store`` (gan_training/eval.py:34-41).

    latents: optional [n, latent] tensor of fixed z (parity runs); otherwise z ~ N(0, I) from `generator`
    (a torch.Generator) or the default device RNG.  Returns (images[n_sample_test, 3, S, S] on device,
    features or None)."""
    was_training = g_ema.training
    g_ema.eval()
    dev = next(g_ema.parameters()).device
    size = g_ema.size
    if out is None:
        out = torch.empty((n_sample_test, 3, size, size), device=dev, dtype=torch.float32)
    feats = []
    done = 0
    while done < n_sample_test:
        nb = n_sample_store
        if latents is not None:
            z = latents[done:done + nb].to(dev)
            if z.shape[0] == 0:
                raise RuntimeError('sample_images: not enough fixed latents')
        else:
            z = torch.randn(nb, latent, device=dev, generator=generator)
        img, _ = g_ema([z])
        take = min(img.shape[0], n_sample_test - done)
        out[done:done + take].copy_(img[:take])
        if feature_fn is not None:
            feats.append(feature_fn(img[:take]))
        done += take
    if was_training:
        g_ema.train()
    return out, (torch.cat(feats, 0) if feats else None)
