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


class FeatureStats:
    """Streaming mean / covariance of feature vectors on the device (fp64 accumulators).

    Replaces ``calculate_activation_statistics`` (gan_training/metrics/fid_score.py:132-142), which
    gathers every activation on the host and calls ``np.mean`` / ``np.cov(rowvar=False)``: here each
    batch adds ``sum x`` and ``x^T x`` (one hipBLAS GEMM) and the statistics are finalised once —
    no [n_samples, dims] array ever exists, on either side of PCIe."""

    def __init__(self, dims, device):
        self.n = 0
        self.sum = torch.zeros(dims, device=device, dtype=torch.float64)
        self.outer = torch.zeros(dims, dims, device=device, dtype=torch.float64)

    @torch.no_grad()
    def update(self, feats):
        f = feats.reshape(feats.shape[0], -1).to(torch.float64)
        if f.shape[1] != self.sum.shape[0]:
            raise RuntimeError(f'FeatureStats: expected {self.sum.shape[0]} features, got {f.shape[1]}')
        self.sum += f.sum(0)
        self.outer += f.t() @ f
        self.n += f.shape[0]
        return self

    def finalize(self):
        """-> (mu[dims], sigma[dims, dims]) with np.cov's default normalisation (n - 1)."""
        if self.n < 2:
            raise RuntimeError('FeatureStats: need at least two samples')
        mu = self.sum / self.n
        sigma = (self.outer - self.n * torch.outer(mu, mu)) / (self.n - 1)
        return mu, sigma


@torch.no_grad()
def frechet_distance(mu1, sigma1, mu2, sigma2):
    """d^2 = |mu1 - mu2|^2 + tr(S1) + tr(S2) - 2 tr sqrt(S1 S2)   (fid_score.py:94-129), on the device.

    The reference takes scipy's ``sqrtm`` of the non-symmetric product and keeps the real part of its
    trace.  S1 S2 is similar to the symmetric PSD matrix S1^(1/2) S2 S1^(1/2), so the same trace is the sum
    of the square roots of that matrix's eigenvalues: two symmetric eigen-decompositions in fp64, no
    host round trip, and no singular-product fallback is needed (negative round-off eigenvalues clamp to 0)."""
    mu1, mu2 = mu1.to(torch.float64), mu2.to(torch.float64)
    s1, s2 = sigma1.to(torch.float64), sigma2.to(torch.float64)
    if mu1.shape != mu2.shape or s1.shape != s2.shape:
        raise RuntimeError('frechet_distance: statistics have different shapes')
    w, u = torch.linalg.eigh((s1 + s1.t()) * 0.5)
    s1h = (u * w.clamp_min(0).sqrt()) @ u.t()
    m = s1h @ s2 @ s1h
    lam = torch.linalg.eigvalsh((m + m.t()) * 0.5)
    tr_covmean = lam.clamp_min(0).sqrt().sum()
    diff = mu1 - mu2
    return diff.dot(diff) + torch.trace(s1) + torch.trace(s2) - 2 * tr_covmean


@torch.no_grad()
def fid_from_generator(g_ema, real_stats, feature_fn, n_sample_test=5000, n_sample_store=25, latent=512,
                       generator=None, latents=None):
    """BASELINE config 4 end to end on the device: sample ``n_sample_test`` images in batches of
    ``n_sample_store`` (gan_training/eval.py:31-46), push each batch through ``feature_fn`` (the Inception
    pool3 plug point: weights are supplied by the user, they are a download in the reference), accumulate
    the statistics, and return the Frechet distance to ``real_stats = (mu, sigma)``."""
    dev = next(g_ema.parameters()).device
    stats = None
    was_training = g_ema.training
    g_ema.eval()
    done = 0
    while done < n_sample_test:
        if latents is not None:
            z = latents[done:done + n_sample_store].to(dev)
        else:
            z = torch.randn(n_sample_store, latent, device=dev, generator=generator)
        img, _ = g_ema([z])
        img = img[:n_sample_test - done]
        f = feature_fn(img)
        if stats is None:
            stats = FeatureStats(f.reshape(f.shape[0], -1).shape[1], dev)
        stats.update(f)
        done += img.shape[0]
    if was_training:
        g_ema.train()
    mu, sigma = stats.finalize()
    return frechet_distance(mu, sigma, real_stats[0].to(dev), real_stats[1].to(dev))


@torch.no_grad()
def kid_from_features(codes_g, codes_r, n_subsets=100, subset_size=1000, degree=3, gamma=None, coef0=1, rng=None):
    """Kernel Inception Distance on the device: the unbiased polynomial-kernel MMD^2 of ``n_subsets`` random subsets
    (gan_metrics/kid_score.py:255-290 with the evaluator's arguments, :391-393) -> (mean, std, mmds[n_subsets]).

    The reference forms three subset_size^2 kernel matrices per subset with sklearn on the host; here each subset is
    three fp64 GEMMs + elementwise powers on the device.  Subset indices are drawn exactly like the reference does
    (``np.random.choice(n, subset_size, replace=False)``, generator first, then real; pass ``rng`` = a
    ``np.random.RandomState`` to make a run reproducible) so a seeded run reproduces the reference's subsets."""
    import numpy as np
    choice = (rng or np.random).choice
    xg, xr = codes_g.to(torch.float64), codes_r.to(torch.float64)
    if xg.shape[1] != xr.shape[1]:
        raise RuntimeError('kid_from_features: feature dimensions differ')
    if subset_size > min(xg.shape[0], xr.shape[0]):
        raise RuntimeError('kid_from_features: subset_size exceeds the number of samples')
    gam = 1.0 / xg.shape[1] if gamma is None else gamma
    m = subset_size
    mmds = torch.empty(n_subsets, dtype=torch.float64, device=xg.device)
    for i in range(n_subsets):
        ig = torch.as_tensor(choice(xg.shape[0], subset_size, replace=False), device=xg.device)
        ir = torch.as_tensor(choice(xr.shape[0], subset_size, replace=False), device=xr.device)
        g, r = xg[ig], xr[ir]
        k_xx = (gam * (g @ g.t()) + coef0) ** degree
        k_yy = (gam * (r @ r.t()) + coef0) ** degree
        k_xy = (gam * (g @ r.t()) + coef0) ** degree
        kt_xx = k_xx.sum() - torch.diagonal(k_xx).sum()          # off-diagonal sums
        kt_yy = k_yy.sum() - torch.diagonal(k_yy).sum()
        mmds[i] = (kt_xx + kt_yy) / (m * (m - 1)) - 2 * k_xy.sum() / (m * m)
    return mmds.mean(), mmds.std(unbiased=False), mmds


@torch.no_grad()
def precision_recall_from_features(feats_real, feats_fake, k=3, block=4096):
    """Improved precision / recall (gan_metrics/precision_recall.py:50-66, 185-246) on the device.

    Each set defines a manifold = union of balls around its features with radius = distance to the k-th nearest
    neighbour in the same set; precision = share of fake features inside the real manifold, recall = share of real
    features inside the fake one.  The reference builds the N x N distance matrices with NumPy on the host and loops
    over rows; here they are fp64 GEMM blocks on the device (`block` columns at a time) reduced with kthvalue / any.
    Returns (precision, recall) as 0-dim device tensors."""
    xr, xf = feats_real.to(torch.float64), feats_fake.to(torch.float64)
    if xr.shape[1] != xf.shape[1]:
        raise RuntimeError('precision_recall_from_features: feature dimensions differ')
    if min(xr.shape[0], xf.shape[0]) <= k:
        raise RuntimeError('precision_recall_from_features: need more than k samples per set')

    def dist(a, b):                       # [len(a), len(b)] Euclidean distances, negative round-off clamped like the reference
        d2 = (a * a).sum(1, keepdim=True) - 2 * (a @ b.t()) + (b * b).sum(1).unsqueeze(0)
        return d2.clamp_min(0).sqrt()

    def radii(x):                         # k-th neighbour = (k+1)-th smallest of the row (the closest one is the point itself)
        out = torch.empty(x.shape[0], dtype=torch.float64, device=x.device)
        for lo in range(0, x.shape[0], block):
            out[lo:lo + block] = dist(x[lo:lo + block], x).kthvalue(k + 1, dim=1).values
        return out

    def covered(ref, ref_radii, subj):    # share of subjects inside at least one ball of the reference manifold
        hit = 0
        for lo in range(0, subj.shape[0], block):
            hit = hit + (dist(ref, subj[lo:lo + block]) < ref_radii.unsqueeze(1)).any(0).sum()
        return hit.to(torch.float64) / subj.shape[0]
    return covered(xr, radii(xr), xf), covered(xf, radii(xf), xr)


class Evaluator:
    """Device-resident counterpart of ``gan_training.eval.Evaluator`` (eval.py:13-66).

    The reference keeps the real images in ``real_imgs.npy``, generates ``inception_nsamples`` fakes onto the host and
    hands both image sets to three separate metric modules, each of which runs its own Inception / VGG forward pass.
    Here the real side is reduced ONCE to features (``real_feats [n, dims]``, any device tensor), every generated batch
    goes through ``feature_fn`` while it is still on the GPU, and FID / KID / precision-recall are all computed from the
    same two feature matrices.  ``feature_fn`` is the plug point for the pretrained networks (weights are downloads in
    the reference; the reference uses Inception pool3 for FID / KID and VGG-16 fc2 for precision-recall — pass
    ``pr_feature_fn`` / ``real_pr_feats`` to keep that split)."""

    def __init__(self, generator, feature_fn, real_feats, n_sample_store=25, latent=512, inception_nsamples=5000,
                 fid_sample_size=5000, pr_feature_fn=None, real_pr_feats=None, k=3):
        self.generator, self.feature_fn, self.real_feats = generator, feature_fn, real_feats
        self.n_sample_store, self.latent = n_sample_store, latent
        self.inception_nsamples, self.sample_size, self.k = inception_nsamples, fid_sample_size, k
        self.pr_feature_fn, self.real_pr_feats = pr_feature_fn, real_pr_feats

    @torch.no_grad()
    def compute_inception_score(self, fid=True, kid=False, pr=False, latents=None, kid_subsets=100, kid_subset_size=1000,
                                rng=None):
        """-> dict with 'fid', 'kid', 'precision', 'recall' (the keys the reference fills, eval.py:44-66)."""
        g = self.generator
        dev = next(g.parameters()).device
        was_training = g.training
        g.eval()
        feats, pr_feats, done = [], [], 0
        while done < self.inception_nsamples:                       # eval.py:34-41: batches of n_sample_store
            if latents is not None:
                z = latents[done:done + self.n_sample_store].to(dev)
            else:
                z = torch.randn(self.n_sample_store, self.latent, device=dev)
            img, _ = g([z])
            feats.append(self.feature_fn(img).reshape(img.shape[0], -1))
            if pr and self.pr_feature_fn is not None:
                pr_feats.append(self.pr_feature_fn(img).reshape(img.shape[0], -1))
            done += img.shape[0]
        if was_training:
            g.train()
        fake = torch.cat(feats, 0)[:self.sample_size]
        real = self.real_feats.to(dev)
        score = {}
        if fid:
            st_r, st_f = FeatureStats(real.shape[1], dev).update(real), FeatureStats(fake.shape[1], dev).update(fake)
            score['fid'] = frechet_distance(*st_r.finalize(), *st_f.finalize())
        if kid:                                                     # eval.py:52-54: the first 2000 of each side
            score['kid'] = kid_from_features(real[:2000], fake[:2000], n_subsets=kid_subsets, subset_size=kid_subset_size,
                                             rng=rng)[0]
        if pr:
            fr = self.real_pr_feats.to(dev) if self.real_pr_feats is not None else real
            ff = torch.cat(pr_feats, 0)[:self.sample_size] if pr_feats else fake
            score['precision'], score['recall'] = precision_recall_from_features(fr, ff, k=self.k)
        return score
