"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the FID statistics the reference
computes around its Inception network (gan_training/metrics/fid_score.py).  Pinned against
tests/golden/fid.npz, which tools/make_golden.py produces by running the reference's own
``calculate_frechet_distance`` / ``np.cov`` path in the build container."""
import numpy as np
from scipy import linalg


def activation_statistics_ref(act):
    """fid_score.py:138-142: mu = mean over samples, sigma = np.cov(act, rowvar=False)."""
    act = np.asarray(act, dtype=np.float64)
    return np.mean(act, axis=0), np.cov(act, rowvar=False)


def frechet_distance_ref(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """fid_score.py:94-129: |mu1-mu2|^2 + tr S1 + tr S2 - 2 tr sqrtm(S1 S2), with the reference's eps-offset retry
    for a singular product and its check on the imaginary part."""
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError('Imaginary component {}'.format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


def polynomial_mmd2_ref(codes_g, codes_r, degree=3, gamma=None, coef0=1):
    """Unbiased MMD^2 with k(x, y) = (gamma <x, y> + coef0)^degree (gan_metrics/kid_score.py:276-346, mmd_est='unbiased')."""
    x, y = np.asarray(codes_g, dtype=np.float64), np.asarray(codes_r, dtype=np.float64)
    gam = 1.0 / x.shape[1] if gamma is None else gamma
    k_xx, k_yy, k_xy = (gam * x @ x.T + coef0) ** degree, (gam * y @ y.T + coef0) ** degree, (gam * x @ y.T + coef0) ** degree
    m = x.shape[0]
    return ((k_xx.sum() - np.trace(k_xx)) + (k_yy.sum() - np.trace(k_yy))) / (m * (m - 1)) - 2 * k_xy.sum() / (m * m)


def kid_ref(codes_g, codes_r, n_subsets, subset_size, rng):
    """polynomial_mmd_averages (:255-273): subsets drawn generator-first with rng.choice(n, subset_size, replace=False)."""
    mmds = np.zeros(n_subsets)
    for i in range(n_subsets):
        g = codes_g[rng.choice(len(codes_g), subset_size, replace=False)]
        r = codes_r[rng.choice(len(codes_r), subset_size, replace=False)]
        mmds[i] = polynomial_mmd2_ref(g, r)
    return mmds


def precision_recall_ref(feats_real, feats_fake, k=3):
    """gan_metrics/precision_recall.py:50-66,185-246: k-NN ball manifolds of both sets, precision = fake-in-real,
    recall = real-in-fake."""
    def dist(a, b):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        d2 = (a ** 2).sum(1, keepdims=True) - 2 * a @ b.T + (b ** 2).sum(1, keepdims=True).T
        return np.sqrt(np.maximum(d2, 0))

    def radii(x):
        return np.sort(dist(x, x), axis=1)[:, k]

    def metric(ref, ref_radii, subj):
        return float((dist(ref, subj) < ref_radii[:, None]).any(0).mean())
    return metric(feats_real, radii(feats_real), feats_fake), metric(feats_fake, radii(feats_fake), feats_real)
