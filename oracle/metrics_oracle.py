"""CPU oracle (numpy / torch-CPU) for the metric + loss kernels -- TEST INFRASTRUCTURE ONLY.

  postprocess_u8, masked_rmse : restate /root/reference/CODON_X4/test.py:127-132 and :148-164
  ssim_exact                  : restates /root/reference/CODON_X4/ssim_2.py:36-52 with the Gaussian filter written
                                out (scipy.ndimage.gaussian_filter(sd): radius int(4*sd+0.5) = 6, weights
                                exp(-x^2/(2 sd^2)) normalised, mode 'reflect' = numpy 'symmetric' padding)
Pinned by tests/golden/metrics_*.npz: ssim values produced by the IMPORTED reference ssim_2.ssim_exact on crops of
the shipped Middlebury PNGs, RMSE values by a literal transcription of the reference's Python loop
(tools/make_golden.py), and the dataset-level means 1.778 / 3.479 / 5.803 of SURVEY.md section 6.
"""
import math

import numpy as np
import torch


def postprocess_u8(x):
    """out = np.clip(out, 0, 1); out = (out * 255).astype(np.uint8)   (test.py:130-132; float32 product)."""
    x = np.asarray(x)
    if x.dtype != np.float16:                       # fp16 arrays keep their dtype, exactly as in the reference script
        x = x.astype(np.float32)
    return (np.clip(x, 0, 1) * 255).astype(np.uint8)


def masked_rmse(label, out):
    """EvaluationResults (test.py:148-164), vectorised; integer arithmetic is exact."""
    label = np.asarray(label)[:out.shape[0], :out.shape[1]].astype(np.int64)
    out = np.asarray(out).astype(np.int64)
    m = label != 0
    return math.sqrt(float(((label - out)[m] ** 2).sum()) / float(m.sum()))


def masked_rmse_loop(label, out):
    """Literal transcription of the reference loop (float64), for small crops."""
    dh = np.asarray(label).astype(np.float64)[:out.shape[0], :out.shape[1]]
    o = np.asarray(out).astype(np.float64)
    mn = dh.size
    e = np.zeros(dh.shape)
    for i in range(dh.shape[0]):
        for j in range(dh.shape[1]):
            if dh[i][j] == 0:
                mn -= 1
            else:
                e[i][j] = dh[i][j] - o[i][j]
    return math.sqrt((e ** 2).sum() / mn)


def gauss_weights(sd=1.5):
    r = int(4.0 * sd + 0.5)
    x = np.arange(-r, r + 1, dtype=np.float64)
    w = np.exp(-0.5 * x * x / (sd * sd))
    return w / w.sum()


def gaussian_filter(img, sd=1.5):
    w = gauss_weights(sd)
    r = (len(w) - 1) // 2
    out = np.asarray(img, dtype=np.float64)
    for axis in (0, 1):
        pad = [(0, 0), (0, 0)]
        pad[axis] = (r, r)
        p = np.pad(out, pad, mode="symmetric")
        acc = np.zeros_like(out)
        n = out.shape[axis]
        for k in range(len(w)):
            sl = [slice(None), slice(None)]
            sl[axis] = slice(k, k + n)
            acc += w[k] * p[tuple(sl)]
        out = acc
    return out


def ssim_exact(img1, img2, sd=1.5, C1=0.01 ** 2, C2=0.03 ** 2):
    mu1, mu2 = gaussian_filter(img1, sd), gaussian_filter(img2, sd)
    s1 = gaussian_filter(img1 * img1, sd) - mu1 * mu1
    s2 = gaussian_filter(img2 * img2, sd) - mu2 * mu2
    s12 = gaussian_filter(img1 * img2, sd) - mu1 * mu2
    return float(np.mean(((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))))


def _reflect_index(n, r):
    i = np.arange(-r, n + r)
    while ((i < 0) | (i >= n)).any():
        i = np.where(i < 0, -1 - i, i)
        i = np.where(i >= n, 2 * n - 1 - i, i)
    return torch.from_numpy(i)


def ssim_torch(a, b, sd=1.5, C1=0.01 ** 2, C2=0.03 ** 2):
    """Differentiable float64 SSIM of (B,1,H,W) tensors (same definition), for gradient checks."""
    w = torch.from_numpy(gauss_weights(sd)).to(a.dtype)
    r = (len(w) - 1) // 2
    H, W = a.shape[-2:]
    iy, ix = _reflect_index(H, r), _reflect_index(W, r)

    def G(x):
        xp = x[..., iy, :]
        x = sum(w[k] * xp[..., k:k + H, :] for k in range(len(w)))
        xp = x[..., :, ix]
        return sum(w[k] * xp[..., :, k:k + W] for k in range(len(w)))

    mu1, mu2 = G(a), G(b)
    s1, s2, s12 = G(a * a) - mu1 * mu1, G(b * b) - mu2 * mu2, G(a * b) - mu1 * mu2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).mean()
