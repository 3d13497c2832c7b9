"""CPU restatement (numpy) of the reference's FGD evaluator path.

TEST INFRASTRUCTURE ONLY.  Follows:
  * PoseEncoderConv.forward, eval mode      scripts/model/embedding_net.py:41-83 (BEAT: scripts_beat/model/motion_autoencoder.py:38-73)
      net: Conv1d(dim,b,3)+BN+LeakyReLU(0.2) -> Conv1d(b,2b,3)+BN+LReLU -> Conv1d(2b,2b,4,stride 2)+BN+LReLU -> Conv1d(2b,b,3)
      out_net: Linear(12b,h1)+BN+LeakyReLU(True) -> Linear(h1,h2)+BN+LeakyReLU(True) -> Linear(h2,b); fc_mu
      (TED: b=32, 384->256->128->32; BEAT: 12b->4b->2b->b.  nn.LeakyReLU(True) is negative_slope = 1.0, i.e. the identity.)
  * EmbeddingSpaceEvaluator.get_scores / calculate_frechet_distance / get_diversity_scores
                                           scripts/model/ted_evaluator.py:61-152
Pinned by tests/golden/eval_golden.npz (tests/golden/make_golden_eval.py runs the reference's own classes).
"""
from __future__ import annotations

import numpy as np
from scipy import linalg


def _conv1d(x, w, b, stride):
    B, C, L = x.shape
    Co, _, K = w.shape
    Lo = (L - K) // stride + 1
    cols = np.stack([x[:, :, k:k + stride * Lo:stride] for k in range(K)], axis=-1)      # [B, C, Lo, K]
    return np.einsum("bclk,ock->bol", cols, w, optimize=True) + b[None, :, None]


def _bn(x, sd, p, axis):
    shp = [1] * x.ndim
    shp[axis] = -1
    rm, rv, g, be = (sd[p + k].reshape(shp) for k in ("running_mean", "running_var", "weight", "bias"))
    return (x - rm) / np.sqrt(rv + 1e-5) * g + be


def _lrelu(x, s):
    return np.where(x >= 0, x, s * x)


def pose_encoder(sd: dict, poses: np.ndarray, prefix="pose_encoder.") -> np.ndarray:
    """poses [B, T, dim] -> mu [B, base] (variational_encoding=False: z = mu)."""
    x = np.asarray(poses, np.float32).transpose(0, 2, 1)
    for i, stride in ((0, 1), (1, 1), (2, 2)):
        p = f"{prefix}net.{i}."
        x = _conv1d(x, sd[p + "0.weight"], sd[p + "0.bias"], stride)
        x = _lrelu(_bn(x, sd, p + "1.", 1), 0.2)
    x = _conv1d(x, sd[prefix + "net.3.weight"], sd[prefix + "net.3.bias"], 1)
    h = x.reshape(x.shape[0], -1)
    for lin, bn in ((0, 1), (3, 4)):
        h = h @ sd[f"{prefix}out_net.{lin}.weight"].T + sd[f"{prefix}out_net.{lin}.bias"]
        h = _lrelu(_bn(h, sd, f"{prefix}out_net.{bn}.", 1), 1.0)
    h = h @ sd[prefix + "out_net.6.weight"].T + sd[prefix + "out_net.6.bias"]
    return (h @ sd[prefix + "fc_mu.weight"].T + sd[prefix + "fc_mu.bias"]).astype(np.float32)


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
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


def scores(generated_feats, real_feats):
    """(frechet_dist, feat_dist) of get_scores (ted_evaluator.py:61-88)."""
    fd = calculate_frechet_distance(np.mean(generated_feats, axis=0), np.cov(generated_feats, rowvar=False),
                                    np.mean(real_feats, axis=0), np.cov(real_feats, rowvar=False))
    return fd, float(np.mean(np.sum(np.abs(real_feats - generated_feats), axis=1)))
