"""``EmbeddingSpaceEvaluator`` drop-in (interface of ``scripts/model/ted_evaluator.py:12-152``): FGD, feature distance and
diversity of generated clips in the latent space of the gesture auto-encoder.

Features come from the gfx950 pose encoder (``embedding_net.EmbeddingNet`` -> ``ls_eval_features``).  The statistics are this
package's own formulation, pinned to the scores the reference's evaluator produces (tests/golden/eval_ted_golden.npz):

  * mean / covariance are ACCUMULATED as batches arrive -- per batch (count, mean, centred scatter matrix) in float64, merged with
    the parallel-variance update (Chan, Golub, LeVeque) -- instead of stacking every feature and calling ``np.cov`` at the end;
  * the Frechet distance ||mu1 - mu2||^2 + Tr(S1) + Tr(S2) - 2 Tr((S1 S2)^(1/2)) takes the trace of the matrix square root from
    symmetric eigendecompositions only: with S1 = V diag(w) V^T, (S1 S2)^(1/2) is similar to (S1^(1/2) S2 S1^(1/2))^(1/2), a
    symmetric PSD matrix, so the trace is the sum of the square roots of its eigenvalues (``numpy.linalg.eigh`` twice; no general
    ``sqrtm``, no complex arithmetic, no "imaginary component" failure mode).

``embed_net_path`` is loaded like the reference does (``ckpt['pose_dim']``, ``ckpt['gen_dict']``) but is a required argument
(the reference hard-codes a path).
"""
from __future__ import annotations

import numpy as np
import torch

from .embedding_net import EmbeddingNet


class _Moments:
    """Running (n, mean, scatter) of row vectors; scatter = sum (x - mean)(x - mean)^T, all float64."""

    def __init__(self):
        self.n, self.mean, self.scatter = 0, None, None

    def add(self, rows: np.ndarray):
        x = np.asarray(rows, dtype=np.float64)
        if x.ndim != 2 or x.shape[0] == 0:
            return
        nb, mb = x.shape[0], x.mean(axis=0)
        xc = x - mb
        sb = xc.T @ xc
        if self.n == 0:
            self.n, self.mean, self.scatter = nb, mb, sb
            return
        nt = self.n + nb
        delta = mb - self.mean
        self.scatter = self.scatter + sb + np.outer(delta, delta) * (self.n * nb / nt)
        self.mean = self.mean + delta * (nb / nt)
        self.n = nt

    def covariance(self) -> np.ndarray:
        """Unbiased (n - 1), what ``np.cov(rows, rowvar=False)`` returns."""
        return self.scatter / max(self.n - 1, 1)


def _trace_sqrt_product(s1: np.ndarray, s2: np.ndarray, eps: float) -> float:
    """Tr((S1 S2)^(1/2)) for symmetric PSD S1, S2 through two symmetric eigendecompositions."""
    w, v = np.linalg.eigh((s1 + s1.T) * 0.5)
    root1 = (v * np.sqrt(np.clip(w, 0.0, None))) @ v.T                       # S1^(1/2)
    inner = root1 @ ((s2 + s2.T) * 0.5) @ root1
    lam = np.linalg.eigvalsh((inner + inner.T) * 0.5)
    if not np.isfinite(lam).all():                                           # degenerate input: regularise like the reference's eps offset
        off = np.eye(s1.shape[0]) * eps
        return _trace_sqrt_product(s1 + off, s2 + off, eps * 10)
    return float(np.sqrt(np.clip(lam, 0.0, None)).sum())


class EmbeddingSpaceEvaluator:
    def __init__(self, embed_net_path=None, device="cuda:0", ckpt=None):
        if ckpt is None:
            if embed_net_path is None:
                raise ValueError("embed_net_path (gesture_autoencoder_checkpoint_best.bin) or ckpt= is required")
            ckpt = torch.load(embed_net_path, map_location="cpu")
        self.pose_dim = ckpt['pose_dim']
        self.net = EmbeddingNet(self.pose_dim, 34).to(device)
        self.net.load_state_dict(ckpt['gen_dict'])
        self.net.train(False)
        self.net.freeze_pose_nets()
        self.reset()

    def reset(self):
        self.real_feat_list, self.generated_feat_list = [], []      # per pushed batch (diversity pairs whole batches)
        self.recon_err_diff = []
        self._real, self._gen = _Moments(), _Moments()
        self._abs_diff_sum, self._rows = 0.0, 0

    def push_samples(self, generated_poses, real_poses):
        feats = []
        for poses in (generated_poses, real_poses):
            f, _, _ = self.net(poses, variational_encoding=False)
            feats.append(f.detach().cpu().numpy())
        g, r = feats
        self.generated_feat_list.append(g)
        self.real_feat_list.append(r)
        self._gen.add(g)
        self._real.add(r)
        self._abs_diff_sum += float(np.abs(r.astype(np.float64) - g).sum())
        self._rows += g.shape[0]

    def get_no_of_samples(self):
        return len(self.real_feat_list)

    def get_scores(self):
        """(frechet_dist, feat_dist): FGD between the generated and the real feature clouds, and the mean L1 distance of pairs."""
        if self._rows == 0:
            raise ValueError("no samples pushed")
        fgd = self.calculate_frechet_distance(self._gen.mean, self._gen.covariance(), self._real.mean, self._real.covariance())
        return fgd, self._abs_diff_sum / self._rows

    @staticmethod
    def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
        mu1, mu2 = np.atleast_1d(np.asarray(mu1, np.float64)), np.atleast_1d(np.asarray(mu2, np.float64))
        sigma1, sigma2 = np.atleast_2d(np.asarray(sigma1, np.float64)), np.atleast_2d(np.asarray(sigma2, np.float64))
        if mu1.shape != mu2.shape or sigma1.shape != sigma2.shape:
            raise AssertionError("mean / covariance shapes differ")
        if not (np.isfinite(mu1).all() and np.isfinite(mu2).all() and np.isfinite(sigma1).all() and np.isfinite(sigma2).all()):
            return float("inf")
        d = mu1 - mu2
        return float(d @ d + np.trace(sigma1) + np.trace(sigma2) - 2.0 * _trace_sqrt_product(sigma1, sigma2, eps))

    def get_diversity_scores(self):
        """Mean L1 distance between the first 500 pushed batches and a random re-pairing of the batches; the permutation is
        drawn with ``torch.randperm`` exactly where the reference draws it, so a seed reproduces the reference's score."""
        first = self.generated_feat_list[:500]
        order = torch.randperm(len(self.generated_feat_list))[:500].tolist()
        a = np.concatenate(first, axis=0).astype(np.float64)
        b = np.concatenate([self.generated_feat_list[i] for i in order], axis=0)
        return float(np.abs(a - b).sum(axis=-1).mean())
