"""``EmbeddingSpaceEvaluator`` drop-in (``scripts/model/ted_evaluator.py:12-152``): FGD / feature distance / diversity.
Features come from the gfx950 pose encoder (``embedding_net.EmbeddingNet`` -> ``ls_eval_features``); the statistics
(mean, covariance, scipy ``sqrtm``) are the same host code as the reference's.  ``embed_net_path`` is loaded like the
reference does (``ckpt['pose_dim']``, ``ckpt['gen_dict']``) but is a required argument (the reference hard-codes a path).
"""
from __future__ import annotations

import numpy as np
import torch
from scipy import linalg

from .embedding_net import EmbeddingNet


class EmbeddingSpaceEvaluator:
    def __init__(self, embed_net_path=None, device="cuda:0", ckpt=None):
        if ckpt is None:
            if embed_net_path is None:
                raise ValueError("embed_net_path (gesture_autoencoder_checkpoint_best.bin) or ckpt= is required")
            ckpt = torch.load(embed_net_path, map_location="cpu")
        n_frames = 34
        self.pose_dim = ckpt['pose_dim']
        self.net = EmbeddingNet(self.pose_dim, n_frames).to(device)
        self.net.load_state_dict(ckpt['gen_dict'])
        self.net.train(False)
        self.net.freeze_pose_nets()
        self.reset()

    def reset(self):
        self.real_feat_list = []
        self.generated_feat_list = []
        self.recon_err_diff = []

    def push_samples(self, generated_poses, real_poses):
        real_feat, _, _ = self.net(real_poses, variational_encoding=False)
        generated_feat, _, _ = self.net(generated_poses, variational_encoding=False)
        self.real_feat_list.append(real_feat.data.cpu().numpy())
        self.generated_feat_list.append(generated_feat.data.cpu().numpy())

    def get_no_of_samples(self):
        return len(self.real_feat_list)

    def get_scores(self):
        generated_feats = np.vstack(self.generated_feat_list)
        real_feats = np.vstack(self.real_feat_list)

        def frechet_distance(samples_A, samples_B):
            try:
                return self.calculate_frechet_distance(np.mean(samples_A, axis=0), np.cov(samples_A, rowvar=False),
                                                       np.mean(samples_B, axis=0), np.cov(samples_B, rowvar=False))
            except ValueError:
                return float("inf")

        frechet_dist = frechet_distance(generated_feats, real_feats)
        feat_dist = np.mean(np.sum(np.absolute(real_feats - generated_feats), axis=1))
        return frechet_dist, feat_dist

    @staticmethod
    def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
        """d^2 = ||mu_1 - mu_2||^2 + Tr(C_1 + C_2 - 2 sqrt(C_1 C_2)) (ted_evaluator.py:91-143)."""
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

    def get_diversity_scores(self):
        feat1 = np.vstack(self.generated_feat_list[:500])
        random_idx = torch.randperm(len(self.generated_feat_list))[:500]
        feat2 = np.vstack([self.generated_feat_list[x] for x in random_idx])
        return np.mean(np.sum(np.absolute(feat1 - feat2), axis=-1))
