"""Timestep samplers for training, same names and draw order as ``scripts/diffusion/resample.py:8-70``.

Only the uniform sampler is built: it is the one ``TrainLoop`` hard-codes (``train_loop.py:75``); the loss-aware
samplers of the reference are never instantiated and raise here.
"""
from __future__ import annotations

import numpy as np
import torch as th


def create_named_schedule_sampler(name, diffusion):
    if name == "uniform":
        return UniformSampler(diffusion)
    raise NotImplementedError(f"unknown / unbuilt schedule sampler: {name}")


class UniformSampler:
    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights

    def sample(self, batch_size, device):
        """(timesteps, weights) -- ``np.random.choice`` exactly as resample.py:52-58, so ``np.random.seed`` reproduces
        the reference's timestep stream."""
        w = self.weights()
        p = w / np.sum(w)
        indices_np = np.random.choice(len(p), size=(batch_size,), p=p)
        indices = th.from_numpy(indices_np).long().to(device)
        weights_np = 1 / (len(p) * p[indices_np])
        weights = th.from_numpy(weights_np).float().to(device)
        return indices, weights
