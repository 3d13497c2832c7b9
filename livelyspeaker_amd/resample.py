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
        """(timesteps, importance weights) for one batch.  The contract with the reference (resample.py:52-58) is ONE call of
        ``np.random.choice(T, size=(batch,), p=probabilities)`` per batch on numpy's global generator, so that ``np.random.seed``
        replays the reference's timestep stream; the weight of a drawn step is 1 / (T * its probability) -- all ones for the
        uniform sampler."""
        prob = np.asarray(self.weights(), dtype=np.float64)
        prob = prob / prob.sum()
        n_steps = prob.shape[0]
        drawn = np.random.choice(n_steps, size=(batch_size,), p=prob)
        importance = np.reciprocal(n_steps * prob[drawn])
        return (th.as_tensor(drawn, dtype=th.long, device=device),
                th.as_tensor(importance, dtype=th.float32, device=device))
