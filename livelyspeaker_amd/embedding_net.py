"""``EmbeddingNet`` drop-in (``scripts/model/embedding_net.py:261-275``; BEAT ``HalfEmbeddingNet``,
``scripts_beat/model/motion_autoencoder.py:156-167``): same constructor, state-dict keys and
``net(poses, variational_encoding=False) -> (feat, mu, logvar)`` contract for the evaluator's use (eval mode,
``variational_encoding=False``), evaluated by the gfx950 engine (``ls_eval_*``).  The torch modules only hold the
parameters under the checkpoint's key names; there is no CPU path.  The decoder and ``fc_logvar`` are accepted by
``load_state_dict`` and unused -- ``logvar`` is returned as ``None`` (the evaluator discards it, ted_evaluator.py:40-41).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib


def _conv_norm_relu(cin, cout, downsample=False):
    k, s = (4, 2) if downsample else (3, 1)
    return nn.Sequential(nn.Conv1d(cin, cout, kernel_size=k, stride=s), nn.BatchNorm1d(cout), nn.LeakyReLU(0.2, True))


class PoseEncoderConv(nn.Module):
    def __init__(self, length, dim, feature_length=32, hidden=None):
        super().__init__()
        b = feature_length
        h1, h2 = hidden if hidden is not None else (8 * b, 4 * b)       # TED: 384 -> 256 -> 128 -> 32
        self.net = nn.Sequential(_conv_norm_relu(dim, b), _conv_norm_relu(b, 2 * b), _conv_norm_relu(2 * b, 2 * b, True),
                                 nn.Conv1d(2 * b, b, 3))
        self.out_net = nn.Sequential(nn.Linear(12 * b, h1), nn.BatchNorm1d(h1), nn.LeakyReLU(True), nn.Linear(h1, h2),
                                     nn.BatchNorm1d(h2), nn.LeakyReLU(True), nn.Linear(h2, b))
        self.fc_mu = nn.Linear(b, b)
        self.fc_logvar = nn.Linear(b, b)
        self.dims = (dim, length, b, h1, h2)


class EmbeddingNet(nn.Module):
    def __init__(self, pose_dim, n_frames, feature_length=32, hidden=None):
        super().__init__()
        if n_frames != 34:
            raise NotImplementedError("the pose encoder's first linear is sized for 34 frames (embedding_net.py:52-53)")
        self.pose_encoder = PoseEncoderConv(n_frames, pose_dim, feature_length, hidden)
        self.requires_grad_(False)
        self._engine = None
        self._dirty = True

    def load_state_dict(self, state_dict, strict=True, **kw):
        own = {k: v for k, v in state_dict.items() if k.startswith("pose_encoder.")}     # decoder.* is never on this path
        res = super().load_state_dict(own, strict=False, **kw)
        self._dirty = True
        return res

    def freeze_pose_nets(self):
        self.requires_grad_(False)

    def _apply(self, fn, *a, **k):
        res = super()._apply(fn, *a, **k)
        self._dirty = True
        return res

    def engine(self):
        dev = self.pose_encoder.fc_mu.weight.device
        if dev.type != "cuda" and not torch.cuda.is_available():
            raise _lib.EngineError("no MI355X visible: livelyspeaker_amd has no CPU path")
        di = dev.index if dev.type == "cuda" and dev.index is not None else torch.cuda.current_device()
        dim, length, b, h1, h2 = self.pose_encoder.dims
        if self._engine is None or self._engine.device != di:
            self._engine = _lib.EvalEngine(dim, length, b, (h1, h2), device=di)
            self._dirty = True
        if self._dirty:
            self._engine.load_state_dict({k: v.detach().cpu().numpy() for k, v in self.state_dict().items()
                                          if "num_batches_tracked" not in k})
            self._dirty = False
        return self._engine

    def forward(self, poses, variational_encoding=False):
        if variational_encoding or self.training:
            raise NotImplementedError("only the evaluator's eval-mode, variational_encoding=False path is built")
        mu = self.engine().features(poses)
        mu = mu if isinstance(mu, torch.Tensor) else torch.from_numpy(mu).to(poses.device)
        return mu, mu, None


class HalfEmbeddingNet(EmbeddingNet):
    """BEAT: ``HalfEmbeddingNet(args)`` with args.pose_length / pose_dims / vae_length; forward(poses) -> features."""

    def __init__(self, args):
        b = args.vae_length
        super().__init__(args.pose_dims, args.pose_length, feature_length=b, hidden=(4 * b, 2 * b))

    def forward(self, poses):
        return super().forward(poses, False)[0]
