"""``Decoder_TRANSFORMER`` drop-in (SAG decoder): same constructor, state-dict keys and
``forward(batch)`` contract as ``scripts/model/motionclip_module.py:98-183``; evaluated by the gfx950 engine
(``ls_sag_decode``).  The torch submodules only hold parameters under the reference's key names and consume the RNG
in the reference's constructor order; there is no CPU execution path.

The CLIP text encoder that produces ``batch['z']`` is a third-party package (openai-clip @ a9b1bf5,
requirements.txt:10) that is absent from this image: callers pass the text feature in ``batch['z']`` as the
reference's decoder expects."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .rag import _PE


class Decoder_TRANSFORMER(nn.Module):
    def __init__(self, modeltype='', njoints=9, nfeats=3, num_frames=34, latent_dim=512, ff_size=1024, num_layers=3,
                 num_heads=4, dropout=0.1, activation="gelu", ablation=None, n_pre_poses=4, use_style=False, **kargs):
        super().__init__()
        self.modeltype, self.njoints, self.nfeats, self.num_frames = modeltype, njoints, nfeats, num_frames
        self.latent_dim, self.ff_size, self.num_layers, self.num_heads = latent_dim, ff_size, num_layers, num_heads
        self.dropout, self.ablation, self.activation = dropout, ablation, activation
        if activation != "gelu":
            raise NotImplementedError("the SAG decoder kernels implement the reference's activation='gelu'")
        self.input_feats = njoints * nfeats
        self.sequence_pos_encoder = _PE(latent_dim)
        layer = nn.TransformerDecoderLayer(d_model=latent_dim, nhead=num_heads, dim_feedforward=ff_size,
                                           dropout=dropout, activation=activation)
        self.seqTransDecoder = nn.TransformerDecoder(layer, num_layers=num_layers)
        self.finallayer = nn.Linear(latent_dim, self.input_feats)
        self.mapping = nn.Linear(self.input_feats + 1, 512)          # nn.Linear(28, 512), motionclip_module.py:133
        self.n_pre_poses = n_pre_poses
        self.requires_grad_(False)
        self._engine = None
        self._weights_dirty = True

    def load_state_dict(self, state_dict, strict=True, **kw):
        res = super().load_state_dict(state_dict, strict=strict, **kw)
        self._weights_dirty = True
        return res

    def _apply(self, fn, *a, **k):
        res = super()._apply(fn, *a, **k)
        self._weights_dirty = True
        return res

    def engine(self) -> "_lib.SagEngine":
        dev = self.finallayer.weight.device
        if dev.type == "cuda":
            di = dev.index if dev.index is not None else torch.cuda.current_device()
        elif torch.cuda.is_available():
            di = torch.cuda.current_device()
        else:
            raise _lib.EngineError("no MI355X visible: livelyspeaker_amd has no CPU path")
        if self._engine is None or self._engine.device != di:
            self._engine = _lib.SagEngine(self.njoints, self.nfeats, self.num_frames, self.latent_dim, self.ff_size,
                                          self.num_layers, self.num_heads, self.n_pre_poses, device=di)
            self._weights_dirty = True
        if self._weights_dirty:
            self._engine.load_state_dict({k: v.detach().cpu().numpy() for k, v in self.state_dict().items()
                                          if not k.endswith(".pe")})
            self._weights_dirty = False
        return self._engine

    def forward(self, batch, use_text_emb=False, wait=True):
        """``wait=False`` (no counterpart in the reference): enqueue the decode on the decoder's own stream and return; the output is
        ordered for a consumer by ``_lib.stream_order(device, decoder.engine()._stream, consumer_stream)`` (examples/livelyspeaker_ted.py)."""
        z, mask = batch["z"], batch["mask"]
        if use_text_emb:
            z = batch["clip_text_emb"]
        batch['final_z'] = z.clone()
        out = self.engine().decode(batch["x"], z.float(), mask, wait=wait)
        out = (out if isinstance(out, torch.Tensor) else torch.from_numpy(out)).to(batch["x"].device)
        batch["txt_output" if use_text_emb else "output"] = out
        return batch
