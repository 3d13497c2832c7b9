"""``RAG`` drop-in: same constructor, attributes, state-dict keys and ``model(x, timesteps, y=...)``
contract as ``scripts/model/RAG.py:16-133`` (BEAT: ``scripts_beat/model/RAG.py``), evaluated by the gfx950
engine through the C-ABI.

The torch submodules below exist only to (a) hold parameters under the reference's state-dict key names
(SURVEY.md section 8b "weight contract") so ``load_state_dict`` / ``state_dict`` / ``parameters`` / ``.to`` behave as
callers expect, and (b) consume torch's RNG in the same order as the reference's ``__init__`` so that
``torch.manual_seed(s); RAG(...)`` yields the same random-init weights.  They are never called: every
forward goes to ``libls_hip.so`` and raises if it (or a GPU) is missing -- there is no CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import _lib


class _LNParams(nn.Module):                      # LN_spatial parameters (mlp_module.py:21-28)
    def __init__(self, dim):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones([1, 1, dim]))
        self.beta = nn.Parameter(torch.zeros([1, 1, dim]))


class _Block(nn.Module):                         # MLPblock parameters (mlp_module.py:37-65)
    def __init__(self, seq_len, dim):
        super().__init__()
        self.block1 = nn.Sequential(_LNParams(dim), nn.Conv1d(seq_len, seq_len, 1), nn.SiLU())
        self.block2 = nn.Sequential(_LNParams(dim), nn.Linear(dim, dim), nn.SiLU())
        nn.init.xavier_uniform_(self.block2[1].weight, gain=1e-8)
        nn.init.constant_(self.block2[1].bias, 0)


class _PE(nn.Module):                            # PositionalEncoding buffer (mlp_module.py:104-116)
    def __init__(self, d_model, max_len=5000):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0).transpose(0, 1))


class _TimeEmbed(nn.Module):                     # TimestepEmbedder (mlp_module.py:123-136)
    def __init__(self, dim, pe):
        super().__init__()
        self.sequence_pos_encoder = pe
        self.time_embed = nn.Sequential(nn.Linear(dim, dim), nn.SiLU(), nn.Linear(dim, dim))


class _Backbone(nn.Module):                      # TransMLP (mlp_module.py:76-91)
    def __init__(self, seq_len, num_layers, dim):
        super().__init__()
        self.mlps = nn.Sequential(*[_Block(seq_len, dim) for _ in range(num_layers)])
        self.sequence_pos_encoder = _PE(dim)
        self.embed_timestep = _TimeEmbed(dim, self.sequence_pos_encoder)


class _WavEncoder(nn.Module):                    # audio_enc.py:6-20
    def __init__(self):
        super().__init__()
        self.feat_extractor = nn.Sequential(
            nn.Conv1d(1, 32, 15, stride=5, padding=1600), nn.InstanceNorm1d(32), nn.LeakyReLU(0.3, inplace=True),
            nn.Conv1d(32, 64, 15, stride=6), nn.InstanceNorm1d(64), nn.LeakyReLU(0.3, inplace=True),
            nn.Conv1d(64, 128, 15, stride=6), nn.InstanceNorm1d(128), nn.LeakyReLU(0.3, inplace=True),
            nn.Conv1d(128, 256, 15, stride=6))


class _OutputProcess(nn.Module):                 # RAG.py:195-203
    def __init__(self, input_feats, latent_dim):
        super().__init__()
        self.poseFinal = nn.Linear(latent_dim, input_feats)


class RAG(nn.Module):
    #: raw audio samples that the conv stack turns into exactly 34 frames (lmdb_data_loader.py:73 / beat.py:371-375)
    AUDIO_LEN = {1: 36267, 2: 36266}

    def __init__(self, modeltype, njoints, nfeats, num_actions, translation, pose_rep, glob, glob_rot,
                 latent_dim=256, ff_size=1024, num_layers=8, num_heads=4, dropout=0.1, ablation=None,
                 activation="gelu", legacy=False, data_rep='rot6d', clip_dim=512, arch='trans_enc', mlpact='silu',
                 n_prefix_tokens=1, n_emotions=8, nframes=34, audio_len=None, **kargs):
        super().__init__()
        self.legacy, self.modeltype, self.njoints, self.nfeats = legacy, modeltype, njoints, nfeats
        self.num_actions, self.data_rep, self.pose_rep = num_actions, data_rep, pose_rep
        self.glob, self.glob_rot, self.translation = glob, glob_rot, translation
        self.cond_mode = kargs.get('cond_mode', 'no_cond')
        self.latent_dim, self.ff_size, self.num_layers, self.num_heads = latent_dim, ff_size, num_layers, num_heads
        self.dropout, self.ablation, self.activation, self.clip_dim = dropout, ablation, activation, clip_dim
        self.action_emb = kargs.get('action_emb', None)
        self.input_feats = njoints * nfeats
        self.cond_mask_prob = kargs.get('cond_mask_prob', 0.)
        self.arch, self.mlpact = arch, mlpact
        if mlpact != 'silu':
            raise NotImplementedError("the gfx950 step kernel fuses SiLU (parser default, parser_util.py:92)")
        self.n_prefix_tokens = n_prefix_tokens
        #: 34 in the reference (the token-mixing conv fixes it).  Any other value builds the SYNTHETIC long-sequence variant
        #: (perf-only; the reference cannot run it) with `audio_len` raw samples chosen so the conv stack yields `nframes` frames.
        self.nframes = int(nframes)
        self.audio_len = int(audio_len) if audio_len is not None else self.AUDIO_LEN[n_prefix_tokens]
        seq_len = self.nframes + n_prefix_tokens                 # 35 (RAG.py:56) | 36 (scripts_beat/model/RAG.py:56)

        # --- parameter holders, constructed in the reference's order (RAG.py:56-77) ---------------------
        self.backbone = _Backbone(seq_len, num_layers, latent_dim)
        self.input_mapping = nn.Linear(self.input_feats * 2 + 1 + 256, latent_dim)
        self.sequence_pos_encoder = _PE(latent_dim)
        self.speaker_embedding = nn.Embedding(1400, 256)
        nn.init.constant_(self.speaker_embedding.weight, 1e-6)
        self.speaker_mu = nn.Linear(256, latent_dim)
        self.speaker_logvar = nn.Linear(256, latent_dim)
        self.n_pre_seq = 4
        if n_prefix_tokens == 2:
            self.emotion_embedding = nn.Embedding(n_emotions, latent_dim)
            nn.init.constant_(self.emotion_embedding.weight, 1e-6)
        self.audio_encoder = _WavEncoder()
        self.output_process = _OutputProcess(self.input_feats, latent_dim)
        self.requires_grad_(False)

        #: reuse the prepared conditioning when the same ``y`` tensors are passed again (the sampling loop calls
        #: the model T times with one ``y``); set False to re-run the once-per-call stage on every call.
        self.cache_conditioning = True
        #: 'fp32' = exact fp32 MFMA (default, what parity/bench numbers refer to); 'bf16x3' = opt-in split-precision
        #: channel mixing (3 bf16 MFMAs per fp32 product, ~2^-16 relative; parity-tested against the 1e-3 contract)
        self.precision = "fp32"
        #: which kernels the diffusion steps run on: None = the engine's default ("auto": chosen per batch from a step-time model --
        #: the sample-split kernel for small batches, batch-level kernels in the middle, one workgroup per sample from ~176 clips),
        #: "fused", "batch", "coop" (ls_set_path; same arithmetic, results agree to ~1e-5)
        self.step_path = None
        self._engine = None
        self._weights_dirty = True
        self._cond_key = None
        self._prefetched_key = None
        self.n_emotions = n_emotions if n_prefix_tokens == 2 else 0

    # ------------------------------------------------------------------ nn.Module plumbing
    def parameters_wo_clip(self):
        return [p for name, p in self.named_parameters() if not name.startswith('clip_model.')]

    def load_state_dict(self, state_dict, strict=True, **kw):
        res = super().load_state_dict(state_dict, strict=strict, **kw)
        self._weights_dirty = True
        return res

    def _apply(self, fn, *a, **k):
        res = super()._apply(fn, *a, **k)
        self._weights_dirty = True
        return res

    def train(self, *args, **kwargs):            # the reference's override returns None (RAG.py:136-137)
        super().train(*args, **kwargs)

    # ------------------------------------------------------------------ engine
    def _device_index(self) -> int:
        dev = self.input_mapping.weight.device
        if dev.type == "cuda":
            return dev.index if dev.index is not None else torch.cuda.current_device()
        if not torch.cuda.is_available():
            raise _lib.EngineError("no MI355X visible: livelyspeaker_amd has no CPU path")
        return torch.cuda.current_device()

    def engine(self) -> "_lib.Engine":
        di = self._device_index()
        if self._engine is None or self._engine.device != di:
            self._engine = _lib.Engine(self.njoints, self.nfeats, self.n_prefix_tokens,
                                       self.audio_len, n_emotions=self.n_emotions,
                                       nframes=self.nframes, n_pre_seq=self.n_pre_seq, latent_dim=self.latent_dim,
                                       layers=self.num_layers, n_speakers=1400, device=di, path=self.step_path)
            self._weights_dirty = True
        if self._weights_dirty:
            sd = {k: v.detach().cpu().numpy() for k, v in self.state_dict().items() if not k.endswith(".pe")}
            self._engine.load_state_dict(sd)
            self._weights_dirty = False
            self._cond_key = self._prefetched_key = None
        if getattr(self._engine, "precision", "fp32") != self.precision:
            self._engine.set_precision(self.precision)
            self._cond_key = self._prefetched_key = None          # the kernel choice (and with it the workspaces of ls_prepare) may change with it
        want_path = self.step_path if self.step_path is not None else _lib.Engine.default_path      # None = the engine's default, both ways
        if self._engine.path != want_path and self.nframes == 34:
            self._engine.set_path(want_path)
            self._cond_key = self._prefetched_key = None
        return self._engine

    def prefetch_condition(self, y):
        """Optional hint (no counterpart in the reference): enqueue the once-per-call stage for ``y`` NOW, without waiting for it.
        The next forward / sampling call with the same ``y`` finds it resident (stream-ordered behind it), and whatever the caller
        runs in between on other streams overlaps it -- LivelySpeaker's SAG decode needs none of it
        (scripts/test_LivelySpeaker_ted.py:88-113 runs the two back to back).  RAG.py:110's in-place ``origin_x[..., 4:] = 0`` happens
        here instead of at the first forward (the SAG decoder reads the four prefix poses only)."""
        self._engine_prepared(y, wait=False)

    def _engine_prepared(self, y, wait=True):
        """Run the once-per-call stage (audio encoder, static projection, speaker style) unless ``y`` is the
        conditioning that is already resident.  Reproduces RAG.py:110's in-place ``origin_x[..., 4:] = 0``."""
        eng = self.engine()
        # RAG.py:110 zeroes origin_x[..., 4:] in place on EVERY forward.  Written through `.data` so that origin_x._version (part
        # of the cache key below) does not move with it -- and unconditionally: testing "is anything non-zero" first was a
        # device -> host round trip on every forward of a step-by-step caller
        y['origin_x'].data[..., self.n_pre_seq:].zero_()
        names = ('audio_input', 'origin_x', 'vid_indices', 'scale') + (('emo',) if self.n_prefix_tokens == 2 else ())
        key = tuple((n, y[n].data_ptr(), tuple(y[n].shape), y[n]._version) for n in names)
        if not wait:                                      # prefetch_condition: enqueue only; the next request for this key is served by it
            eng.prepare({n: y[n] for n in names}, wait=False)
            self._cond_key = self._prefetched_key = key
        elif key == getattr(self, "_prefetched_key", None):
            self._prefetched_key = None                   # one shot, also with cache_conditioning off (the reference re-runs it per call)
        elif key != self._cond_key or not self.cache_conditioning:
            eng.prepare({n: y[n] for n in names})
            self._cond_key = key
            self._prefetched_key = None                   # whatever was prefetched is no longer what the engine holds
        return eng

    def _forward_engine(self, x, timesteps, y, want):
        eng = self._engine_prepared(y)
        B = x.shape[0]
        uncond = bool(y.get('uncond', False))
        if want == "cfg":
            eps_c = torch.randn(B, 1, self.latent_dim)
            eps_u = torch.randn(B, 1, self.latent_dim)
        else:                                   # one pass only: a single reparameterize draw (RAG.py:120)
            eps_c = eps_u = torch.randn(B, 1, self.latent_dim)
        oc, ou, og = eng.forward(x, timesteps, eps_c, eps_u)
        pick = og if want == "cfg" else (ou if uncond else oc)
        return (pick if isinstance(pick, torch.Tensor) else torch.from_numpy(pick)).to(x.device)

    def forward(self, x, timesteps, y=None):
        """x: [B, njoints, nfeats, nframes] (x_t); timesteps: [B] int; y: conditioning dict (RAG.py:98-133)."""
        out = self._forward_engine(x, timesteps, y, want="single")
        eng = self._engine
        dev = x.device
        return {'output': out,
                'z_mu': torch.from_numpy(eng.read("z_mu")).to(dev)[:, None],
                'z_logvar': torch.from_numpy(eng.read("z_logvar")).to(dev)[:, None]}
