"""Deterministic synthetic weights / conditioning / noise for the RAG denoising path.

Everything is drawn from ``numpy.random.Generator(PCG64(seed))`` so that this
container (where golden fixtures are generated from the imported reference) and
the GPU box (where the HIP path is checked) regenerate bit-identical inputs; only
the *outputs* are committed as fixtures.

Shapes / key names follow the reference's state-dict contract:
``scripts/model/RAG.py:56-77`` (TED), ``scripts_beat/model/RAG.py:56-77`` (BEAT),
``scripts/model/mlp_module.py:37-100``, ``scripts/model/audio_enc.py:9-20``.

Weights are re-drawn with non-degenerate scale on purpose: the reference's own
init (``mlp_module.py:63-65`` xavier gain 1e-8, ``RAG.py:67`` constant 1e-6) makes
92 % of the FLOPs numerically invisible and would hide kernel bugs.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

SEED_WEIGHTS = 7
SEED_COND = 3
SEED_NOISE = 99


@dataclass(frozen=True)
class PathConfig:
    """Static shape of one dataset variant of the path."""
    name: str
    njoints: int
    nfeats: int
    nframes: int = 34
    n_prefix_tokens: int = 1      # style (TED) / style+emotion (BEAT)
    n_pre_seq: int = 4            # RAG.py:70
    audio_len: int = 36267
    latent_dim: int = 512
    layers: int = 8
    n_speakers: int = 1400        # RAG.py:65
    n_emotions: int = 0
    audio_feat: int = 256

    @property
    def jf(self) -> int:
        return self.njoints * self.nfeats

    @property
    def seq_len(self) -> int:
        return self.nframes + self.n_prefix_tokens

    @property
    def in_feats(self) -> int:     # input_mapping fan-in, RAG.py:62
        return 2 * self.jf + 1 + self.audio_feat


TED = PathConfig("ted", njoints=9, nfeats=3, n_prefix_tokens=1, audio_len=36267)
BEAT = PathConfig("beat", njoints=47, nfeats=6, n_prefix_tokens=2, audio_len=36266,
                  n_emotions=8)
# SYNTHETIC long-sequence variant (BASELINE configs[4]'s "150 frames" wording, SURVEY.md 8d "Config 5"): the reference cannot run it
# (its token-mixing conv fixes 34 frames, its audio encoder yields 149 frames for 160 000 samples); 160 745 samples is the clip length
# for which the conv stack yields exactly 150 frames (-> 32787 -> 5463 -> 909 -> 150).  Perf-only; checked against this repo's oracle.
BEAT150 = PathConfig("beat150", njoints=47, nfeats=6, nframes=150, n_prefix_tokens=2, audio_len=160745, n_emotions=8)
CONFIGS = {"ted": TED, "beat": BEAT, "beat150": BEAT150}

AUDIO_CONV = [(1, 32, 5, 1600), (32, 64, 6, 0), (64, 128, 6, 0), (128, 256, 6, 0)]  # (cin,cout,stride,pad), k=15
AUDIO_KERNEL = 15


def audio_lengths(n: int):
    """Conv output lengths, audio_enc.py:9-20 (36267 -> 7891 -> 1313 -> 217 -> 34)."""
    out = []
    for (_, _, s, p) in AUDIO_CONV:
        n = (n + 2 * p - AUDIO_KERNEL) // s + 1
        out.append(n)
    return out


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def make_state_dict(cfg: PathConfig, seed: int = SEED_WEIGHTS) -> dict:
    """Non-degenerate weights under the reference's state-dict key names."""
    g = _rng(seed)
    D, S, L = cfg.latent_dim, cfg.seq_len, cfg.layers
    sd = {}

    def uni(shape, fan_in):
        b = 1.0 / np.sqrt(fan_in)
        return _f32(g.uniform(-b, b, size=shape))

    for i in range(L):
        p = f"backbone.mlps.{i}."
        sd[p + "block1.0.alpha"] = _f32(1.0 + 0.1 * g.standard_normal((1, 1, D)))
        sd[p + "block1.0.beta"] = _f32(0.05 * g.standard_normal((1, 1, D)))
        sd[p + "block1.1.weight"] = _f32(g.standard_normal((S, S, 1)) * (0.6 / np.sqrt(S)))
        sd[p + "block1.1.bias"] = _f32(0.05 * g.standard_normal((S,)))
        sd[p + "block2.0.alpha"] = _f32(1.0 + 0.1 * g.standard_normal((1, 1, D)))
        sd[p + "block2.0.beta"] = _f32(0.05 * g.standard_normal((1, 1, D)))
        sd[p + "block2.1.weight"] = _f32(g.standard_normal((D, D)) * (0.6 / np.sqrt(D)))
        sd[p + "block2.1.bias"] = _f32(0.05 * g.standard_normal((D,)))
    for j in (0, 2):
        sd[f"backbone.embed_timestep.time_embed.{j}.weight"] = uni((D, D), D)
        sd[f"backbone.embed_timestep.time_embed.{j}.bias"] = uni((D,), D)
    sd["input_mapping.weight"] = uni((D, cfg.in_feats), cfg.in_feats)
    sd["input_mapping.bias"] = uni((D,), cfg.in_feats)
    sd["speaker_embedding.weight"] = _f32(0.5 * g.standard_normal((cfg.n_speakers, 256)))
    sd["speaker_mu.weight"] = uni((D, 256), 256)
    sd["speaker_mu.bias"] = uni((D,), 256)
    sd["speaker_logvar.weight"] = _f32(0.3 * g.uniform(-1, 1, size=(D, 256)) / 16.0)
    sd["speaker_logvar.bias"] = _f32(-2.0 + 0.1 * g.standard_normal((D,)))
    for idx, (cin, cout, _, _) in zip((0, 3, 6, 9), AUDIO_CONV):
        fan = cin * AUDIO_KERNEL
        sd[f"audio_encoder.feat_extractor.{idx}.weight"] = uni((cout, cin, AUDIO_KERNEL), fan)
        sd[f"audio_encoder.feat_extractor.{idx}.bias"] = uni((cout,), fan)
    sd["output_process.poseFinal.weight"] = uni((cfg.jf, D), D)
    sd["output_process.poseFinal.bias"] = uni((cfg.jf,), D)
    if cfg.n_emotions:
        sd["emotion_embedding.weight"] = _f32(0.3 * g.standard_normal((cfg.n_emotions, D)))
    return sd


def make_cond(cfg: PathConfig, batch: int, seed: int = SEED_COND, scale: float = 1.5,
              first_sample: int = 0, total: int | None = None) -> dict:
    """Synthetic conditioning. Drawn for ``total`` samples and sliced
    ``[first_sample, first_sample+batch)`` so shards of a larger job see the same data."""
    total = batch if total is None else total
    g = _rng(seed)
    audio = _f32(0.1 * g.standard_normal((total, cfg.audio_len)))
    origin_x = _f32(0.3 * g.standard_normal((total, cfg.njoints, cfg.nfeats, cfg.nframes)))
    n_vid = 1370 if cfg.name == "ted" else 30          # speakers: TED 1370, BEAT 30
    vid = g.integers(0, n_vid, size=(total,)).astype(np.int64)
    sl = slice(first_sample, first_sample + batch)
    y = {"audio_input": audio[sl].copy(), "origin_x": origin_x[sl].copy(),
         "vid_indices": vid[sl].copy(),
         "scale": np.full((batch,), scale, dtype=np.float32)}
    if cfg.n_emotions:
        emo = g.integers(0, cfg.n_emotions, size=(total, 1)).astype(np.int64)
        y["emo"] = np.repeat(emo, cfg.nframes, axis=1)[sl].copy()
    return y


def make_init_image(cfg: PathConfig, batch: int, seed: int = SEED_COND + 1000) -> np.ndarray:
    """Stand-in for the SAG decoder output (config 3 until the SAG row is built)."""
    g = _rng(seed)
    return _f32(0.3 * g.standard_normal((batch, cfg.njoints, cfg.nfeats, cfg.nframes)))


def make_sag_state_dict(cfg: PathConfig = None, seed: int = SEED_WEIGHTS + 100, latent: int = 512, ff: int = 1024,
                        layers: int = 3) -> dict:
    """SAG decoder weights under Decoder_TRANSFORMER's state-dict keys (scripts/model/motionclip_module.py:98-136;
    in the SAG.pth checkpoint they carry a 'decoder.' prefix)."""
    cfg = cfg or TED
    g = _rng(seed)
    D = latent
    sd = {}

    def uni(shape, fan_in):
        b = 1.0 / np.sqrt(fan_in)
        return _f32(g.uniform(-b, b, size=shape))

    for i in range(layers):
        p = f"seqTransDecoder.layers.{i}."
        for att in ("self_attn", "multihead_attn"):
            sd[p + att + ".in_proj_weight"] = _f32(g.standard_normal((3 * D, D)) / np.sqrt(D))
            sd[p + att + ".in_proj_bias"] = _f32(0.05 * g.standard_normal((3 * D,)))
            sd[p + att + ".out_proj.weight"] = uni((D, D), D)
            sd[p + att + ".out_proj.bias"] = _f32(0.05 * g.standard_normal((D,)))
        sd[p + "linear1.weight"] = uni((ff, D), D)
        sd[p + "linear1.bias"] = uni((ff,), D)
        sd[p + "linear2.weight"] = uni((D, ff), ff)
        sd[p + "linear2.bias"] = uni((D,), ff)
        for n in ("norm1", "norm2", "norm3"):
            sd[p + n + ".weight"] = _f32(1.0 + 0.1 * g.standard_normal((D,)))
            sd[p + n + ".bias"] = _f32(0.05 * g.standard_normal((D,)))
    sd["finallayer.weight"] = uni((cfg.jf, D), D)
    sd["finallayer.bias"] = uni((cfg.jf,), D)
    sd["mapping.weight"] = uni((D, cfg.jf + 1), cfg.jf + 1)
    sd["mapping.bias"] = uni((D,), cfg.jf + 1)
    return sd


def make_text_features(batch: int, seed: int = SEED_COND + 2000, latent: int = 512) -> np.ndarray:
    """Stand-in for clip_model.encode_text(...) (CLIP is an absent third-party package): z ~ 0.3 N(0,1) [B,512]."""
    return _f32(0.3 * _rng(seed).standard_normal((batch, latent)))


def make_embedding_net_state_dict(pose_dim: int = 27, base: int = 32, seed: int = SEED_WEIGHTS + 200, hidden=(8, 4)) -> dict:
    """Pose-encoder part of the FGD auto-encoder under EmbeddingNet's state-dict keys (scripts/model/embedding_net.py:41-66,
    checkpoint entry 'gen_dict'; hidden = out_net widths in units of base: (8, 4) TED 384-256-128-32, (4, 2) BEAT
    motion_autoencoder.py:48-56); BatchNorm running statistics are drawn non-trivial so that eval-mode folding is tested."""
    g = _rng(seed)
    sd = {}

    def uni(shape, fan_in):
        b = 2.5 / np.sqrt(fan_in)          # gain > 1 so that features keep O(1) spread through the 8 layers
        return _f32(g.uniform(-b, b, size=shape))

    def bn(p, n):
        sd[p + "weight"] = _f32(1.0 + 0.2 * g.standard_normal(n))
        sd[p + "bias"] = _f32(0.1 * g.standard_normal(n))
        sd[p + "running_mean"] = _f32(0.2 * g.standard_normal(n))
        sd[p + "running_var"] = _f32(g.uniform(0.5, 1.5, size=n))

    for i, (cin, cout, k) in enumerate(((pose_dim, base, 3), (base, 2 * base, 3), (2 * base, 2 * base, 4))):
        sd[f"pose_encoder.net.{i}.0.weight"] = uni((cout, cin, k), cin * k)
        sd[f"pose_encoder.net.{i}.0.bias"] = uni((cout,), cin * k)
        bn(f"pose_encoder.net.{i}.1.", cout)
    sd["pose_encoder.net.3.weight"] = uni((base, 2 * base, 3), 2 * base * 3)
    sd["pose_encoder.net.3.bias"] = uni((base,), 2 * base * 3)
    h1, h2 = hidden[0] * base, hidden[1] * base
    for lin, (fin, fout) in ((0, (12 * base, h1)), (3, (h1, h2)), (6, (h2, base))):
        sd[f"pose_encoder.out_net.{lin}.weight"] = uni((fout, fin), fin)
        sd[f"pose_encoder.out_net.{lin}.bias"] = uni((fout,), fin)
        if lin != 6:
            bn(f"pose_encoder.out_net.{lin + 1}.", fout)
    for n in ("fc_mu", "fc_logvar"):
        sd[f"pose_encoder.{n}.weight"] = uni((base, base), base)
        sd[f"pose_encoder.{n}.bias"] = uni((base,), base)
    return sd


def make_pose_sets(n: int, pose_dim: int = 27, nframes: int = 34, seed: int = SEED_COND + 3000):
    """(generated, real) pose clips [n, nframes, pose_dim] for the evaluator: real ~ smooth random walks, generated = real +
    perturbation, so the Frechet distance is finite and non-trivial."""
    g = _rng(seed)
    real = _f32(np.cumsum(0.15 * g.standard_normal((n, nframes, pose_dim)), axis=1))
    gen = _f32(0.7 * real + 0.3 * g.standard_normal((n, nframes, pose_dim)) + 0.2)
    return gen, real


def make_train_batch(cfg: PathConfig, batch: int, step: int = 0, first_sample: int = 0, total: int | None = None):
    """Deterministic training batch + the step's random draws, in the reference's draw order
    (x_start/cond from the data loader, then noise = randn_like(x_start) gaussian_diffusion.py:1281, the mask_cond
    bernoulli RAG.py:88, eps = randn_like(z_mu) RAG.py:12).  Returns (x_start, y, noise, drop, eps)."""
    total = batch if total is None else total
    y = make_cond(cfg, total, seed=SEED_COND + 10 * step)
    g = _rng(4242 + step)
    x_start = _f32(0.4 * g.standard_normal((total, cfg.njoints, cfg.nfeats, cfg.nframes)))
    y["origin_x"] = x_start.copy()                      # train_loop.py:131: 'origin_x': motion.clone()
    noise = _f32(g.standard_normal(x_start.shape))
    drop = (g.uniform(size=(total,)) < 0.34).astype(np.float32)
    drop[0], drop[1 % total] = 1.0, 0.0                 # both mask_cond branches present in every batch
    eps = _f32(g.standard_normal((total, cfg.latent_dim)))
    sl = slice(first_sample, first_sample + batch)
    y = {k: v[sl].copy() for k, v in y.items()}
    return x_start[sl].copy(), y, noise[sl].copy(), drop[sl].copy(), eps[sl].copy()


class NoiseTape:
    """Pre-drawn N(0,1) tape consumed in the reference's draw order:
    ``randn(B,J,F,T)`` once, then per step ``randn_like(B,1,512)`` (cond pass),
    ``randn_like(B,1,512)`` (uncond pass), ``randn_like(B,J,F,T)`` (step noise).
    RAG.py:120, cfg_sampler.py:29-30, gaussian_diffusion.py:543/787, :704/:975."""

    def __init__(self, cfg: PathConfig, batch: int, n_steps: int, seed: int = SEED_NOISE):
        g = _rng(seed)
        shp = (batch, cfg.njoints, cfg.nfeats, cfg.nframes)
        self.x_init = _f32(g.standard_normal(shp))
        self.eps = _f32(g.standard_normal((n_steps, 2, batch, cfg.latent_dim)))
        self.noise = _f32(g.standard_normal((n_steps,) + shp))
        self.n_steps = n_steps
        self.shape = shp

    def draws(self):
        """Flat list of arrays in consumption order."""
        out = [self.x_init]
        for i in range(self.n_steps):
            out.append(self.eps[i, 0][:, None, :])
            out.append(self.eps[i, 1][:, None, :])
            out.append(self.noise[i])
        return out


def make_inpainting(cfg: PathConfig, batch: int, n_steps: int, seed: int = SEED_COND + 3000):
    """Inputs of p_mean_variance's inpainting branch (gaussian_diffusion.py:314-320): y['inpainting_mask'] (bool: True = take the given
    motion), y['inpainted_motion'], and the randn_like(inpainted_motion) draws of its q_sample, one per step with t > 0.  The mask
    keeps the first ten frames of every joint and, beyond them, every third joint."""
    g = _rng(seed)
    shp = (batch, cfg.njoints, cfg.nfeats, cfg.nframes)
    mask = np.zeros(shp, dtype=bool)
    mask[..., :10] = True
    mask[:, ::3, :, :] = True
    motion = _f32(g.standard_normal(shp) * 0.3)
    noise = _f32(g.standard_normal((n_steps,) + shp))
    return mask, motion, noise


def schedule(diffusion_steps: int, timestep_respacing: str = ""):
    """The product's own schedule object (SpacedDiffusion tables) for tools and examples that drive the engine directly."""
    from types import SimpleNamespace
    from .model_util import create_gaussian_diffusion
    return create_gaussian_diffusion(SimpleNamespace(diffusion_steps=diffusion_steps, noise_schedule="cosine", sigma_small=True,
                                                     lambda_vel=1.0, lambda_rcxyz=0.0, lambda_fc=0.0), timestep_respacing)
