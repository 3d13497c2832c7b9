"""CPU oracle for the RAG denoising hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A plain numpy (fp32) restatement of the reference's algorithm for the path named in
BASELINE.json. Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s checker legs (``cpu_baseline``, ``parity_in_run``) may import this module; the product package
``livelyspeaker_amd`` never does (it fails loudly if the HIP library is missing).

Parity pin: the reference has no golden vectors of its own.  This oracle
is pinned against outputs of the *reference itself*, imported in the build container by
``tests/golden/make_golden.py`` (weights/noise from ``livelyspeaker_amd.synth``), and
committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` re-checks it on
every CPU run.

Each function cites the reference file:line it follows (paths relative to
/root/reference).  Two evaluation modes are offered for the denoiser:
  * ``hoisted=True``  - step-invariant work (audio encoder, static part of input_mapping,
    speaker mu/logvar, timestep-embedding table) computed once per sampling call; this is
    the algebraic form the HIP path uses (validated equal to the reference to ~4e-6).
  * ``hoisted=False`` - "reference-faithful": audio encoder + full input_mapping re-run in
    both forwards of every step exactly as scripts/model/RAG.py:106-114 does.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32
AUDIO_CONV = [(5, 1600), (6, 0), (6, 0), (6, 0)]      # (stride, padding); kernel 15  (audio_enc.py:9-20)
AUDIO_KEYS = (0, 3, 6, 9)


# --------------------------------------------------------------------------- schedule
def cosine_betas(T: int, max_beta: float = 0.999) -> np.ndarray:
    """scripts/diffusion/gaussian_diffusion.py:26-70 (get_named_beta_schedule 'cosine')."""
    ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return np.array([min(1 - ab((i + 1) / T) / ab(i / T), max_beta) for i in range(T)],
                    dtype=np.float64)


def space_timesteps(num_timesteps: int, section_counts) -> list:
    """scripts/diffusion/respace.py:9-62."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == want:
                    return sorted(set(range(0, num_timesteps, i)))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start, steps = 0, []
    for i, cnt in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError(f"cannot divide section of {size} steps into {cnt}")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return sorted(set(steps))


class Schedule:
    """Tables of GaussianDiffusion.__init__ (gaussian_diffusion.py:168-204) after
    SpacedDiffusion's beta re-derivation (respace.py:74-88). All float64."""

    def __init__(self, diffusion_steps: int = 1000, timestep_respacing="", noise_schedule="cosine"):
        assert noise_schedule == "cosine"
        base = cosine_betas(diffusion_steps)
        use = space_timesteps(diffusion_steps, timestep_respacing or [diffusion_steps])
        base_ac = np.cumprod(1.0 - base, axis=0)
        last, nb, tmap = 1.0, [], []
        for i, ac in enumerate(base_ac):
            if i in set(use):
                nb.append(1 - ac / last)
                last = ac
                tmap.append(i)
        betas = np.array(nb, dtype=np.float64)
        self.timestep_map = np.array(tmap, dtype=np.int64)
        self.betas = betas
        self.num_timesteps = len(betas)
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(
            np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = ((1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas)
                                     / (1.0 - self.alphas_cumprod))

    TABLES = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next",
              "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
              "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
              "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")

    def f32(self, name: str, i: int) -> np.float32:
        """_extract_into_tensor (gaussian_diffusion.py:1651-1664): fp64 table entry cast to fp32."""
        return F32(getattr(self, name)[i])


# --------------------------------------------------------------------------- small ops
def silu(x):
    return (x / (F32(1) + np.exp(-x))).astype(F32)


def ln_spatial(x, alpha, beta, eps=1e-5):
    """scripts/model/mlp_module.py:21-35 (LN_spatial)."""
    mean = x.mean(axis=-1, keepdims=True, dtype=F32)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True, dtype=F32)
    std = np.sqrt(var + F32(eps))
    return ((x - mean) / std * alpha + beta).astype(F32)


def positional_row(t: np.ndarray, d_model: int = 512) -> np.ndarray:
    """pe[t] of PositionalEncoding (mlp_module.py:104-116); fp32 like the torch buffer."""
    pos = t.astype(F32)[:, None]
    div = np.exp(np.arange(0, d_model, 2, dtype=F32) * F32(-math.log(10000.0) / d_model)).astype(F32)
    pe = np.zeros((len(t), d_model), dtype=F32)
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return pe


def conv1d(x, w, b, stride, pad, chunk=8):
    """nn.Conv1d forward: x[B,Cin,L], w[Cout,Cin,K] -> [B,Cout,Lout]."""
    B, Cin, L = x.shape
    Cout, _, K = w.shape
    if pad:
        x = np.pad(x, ((0, 0), (0, 0), (pad, pad)))
    Lout = (x.shape[2] - K) // stride + 1
    wm = w.reshape(Cout, Cin * K).T.copy()
    out = np.empty((B, Cout, Lout), dtype=F32)
    for s in range(0, B, chunk):
        win = np.lib.stride_tricks.sliding_window_view(x[s:s + chunk], K, axis=2)[:, :, ::stride]  # [b,Cin,Lout,K]
        cols = np.ascontiguousarray(win.transpose(0, 2, 1, 3)).reshape(-1, Cin * K)
        y = cols @ wm + b
        out[s:s + chunk] = y.reshape(-1, Lout, Cout).transpose(0, 2, 1)
    return out


def instance_norm_lrelu(x, eps=1e-5, slope=0.3):
    """nn.InstanceNorm1d(affine=False) + LeakyReLU(0.3) (audio_enc.py:10-11)."""
    mean = x.mean(axis=2, keepdims=True, dtype=F32)
    var = ((x - mean) ** 2).mean(axis=2, keepdims=True, dtype=F32)
    y = (x - mean) / np.sqrt(var + F32(eps))
    return np.where(y >= 0, y, y * F32(slope)).astype(F32)


# --------------------------------------------------------------------------- the denoiser
class RagOracle:
    """RAG.forward + ClassifierFreeSampleModel.forward restated (RAG.py:98-133, cfg_sampler.py:24-31)."""

    def __init__(self, sd: dict, njoints: int, nfeats: int, n_prefix_tokens: int = 1,
                 nframes: int = 34, n_pre_seq: int = 4):
        self.sd = {k: np.asarray(v, dtype=F32) for k, v in sd.items()}
        self.J, self.Fe, self.T = njoints, nfeats, nframes
        self.JF = njoints * nfeats
        self.npre = n_prefix_tokens
        self.n_pre_seq = n_pre_seq
        self.S = nframes + n_prefix_tokens
        self.D = self.sd["input_mapping.weight"].shape[0]
        self.L = 1 + max(int(k.split(".")[2]) for k in self.sd if k.startswith("backbone.mlps."))
        self._prep = None

    # -- a13: WavEncoder (audio_enc.py:6-25)
    def audio_encoder(self, audio):
        x = np.asarray(audio, dtype=F32)[:, None, :]
        for li, (key, (stride, pad)) in enumerate(zip(AUDIO_KEYS, AUDIO_CONV)):
            w = self.sd[f"audio_encoder.feat_extractor.{key}.weight"]
            b = self.sd[f"audio_encoder.feat_extractor.{key}.bias"]
            x = conv1d(x, w, b, stride, pad)
            if li < 3:
                x = instance_norm_lrelu(x)
        return np.ascontiguousarray(x.transpose(0, 2, 1))           # [B,T,256]

    # -- a18: TimestepEmbedder (mlp_module.py:123-136)
    def time_embed(self, t):
        pe = positional_row(np.asarray(t))
        w0, b0 = self.sd["backbone.embed_timestep.time_embed.0.weight"], self.sd["backbone.embed_timestep.time_embed.0.bias"]
        w2, b2 = self.sd["backbone.embed_timestep.time_embed.2.weight"], self.sd["backbone.embed_timestep.time_embed.2.bias"]
        return (silu(pe @ w0.T + b0) @ w2.T + b2).astype(F32)        # [B,512]

    # -- a15: InputProcess features without the x_t columns (RAG.py:110-112, 184-192)
    def _static_feats(self, origin_x, af, uncond):
        B = origin_x.shape[0]
        ox = np.array(origin_x, dtype=F32, copy=True)
        ox[..., self.n_pre_seq:] = 0                                   # RAG.py:110 (in place there)
        ox = ox.transpose(0, 3, 1, 2).reshape(B, self.T, self.JF)
        bit = np.zeros((B, self.T, 1), dtype=F32)
        bit[:, :self.n_pre_seq] = 1
        a = np.zeros_like(af) if uncond else af                       # mask_cond, RAG.py:80-96
        return np.concatenate([ox, bit, a], axis=-1)                   # [B,T,JF+1+256]

    def prepare(self, y: dict):
        """Once-per-call work of the hoisted form."""
        sd = self.sd
        af = self.audio_encoder(y["audio_input"])
        W, b = sd["input_mapping.weight"], sd["input_mapping.bias"]
        Ws = W[:, self.JF:]
        static_c = (self._static_feats(y["origin_x"], af, False) @ Ws.T + b).astype(F32)
        static_u = (self._static_feats(y["origin_x"], af, True) @ Ws.T + b).astype(F32)
        z = sd["speaker_embedding.weight"][np.asarray(y["vid_indices"])]
        mu = (z @ sd["speaker_mu.weight"].T + sd["speaker_mu.bias"]).astype(F32)
        logvar = (z @ sd["speaker_logvar.weight"].T + sd["speaker_logvar.bias"]).astype(F32)
        emo = None
        if self.npre == 2:
            emo = sd["emotion_embedding.weight"][np.asarray(y["emo"])[:, 0]]   # scripts_beat/model/RAG.py:125
        self._prep = dict(af=af, static=(static_c, static_u), mu=mu, logvar=logvar,
                          std=np.exp(F32(0.5) * logvar).astype(F32), emo=emo)
        return self._prep

    # -- a19/a20: TransMLP / MLPblock (mlp_module.py:67-91)
    def backbone(self, xseq, temb, trace=None):
        sd = self.sd
        x = xseq
        emb = temb[:, None, :]
        for i in range(self.L):
            p = f"backbone.mlps.{i}."
            x = x + emb
            u = ln_spatial(x, sd[p + "block1.0.alpha"], sd[p + "block1.0.beta"])
            wt = sd[p + "block1.1.weight"][:, :, 0]
            u = np.einsum("st,btd->bsd", wt, u, optimize=True).astype(F32) + sd[p + "block1.1.bias"][None, :, None]
            x = x + silu(u)
            v = ln_spatial(x, sd[p + "block2.0.alpha"], sd[p + "block2.0.beta"])
            v = (v @ sd[p + "block2.1.weight"].T + sd[p + "block2.1.bias"]).astype(F32)
            x = (x + silu(v)).astype(F32)
            if trace is not None:
                trace.append(x.copy())
        return x

    def forward(self, x, t, y=None, uncond=False, eps=None, hoisted=True, trace=None):
        """RAG.forward (RAG.py:98-133). x [B,J,F,T]; t int [B]; eps [B,512] = the randn_like
        drawn by reparameterize (RAG.py:10-13). Returns output [B,J,F,T] (contiguous)."""
        sd = self.sd
        B = x.shape[0]
        xt = np.asarray(x, dtype=F32).transpose(0, 3, 1, 2).reshape(B, self.T, self.JF)
        W, b = sd["input_mapping.weight"], sd["input_mapping.bias"]
        if hoisted:
            pr = self._prep if self._prep is not None else self.prepare(y)
            tok = (xt @ W[:, :self.JF].T + pr["static"][1 if uncond else 0]).astype(F32)
            mu, std, emo = pr["mu"], pr["std"], pr["emo"]
        else:
            af = self.audio_encoder(y["audio_input"])
            feats = np.concatenate([xt, self._static_feats(y["origin_x"], af, uncond)], axis=-1)
            tok = (feats @ W.T + b).astype(F32)
            z = sd["speaker_embedding.weight"][np.asarray(y["vid_indices"])]
            mu = (z @ sd["speaker_mu.weight"].T + sd["speaker_mu.bias"]).astype(F32)
            std = np.exp(F32(0.5) * (z @ sd["speaker_logvar.weight"].T + sd["speaker_logvar.bias"])).astype(F32)
            emo = sd["emotion_embedding.weight"][np.asarray(y["emo"])[:, 0]] if self.npre == 2 else None
        style = (mu + np.asarray(eps, dtype=F32) * std).astype(F32)     # reparameterize
        pre = [style[:, None, :]] + ([emo[:, None, :]] if emo is not None else [])
        xseq = np.concatenate(pre + [tok], axis=1)
        if trace is not None:
            trace.append(xseq.copy())
        h = self.backbone(xseq, self.time_embed(t), trace)[:, self.npre:]
        out = (h @ sd["output_process.poseFinal.weight"].T + sd["output_process.poseFinal.bias"]).astype(F32)
        return np.ascontiguousarray(out.reshape(B, self.T, self.J, self.Fe).transpose(0, 2, 3, 1))

    def cfg_forward(self, x, t, y, eps_c, eps_u, hoisted=True):
        """ClassifierFreeSampleModel.forward (cfg_sampler.py:24-31)."""
        out = self.forward(x, t, y, False, eps_c, hoisted)
        out_u = self.forward(x, t, y, True, eps_u, hoisted)
        return (out_u + np.asarray(y["scale"], dtype=F32).reshape(-1, 1, 1, 1) * (out - out_u)).astype(F32)


# --------------------------------------------------------------------------- sampler
def q_sample(sch: Schedule, x_start, i: int, noise):
    """gaussian_diffusion.py:240-258."""
    return (sch.f32("sqrt_alphas_cumprod", i) * x_start
            + sch.f32("sqrt_one_minus_alphas_cumprod", i) * noise).astype(F32)


def p_sample_update(sch: Schedule, x, x0, i: int, noise):
    """q_posterior_mean_variance + p_sample (gaussian_diffusion.py:260-282, 507-558), FIXED_SMALL."""
    mean = sch.f32("posterior_mean_coef1", i) * x0 + sch.f32("posterior_mean_coef2", i) * x
    if i == 0:
        return mean.astype(F32)
    return (mean + np.exp(F32(0.5) * sch.f32("posterior_log_variance_clipped", i)) * noise).astype(F32)


def ddim_update(sch: Schedule, x, x0, i: int, noise, eta: float = 0.0):
    """ddim_sample (gaussian_diffusion.py:745-798); sqrt taken in fp32 on the cast alpha_bar."""
    eps = ((sch.f32("sqrt_recip_alphas_cumprod", i) * x - x0)
           / sch.f32("sqrt_recipm1_alphas_cumprod", i)).astype(F32)
    ab, abp = sch.f32("alphas_cumprod", i), sch.f32("alphas_cumprod_prev", i)
    sigma = F32(eta) * np.sqrt((F32(1) - abp) / (F32(1) - ab)) * np.sqrt(F32(1) - ab / abp)
    mean = x0 * np.sqrt(abp) + np.sqrt(F32(1) - abp - sigma ** 2) * eps
    if i == 0:
        return mean.astype(F32)
    return (mean + sigma * noise).astype(F32)


def sample_loop(model: RagOracle, sch: Schedule, y: dict, x_init, eps_tape, noise_tape,
                ddim=False, eta=0.0, skip_timesteps=0, init_image=None, hoisted=True,
                dump_steps=None, max_steps=None, clip_denoised=False, inpaint=None):
    """p_sample_loop / ddim_sample_loop (gaussian_diffusion.py:608-743, 895-1014) with the CFG
    wrapper inlined. eps_tape[k] = (eps_cond, eps_uncond) [2,B,512]; noise_tape[k] [B,J,F,T];
    k counts executed steps. Returns final sample (and pred_xstart dumps if requested).
    inpaint = (mask bool [B,J,F,T], motion [B,J,F,T], noise [n,B,J,F,T] or None): p_mean_variance's inpainting branch
    (gaussian_diffusion.py:314-320): model_output = model_output * ~mask + q_sample(motion, t - 1) * mask while t > 0 (the TED tree; its
    q_sample draws noise[k]), + motion * mask at t = 0 -- and always un-noised when noise is None (scripts_beat/diffusion/
    gaussian_diffusion.py:319)."""
    img = np.asarray(x_init, dtype=F32)
    if skip_timesteps and init_image is None:
        init_image = np.zeros_like(img)
    indices = list(range(sch.num_timesteps - skip_timesteps))[::-1]
    if init_image is not None:
        img = q_sample(sch, np.asarray(init_image, dtype=F32), indices[0], img)
    if hoisted:
        model.prepare(y)
    dumps = []
    B = img.shape[0]
    for k, i in enumerate(indices):
        if max_steps is not None and k >= max_steps:
            break
        t_model = np.full((B,), sch.timestep_map[i], dtype=np.int64)     # _WrappedModel, respace.py:125-130
        x0 = model.cfg_forward(img, t_model, y, eps_tape[k][0], eps_tape[k][1], hoisted)
        if inpaint is not None:
            mask, motion, inz = inpaint
            motion = np.asarray(motion, dtype=F32)
            given = q_sample(sch, motion, i - 1, inz[k]) if (inz is not None and i > 0) else motion
            x0 = np.where(mask, given, x0).astype(F32)
        if clip_denoised:                                                 # process_xstart, gaussian_diffusion.py:365-371
            x0 = np.clip(x0, -1, 1)
        if dump_steps is not None and k in dump_steps:
            dumps.append(x0.copy())
        img = (ddim_update(sch, img, x0, i, noise_tape[k], eta) if ddim
               else p_sample_update(sch, img, x0, i, noise_tape[k]))
    if dump_steps is not None:
        return img, dumps
    return img


# --------------------------------------------------------------------------- SAG decoder
def gelu_exact(x):
    from scipy.special import erf
    return (0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))).astype(F32)


def layer_norm(x, w, b, eps=1e-5):
    mean = x.mean(axis=-1, keepdims=True, dtype=F32)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True, dtype=F32)
    return ((x - mean) / np.sqrt(var + F32(eps)) * w + b).astype(F32)


class SagDecoderOracle:
    """Decoder_TRANSFORMER.forward (scripts/model/motionclip_module.py:138-183): 3 post-norm
    nn.TransformerDecoderLayer (self-attn over the 34 queries, cross-attn to a length-1 memory, GELU FFN)."""

    def __init__(self, sd: dict, njoints=9, nfeats=3, nframes=34, heads=4, n_pre_poses=4):
        self.sd = {k: np.asarray(v, dtype=F32) for k, v in sd.items()}
        self.J, self.Fe, self.T, self.H, self.npre = njoints, nfeats, nframes, heads, n_pre_poses
        self.L = 1 + max(int(k.split(".")[2]) for k in self.sd if k.startswith("seqTransDecoder.layers."))

    def _mha(self, p, q_in, kv_in):
        sd, H = self.sd, self.H
        Wi, bi = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
        D = Wi.shape[1]
        q = q_in @ Wi[:D].T + bi[:D]
        k = kv_in @ Wi[D:2 * D].T + bi[D:2 * D]
        v = kv_in @ Wi[2 * D:].T + bi[2 * D:]
        B, Tq, Tk, hd = q.shape[0], q.shape[1], k.shape[1], D // H
        qh = q.reshape(B, Tq, H, hd).transpose(0, 2, 1, 3) * F32(1.0 / math.sqrt(hd))
        kh = k.reshape(B, Tk, H, hd).transpose(0, 2, 1, 3)
        vh = v.reshape(B, Tk, H, hd).transpose(0, 2, 1, 3)
        s = qh @ kh.transpose(0, 1, 3, 2)
        s = np.exp(s - s.max(axis=-1, keepdims=True))
        s = (s / s.sum(axis=-1, keepdims=True)).astype(F32)
        o = (s @ vh).transpose(0, 2, 1, 3).reshape(B, Tq, D)
        return (o @ sd[p + "out_proj.weight"].T + sd[p + "out_proj.bias"]).astype(F32)

    def decode(self, x, z, mask=None):
        sd = self.sd
        B = x.shape[0]
        JF = self.J * self.Fe
        motion = np.asarray(x, dtype=F32).transpose(0, 3, 1, 2).reshape(B, self.T, JF).copy()
        pre = np.zeros((B, self.T, JF + 1), dtype=F32)
        pre[:, :self.npre, :JF] = motion[:, :self.npre]
        pre[:, :self.npre, JF] = 1                                       # indicating bit (:164-166)
        h = (pre @ sd["mapping.weight"].T + sd["mapping.bias"]).astype(F32)
        h = h + positional_row(np.arange(self.T))[None]                  # sequence_pos_encoder (:168), eval mode
        mem = np.asarray(z, dtype=F32)[:, None, :]
        for i in range(self.L):
            p = f"seqTransDecoder.layers.{i}."
            h = layer_norm(h + self._mha(p + "self_attn.", h, h), sd[p + "norm1.weight"], sd[p + "norm1.bias"])
            h = layer_norm(h + self._mha(p + "multihead_attn.", h, mem), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
            ff = gelu_exact(h @ sd[p + "linear1.weight"].T + sd[p + "linear1.bias"]) @ sd[p + "linear2.weight"].T + sd[p + "linear2.bias"]
            h = layer_norm(h + ff.astype(F32), sd[p + "norm3.weight"], sd[p + "norm3.bias"])
        out = (h @ sd["finallayer.weight"].T + sd["finallayer.bias"]).astype(F32)      # [B,T,JF]
        if mask is not None:
            out = out * np.asarray(mask, dtype=bool)[:, :, None]                        # "zero for padded area" (:175)
        return np.ascontiguousarray(out.reshape(B, self.T, self.J, self.Fe).transpose(0, 2, 3, 1))


# --------------------------------------------------------------------------- caller plumbing, BEAT twin
def beat_post(sample):
    """scripts_beat/test_RAG_beat.py:86 (layout) and :101 (rot6d -> matrix -> Euler XYZ in degrees) restated:
    rotation_6d_to_matrix (scripts_beat/dataloaders/rot_utils.py:529-534) and matrix_to_euler_angles(., "XYZ") (:238-257)."""
    x = np.asarray(sample, dtype=F32)
    B, J, _, T = x.shape
    dec = np.ascontiguousarray(x.transpose(0, 3, 1, 2).reshape(B, T, J * 6))
    d6 = dec.reshape(B, T, J, 6)
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = a1 / np.maximum(np.sqrt((a1 * a1).sum(-1, keepdims=True, dtype=F32)), F32(1e-12))
    b2 = a2 - (b1 * a2).sum(-1, keepdims=True, dtype=F32) * b1
    b2 = b2 / np.maximum(np.sqrt((b2 * b2).sum(-1, keepdims=True, dtype=F32)), F32(1e-12))
    b3z = b1[..., 0] * b2[..., 1] - b1[..., 1] * b2[..., 0]
    k = F32(180.0 / math.pi)
    eul = np.stack((np.arctan2(-b2[..., 2], b3z), np.arcsin(b1[..., 2]), np.arctan2(-b1[..., 1], b1[..., 0])), axis=-1).astype(F32) * k
    return {"decoded_motions": dec, "pred_euler": eul.reshape(B, T, J * 3).astype(F32)}


# --------------------------------------------------------------------------- caller plumbing
def ted_post(sample, mean_dir_vec, angle_pairs, change_angle, thres, dir_vec_pairs):
    """scripts/test_RAG_ted.py:84-111 and convert_dir_vec_to_pose (scripts/utils/data_utils.py:77-97) restated."""
    B = sample.shape[0]
    aligned = np.ascontiguousarray(np.asarray(sample, dtype=F32).transpose(0, 3, 1, 2).reshape(B, sample.shape[3], -1))
    vec = (aligned + np.asarray(mean_dir_vec, dtype=F32)).astype(F32)
    v = vec.reshape(B, aligned.shape[1], -1, 3)
    n = v / np.maximum(np.sqrt((v * v).sum(-1, keepdims=True)), F32(1e-12))
    diff = np.zeros((B, aligned.shape[1]), dtype=F32)
    for k, (pa, pb) in enumerate(angle_pairs):
        ip = np.clip((n[:, :, pa] * n[:, :, pb]).sum(-1), -1, 1)
        ang = (np.arccos(ip) / F32(math.pi)).astype(F32)
        diff[:, 1:] += np.abs(ang[:, 1:] - ang[:, :-1]) / F32(change_angle[k]) / F32(len(change_angle))
    mask = np.zeros((B, aligned.shape[1]), dtype=bool)
    for t in range(2, 33):
        c, l, r = diff[:, t], diff[:, t - 1], diff[:, t + 1]
        mask[:, t] = (c < l) & (c < r) & ((l - c >= thres) | (r - c >= thres))
    pose = np.zeros((B, aligned.shape[1], 10, 3), dtype=np.float64)
    for j, (pa, ch, ln) in enumerate(dir_vec_pairs):
        pose[:, :, ch] = pose[:, :, pa] + ln * v[:, :, j]
    return {"aligned": aligned, "angle_diff": diff, "beat_mask": mask, "pose": pose}
