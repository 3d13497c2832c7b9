"""TEST / BASELINE INFRASTRUCTURE -- torch-CPU fp32 restatement of the SAMPLING path (SURVEY.md section 8d "CPU baseline timing").

The reference's Python cannot travel to the GPU box, so bench.py's ``cpu_baseline`` leg times THIS port of its algorithm on the
box's host cores (multi-threaded aten: oneDNN convolutions, MKL/OpenBLAS addmm -- the same arithmetic libraries the reference's
CPU path runs on).  Two modes, reported separately so the algorithmic saving is never mistaken for kernel speed:

  * reference-faithful: per step, per CFG pass the full ``RAG.forward`` incl. the audio encoder, exactly the op sequence of
      ClassifierFreeSampleModel.forward   scripts/model/cfg_sampler.py:24-31
      RAG.forward (eval: mask_cond)       scripts/model/RAG.py:80-133
      WavEncoder                          scripts/model/audio_enc.py:6-25
      TransMLP / MLPblock / LN_spatial    scripts/model/mlp_module.py:21-91, TimestepEmbedder :123-136
      p_sample / ddim_sample              scripts/diffusion/gaussian_diffusion.py:507-558, 745-798
    (it reuses ``oracle.train_oracle.rag_forward_train``, the functional RAG.forward pinned to the reference's training
    fixtures, with drop = 0 / 1 standing for the cond / uncond pass);
  * hoisted: the step-invariant work (audio encoder, static columns of input_mapping, speaker mu / std, timestep table) done
    once per call, as the MI355X engine does -- the fair "same algorithm on CPU" comparison.

Pinned by tests/test_oracle_golden.py against the numpy oracle and the reference-generated fixture G3.  Only tests/ and
bench.py's cpu_baseline leg import this; the product path never does.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .rag_oracle import Schedule
from .train_oracle import ln_spatial, positional_table, rag_forward_train, wav_encoder


class TorchCpuSampler:
    def __init__(self, sd: dict, njoints: int, nfeats: int, n_prefix_tokens: int = 1, n_pre_seq: int = 4, layers: int = 8):
        self.P = {k: torch.from_numpy(np.array(v, dtype=np.float32, copy=True)) for k, v in sd.items() if not k.endswith(".pe")}
        self.J, self.Fd, self.npt, self.n_pre_seq, self.layers = njoints, nfeats, n_prefix_tokens, n_pre_seq, layers
        self.JF = njoints * nfeats
        self.pe = positional_table()
        self.prep = None

    @staticmethod
    def _y(y):
        return {k: (v if torch.is_tensor(v) else torch.from_numpy(np.asarray(v))) for k, v in y.items()}

    # ---- reference-faithful: what one ClassifierFreeSampleModel.forward costs on the reference's CPU path ------------------
    def cfg_forward_faithful(self, x, t, y, eps_c, eps_u):
        B = x.shape[0]
        zero, one = torch.zeros(B), torch.ones(B)
        out_c, _, _ = rag_forward_train(self.P, x, t, y, zero, eps_c.view(B, 1, 512), self.npt, self.n_pre_seq, self.layers)
        out_u, _, _ = rag_forward_train(self.P, x, t, y, one, eps_u.view(B, 1, 512), self.npt, self.n_pre_seq, self.layers)
        return out_u + y["scale"].view(-1, 1, 1, 1) * (out_c - out_u)

    # ---- hoisted --------------------------------------------------------------------------------------------------------
    def prepare(self, y):
        P, B, T = self.P, y["audio_input"].shape[0], y["origin_x"].shape[-1]
        af = wav_encoder(P, y["audio_input"])                                           # [B, T, 256]
        ox = y["origin_x"].clone()
        ox[..., self.n_pre_seq:] = 0
        pre = ox.permute(0, 3, 1, 2).reshape(B, T, self.JF)
        bit = torch.zeros(B, T, 1)
        bit[:, :self.n_pre_seq] = 1
        Wst = P["input_mapping.weight"][:, self.JF:]                                     # static columns (RAG.py:110-114)
        static_c = F.linear(torch.cat([pre, bit, af], -1), Wst, P["input_mapping.bias"])
        static_u = F.linear(torch.cat([pre, bit, torch.zeros_like(af)], -1), Wst, P["input_mapping.bias"])
        z = P["speaker_embedding.weight"][y["vid_indices"]]
        mu = F.linear(z, P["speaker_mu.weight"], P["speaker_mu.bias"])
        std = torch.exp(0.5 * F.linear(z, P["speaker_logvar.weight"], P["speaker_logvar.bias"]))
        emo = P["emotion_embedding.weight"][y["emo"][:, 0]] if self.npt == 2 else None
        self.prep = dict(static=torch.cat([static_c, static_u], 0), mu=mu, std=std, emo=emo)
        return self.prep

    def time_embed(self, t_model):
        P = self.P
        h = F.silu(F.linear(self.pe[t_model], P["backbone.embed_timestep.time_embed.0.weight"], P["backbone.embed_timestep.time_embed.0.bias"]))
        return F.linear(h, P["backbone.embed_timestep.time_embed.2.weight"], P["backbone.embed_timestep.time_embed.2.bias"])

    def cfg_forward_hoisted(self, x, temb, y, eps_c, eps_u):
        """Both CFG passes as one [2B, S, 512] batch; temb [512] (the timestep is uniform over the batch in sampling)."""
        P, pr, B, T = self.P, self.prep, x.shape[0], x.shape[-1]
        xt = x.permute(0, 3, 1, 2).reshape(B, T, self.JF)
        h = F.linear(xt, P["input_mapping.weight"][:, :self.JF])                          # x_t columns only
        h = torch.cat([h, h], 0) + pr["static"]
        style = torch.cat([pr["mu"] + eps_c.view(B, 512) * pr["std"], pr["mu"] + eps_u.view(B, 512) * pr["std"]], 0)[:, None]
        toks = [style] + ([torch.cat([pr["emo"], pr["emo"]], 0)[:, None]] if self.npt == 2 else [])
        xs = torch.cat(toks + [h], 1)
        for l in range(self.layers):
            p = f"backbone.mlps.{l}."
            xs = xs + temb
            u = ln_spatial(xs, P[p + "block1.0.alpha"], P[p + "block1.0.beta"])
            xs = xs + F.silu(F.conv1d(u, P[p + "block1.1.weight"], P[p + "block1.1.bias"]))
            u = ln_spatial(xs, P[p + "block2.0.alpha"], P[p + "block2.0.beta"])
            xs = xs + F.silu(F.linear(u, P[p + "block2.1.weight"], P[p + "block2.1.bias"]))
        out = F.linear(xs[:, self.npt:], P["output_process.poseFinal.weight"], P["output_process.poseFinal.bias"])
        out = out.reshape(2 * B, T, self.J, self.Fd).permute(0, 2, 3, 1)
        return out[B:] + y["scale"].view(-1, 1, 1, 1) * (out[:B] - out[B:])

    # ---- loop (gaussian_diffusion.py:608-743 / 895-1014) ---------------------------------------------------------------------
    @torch.no_grad()
    def sample_loop(self, sch: Schedule, y, x_init, eps_tape, noise_tape, ddim=False, eta=0.0, skip_timesteps=0, init_image=None,
                    hoisted=True, max_steps=None):
        y = self._y(y)
        img = torch.as_tensor(x_init, dtype=torch.float32).clone()
        eps_tape, noise_tape = torch.as_tensor(eps_tape), torch.as_tensor(noise_tape)
        B = img.shape[0]
        indices = list(range(sch.num_timesteps - skip_timesteps))[::-1]
        if skip_timesteps and init_image is None:
            init_image = torch.zeros_like(img)
        if init_image is not None:
            i0 = indices[0]
            img = float(sch.f32("sqrt_alphas_cumprod", i0)) * torch.as_tensor(init_image, dtype=torch.float32) + \
                float(sch.f32("sqrt_one_minus_alphas_cumprod", i0)) * img
        if hoisted:
            self.prepare(y)
            temb_all = self.time_embed(torch.as_tensor(np.asarray(sch.timestep_map), dtype=torch.long))
        f = lambda name, i: float(sch.f32(name, i))
        for k, i in enumerate(indices):
            if max_steps is not None and k >= max_steps:
                break
            if hoisted:
                x0 = self.cfg_forward_hoisted(img, temb_all[i], y, eps_tape[k][0], eps_tape[k][1])
            else:
                t_model = torch.full((B,), int(sch.timestep_map[i]), dtype=torch.long)
                x0 = self.cfg_forward_faithful(img, t_model, y, eps_tape[k][0], eps_tape[k][1])
            nz = noise_tape[k]
            if not ddim:                                                           # p_sample (:507-558), FIXED_SMALL
                mean = f("posterior_mean_coef1", i) * x0 + f("posterior_mean_coef2", i) * img
                img = mean + (float(np.exp(np.float32(0.5) * sch.f32("posterior_log_variance_clipped", i))) * nz if i != 0 else 0)
            else:                                                                  # ddim_sample (:745-798), fp32 sqrt after the cast
                ab, abp = sch.f32("alphas_cumprod", i), sch.f32("alphas_cumprod_prev", i)
                e = (f("sqrt_recip_alphas_cumprod", i) * img - x0) / f("sqrt_recipm1_alphas_cumprod", i)
                sigma = np.float32(eta) * np.sqrt((np.float32(1) - abp) / (np.float32(1) - ab)) * np.sqrt(np.float32(1) - ab / abp)
                img = x0 * float(np.sqrt(abp)) + float(np.sqrt(np.float32(1) - abp - sigma * sigma)) * e
                if i != 0:
                    img = img + float(sigma) * nz
        return img.numpy()
