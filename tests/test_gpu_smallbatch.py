"""Small batches on the batch-level kernels (ls_set_path, round 3).

The fused step kernel gives each sample one workgroup = one CU, so a diffusion step costs one CU's time for eight layers (0.68 ms
TED) however small the batch: at the reference's own batch of 4 clips, 252 CUs idle.  The batch-level kernels of the long-sequence
path (ls_long.hip: every row of the batch through GEMM / token-mixing launches that fill the chip) run the same arithmetic; the
engine picks them for small batches of a 34-frame model ("auto").  That path is synthetic-only at 150 frames -- at 34 frames it is
pinned here to the REFERENCE's fixtures like the fused kernel: G2 single steps, G3 config-1 loop, G4 ddim100 / skip 80 +
init_image, G5 1000 steps, G7 identical seeds, G11 guidance scale 1 (both passes), BEAT G12 / G13."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, max_abs
from livelyspeaker_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.engine_path_auto]
TOL_FWD, TOL_LOOP = 2e-4, 3e-4


def _engine(ds, path):
    from livelyspeaker_amd import _lib
    cfg = synth.CONFIGS[ds]
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, path=path)
    eng.load_state_dict(synth.make_state_dict(cfg))
    return cfg, eng


def _g1_inputs(cfg, B=4):
    g = np.random.Generator(np.random.PCG64(1234))      # same stream as tests/golden/make_golden.py
    x = g.standard_normal((B, cfg.njoints, cfg.nfeats, cfg.nframes)).astype(np.float32)
    eps = g.standard_normal((2, B, 512)).astype(np.float32)
    noise = g.standard_normal(x.shape).astype(np.float32)
    return x, eps, noise


def _loop(eng, cfg, steps, resp, ddim, skip, use_init, dump=None, use_graph=True, B=4, scale=1.5):
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    sch = orc.Schedule(steps, resp)
    eng.set_schedule(sch)
    eng.prepare(synth.make_cond(cfg, B, scale=scale))
    n_exec = sch.num_timesteps - skip
    tape = synth.NoiseTape(cfg, B, n_exec)
    init = synth.make_init_image(cfg, B) if use_init else None
    return eng.sample(sampler=_lib.LS_SAMPLER_DDIM if ddim else _lib.LS_SAMPLER_DDPM, x_init=tape.x_init, eps_tape=tape.eps,
                      noise_tape=tape.noise, init_image=init, skip_timesteps=skip, dump_steps=dump, use_graph=use_graph)


@pytest.mark.parametrize("ds", ["ted", "beat"])
@pytest.mark.parametrize("path", ["batch", "auto"])
def test_reference_fixtures_on_the_batch_level_kernels(ds, path, golden):
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg, eng = _engine(ds, path)
    g = golden[ds]
    try:
        # single p_sample / ddim_sample steps (G2)
        x, eps, noise = _g1_inputs(cfg)
        eng.prepare(synth.make_cond(cfg, 4))
        for name, resp, steps in (("p", "", (0, 7, 999)), ("ddim", "ddim100", (0, 50, 99))):
            eng.set_schedule(orc.Schedule(1000, resp))
            for t in steps:
                s, x0 = eng.step(_lib.LS_SAMPLER_DDPM if name == "p" else _lib.LS_SAMPLER_DDIM, t, x, eps[0], eps[1], noise)
                assert max_abs(s, g[f"G2_{name}_t{t}_sample"]) < TOL_FWD and max_abs(x0, g[f"G2_{name}_t{t}_x0"]) < TOL_FWD, (name, t)
        # config 1: B = 4, 50-step DDPM, CFG 1.5 (G3), with pred_xstart dumps; hipGraph replay == plain launches
        out, dumps = _loop(eng, cfg, 50, "", False, 0, False, dump=[0, 25, 49])
        d3 = max_abs(out, g["G3_ddpm50_final"])
        if ds == "ted":
            for k, dmp in zip((0, 25, 49), dumps):
                assert max_abs(dmp, g[f"G3_ddpm50_dump_x0_step{k}"]) < TOL_LOOP, k
        assert np.array_equal(out, _loop(eng, cfg, 50, "", False, 0, False, use_graph=False))
        assert eng.timing()["single_pass"] == 0
        # the LivelySpeaker refine schedule (G4)
        d4 = max_abs(_loop(eng, cfg, 1000, "ddim100", True, 80, True), g["G4_ddim100_skip80_final"])
        print(f"{ds} [{path}]: G3 {d3:.3e}  G4 skip80 {d4:.3e}")
        assert d3 < TOL_LOOP and d4 < TOL_LOOP
        if ds == "ted":
            d5 = max_abs(_loop(eng, cfg, 1000, "", False, 0, False), g["G5_ddpm1000_final"])
            d4f = max_abs(_loop(eng, cfg, 1000, "ddim100", True, 0, False), g["G4_ddim100_full_final"])
            print(f"      G5 1000 steps {d5:.3e}  G4 full {d4f:.3e}")
            assert d5 < TOL_LOOP and d4f < TOL_LOOP
    finally:
        eng.close()


@pytest.mark.parametrize("ds", ["ted", "beat"])
def test_round2_fixtures_scale1_and_beat_loops_on_the_batch_level_kernels(ds):
    """G11: guidance scale 1 with an odd batch (the reference evaluates both passes, as this path does); G12 / G13: BEAT 1000-step
    DDPM and full ddim100."""
    g = np.load(os.path.join(GOLDEN, f"{ds}_golden_r2.npz"))
    cfg, eng = _engine(ds, "batch")
    try:
        d1 = max_abs(_loop(eng, cfg, 50, "", False, 0, False, B=5, scale=1.0), g["G11_scale1_ddpm50_B5_final"])
        d2 = max_abs(_loop(eng, cfg, 1000, "ddim100", True, 80, True, B=5, scale=1.0), g["G11_scale1_ddim100_skip80_B5_final"])
        print(f"{ds}: G11 ddpm50 {d1:.3e}, ddim100/skip80 {d2:.3e}")
        assert d1 < TOL_LOOP and d2 < TOL_LOOP and eng.timing()["single_pass"] == 0
        if ds == "beat":
            for key, args in (("G12_ddpm1000_final", (1000, "", False, 0, False)), ("G13_ddim100_full_final", (1000, "ddim100", True, 0, False))):
                if key in g:
                    d = max_abs(_loop(eng, cfg, *args), g[key])
                    print(f"beat {key}: {d:.3e}")
                    assert d < TOL_LOOP
    finally:
        eng.close()


def test_the_two_paths_agree_and_auto_switches_with_the_batch():
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg = synth.TED
    outs = {}
    for path in ("fused", "batch"):
        _, eng = _engine("ted", path)
        try:
            outs[path] = _loop(eng, cfg, 30, "", False, 0, False, B=9)
        finally:
            eng.close()
    d = max_abs(outs["fused"], outs["batch"])
    print(f"fused vs batch-level kernels, 30 steps, B = 9: {d:.3e}")
    assert 0 < d < 5e-5                                              # same arithmetic, different summation order: close, not bitwise
    # auto: a small batch takes the sample-split kernel (up to ~80 clips: 64 on two slices + the rest on eight), one that puts a pass workgroup on most CUs the one-pass-per-workgroup
    # kernel (8-wave workgroups, one per CU; 300 single-pass clips: 256 there + 44 on the sample-split kernel), a large one the fused
    # kernel; scale 1 runs single-pass wherever the kernels have that form.  (The batch-level kernels, round 3's answer to small batches,
    # no longer win anywhere at 34 frames; they stay selectable and are what other frame counts run on.)
    _, eng = _engine("ted", "auto")
    try:
        eng.set_schedule(orc.Schedule(4, ""))
        for B, scale, want_path, want_single in ((6, 1.0, 2, 1), (6, 1.5, 2, 0), (72, 1.5, 2, 0), (120, 1.5, 3, 0), (300, 1.0, 3, 1), (256, 1.5, 0, 0)):
            eng.prepare(synth.make_cond(cfg, B, scale=scale))
            out = eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=5)
            t = eng.timing()
            assert np.isfinite(out).all() and (t["step_path"], t["single_pass"]) == (want_path, want_single), (B, scale, t)
        # the opt-in split-precision mode exists in the fused kernel only: a small batch stays there
        eng.set_precision("bf16x3")
        eng.prepare(synth.make_cond(cfg, 6, scale=1.0))
        eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=5)
        assert eng.timing()["single_pass"] == 1
    finally:
        eng.close()


def test_dropin_identical_seeds_mode_on_the_default_path(golden):
    """The reference's own configuration (4 clips, 50-step DDPM, torch.manual_seed(233)) through the drop-in modules with the engine
    left at its default: the batch-level kernels -- still the reference's CPU sample (fixture G7)."""
    import torch
    from types import SimpleNamespace
    from livelyspeaker_amd.cfg_sampler import ClassifierFreeSampleModel
    from livelyspeaker_amd.model_util import create_model_and_diffusion, load_model_wo_clip
    cfg = synth.TED
    args = SimpleNamespace(mdm_condm="text", latent_dim=512, ff_size=1024, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                           emb_trans_dec=False, dataset="humanml", lang_model=None, mlpact="silu", diffusion_steps=50,
                           noise_schedule="cosine", sigma_small=True, lambda_vel=1.0, lambda_rcxyz=0.0, lambda_fc=0.0, njoints=9)
    model, diffusion = create_model_and_diffusion(args, "")
    load_model_wo_clip(model, {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()})
    model = ClassifierFreeSampleModel(model).to("cuda:0")
    model.eval()
    y = {k: torch.from_numpy(v) for k, v in synth.make_cond(cfg, 4).items()}
    torch.manual_seed(233)
    sample = diffusion.p_sample_loop(model, (4, 9, 3, 34), clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=0, init_image=None,
                                     progress=False, dump_steps=None, noise=None, const_noise=False)
    assert model.model.engine().path == "auto"
    d = max_abs(sample.cpu().numpy(), golden["ted"]["G7_seed233_ddpm50_final"])
    print(f"G7 identical seeds on the default path: {d:.3e}")
    assert d < TOL_LOOP
    # one step of the same model pinned to the fused kernel gives the same sample to rounding
    model.model.step_path = "fused"
    torch.manual_seed(233)
    fused = diffusion.p_sample_loop(model, (4, 9, 3, 34), clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=0, init_image=None,
                                    progress=False, dump_steps=None, noise=None, const_noise=False)
    assert max_abs(fused.cpu().numpy(), sample.cpu().numpy()) < 5e-5
