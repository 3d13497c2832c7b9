"""SYNTHETIC long-sequence variant (BASELINE configs[4] as worded: BEAT, 150 frames; SURVEY.md 8d "Config 5").  The reference cannot
run this shape (its token-mixing conv fixes 34 frames), so there is NO parity claim against it: the HIP long-sequence path
(csrc/ls_long.hip, batch-level kernels on the fp32 MFMA GEMM) is checked against this repository's CPU oracle, which is generic in
the frame count and pinned to the reference at 34 frames."""
import numpy as np
import pytest

from conftest import max_abs
from livelyspeaker_amd import synth

pytestmark = pytest.mark.gpu
TOL = 3e-4


@pytest.fixture(scope="module")
def long_ctx():
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg = synth.BEAT150
    sd = synth.make_state_dict(cfg)
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, nframes=cfg.nframes)
    eng.load_state_dict(sd)
    oracle = orc.RagOracle(sd, cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, nframes=cfg.nframes)
    yield dict(cfg=cfg, eng=eng, orc=orc, oracle=oracle, L=_lib)
    eng.close()


@pytest.mark.parametrize("B", [3, 8])      # 8: 2 * 8 * 152 = 2432 rows = whole 128-row tiles -> the GEMM's LDS-DMA path; 3: its general path
def test_long_prepare_and_forward_vs_oracle(long_ctx, B):
    cfg, eng, orc, oracle = (long_ctx[k] for k in ("cfg", "eng", "orc", "oracle"))
    y = synth.make_cond(cfg, B)
    eng.prepare(y)
    prep = oracle.prepare(y)
    assert max_abs(eng.read("audio_feat"), prep["af"]) < 5e-5
    assert max_abs(eng.read("static_c"), prep["static"][0]) < 5e-5 and max_abs(eng.read("static_u"), prep["static"][1]) < 5e-5
    g = np.random.Generator(np.random.PCG64(5))
    x = g.standard_normal((B, cfg.njoints, cfg.nfeats, cfg.nframes)).astype(np.float32)
    eps = g.standard_normal((2, B, 512)).astype(np.float32)
    for t in ((0, 700) if B == 3 else (300,)):
        oc, ou, og = eng.forward(x, np.full((B,), t), eps[0], eps[1])
        wc = oracle.forward(x, np.full((B,), t), y, False, eps[0])
        wu = oracle.forward(x, np.full((B,), t), y, True, eps[1])
        d = max(max_abs(oc, wc), max_abs(ou, wu), max_abs(og, wu + 1.5 * (wc - wu)))
        print(f"150-frame forward t={t}: max|hip - oracle| = {d:.3e}")
        assert d < TOL


@pytest.mark.parametrize("ddim", [False, True])
def test_long_loops_vs_oracle(long_ctx, ddim):
    cfg, eng, orc, oracle, L = (long_ctx[k] for k in ("cfg", "eng", "orc", "oracle", "L"))
    B = 2
    y = synth.make_cond(cfg, B)
    sch = orc.Schedule(1000, "ddim100") if ddim else orc.Schedule(12, "")
    skip = 92 if ddim else 0
    n_exec = sch.num_timesteps - skip
    eng.set_schedule(sch)
    eng.prepare(y)
    tape = synth.NoiseTape(cfg, B, n_exec)
    init = synth.make_init_image(cfg, B) if ddim else None
    got = eng.sample(sampler=L.LS_SAMPLER_DDIM if ddim else L.LS_SAMPLER_DDPM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise,
                     skip_timesteps=skip, init_image=init)
    assert eng.timing()["single_pass"] == 0
    want = orc.sample_loop(oracle, sch, y, tape.x_init, tape.eps, tape.noise, ddim=ddim, skip_timesteps=skip, init_image=init)
    d = max_abs(got, want)
    print(f"150-frame {'DDIM' if ddim else 'DDPM'} loop ({n_exec} steps): max|hip - oracle| = {d:.3e}")
    assert d < TOL
    again = eng.sample(sampler=L.LS_SAMPLER_DDIM if ddim else L.LS_SAMPLER_DDPM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise,
                       skip_timesteps=skip, init_image=init, use_graph=False)
    assert np.array_equal(got, again)                      # graph replay == plain launches, bit for bit


def test_long_philox_mode_and_limits(long_ctx):
    from oracle import philox_oracle as po
    cfg, eng, orc, oracle, L = (long_ctx[k] for k in ("cfg", "eng", "orc", "oracle", "L"))
    B, steps, seed, off = 2, 5, 99, 40
    y = synth.make_cond(cfg, B)
    sch = orc.Schedule(steps, "")
    eng.set_schedule(sch)
    eng.prepare(y)
    got = eng.sample(sampler=L.LS_SAMPLER_DDPM, philox_seed=seed, sample_offset=off)
    gidx = off + np.arange(B)
    eps, noise = po.step_tapes(seed, gidx, steps, (cfg.njoints, cfg.nfeats, cfg.nframes))
    want = orc.sample_loop(oracle, sch, y, po.x_init(seed, gidx, cfg.jf, cfg.nframes, (cfg.njoints, cfg.nfeats)), eps, noise)
    assert max_abs(got, want) < TOL
    with pytest.raises(L.EngineError, match="fp32 only"):
        eng.set_precision("bf16x3")


def test_sequences_longer_than_the_fused_token_mixing_limit_take_the_gemm_form():
    """S > 160 tokens: the token-mixing operand no longer fits the fused kernel's LDS budget, so the step falls back to LayerNorm +
    a batched, transposed GEMM per sequence.  TED layout with 200 frames (214 745 samples -> 43 587 -> 7 263 -> 1 209 -> 200)."""
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg = synth.PathConfig("ted200", njoints=9, nfeats=3, nframes=200, n_prefix_tokens=1, audio_len=214745)
    assert synth.audio_lengths(cfg.audio_len)[-1] == 200
    sd = synth.make_state_dict(cfg)
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, nframes=cfg.nframes)
    try:
        eng.load_state_dict(sd)
        oracle = orc.RagOracle(sd, cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, nframes=cfg.nframes)
        B, steps = 2, 4
        y = synth.make_cond(cfg, B)
        sch = orc.Schedule(steps, "")
        eng.set_schedule(sch)
        eng.prepare(y)
        tape = synth.NoiseTape(cfg, B, steps)
        got = eng.sample(sampler=_lib.LS_SAMPLER_DDPM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise)
        want = orc.sample_loop(oracle, sch, y, tape.x_init, tape.eps, tape.noise)
        d = max_abs(got, want)
        print(f"200-frame TED-layout loop (GEMM-form token mixing): max|hip - oracle| = {d:.3e}")
        assert d < TOL
    finally:
        eng.close()


def test_long_single_steps_vs_oracle(long_ctx):
    """ls_step (p_sample / ddim_sample one step at a time, what a step-by-step caller uses) on the long-sequence path."""
    cfg, eng, orc, oracle, L = (long_ctx[k] for k in ("cfg", "eng", "orc", "oracle", "L"))
    B = 2
    y = synth.make_cond(cfg, B)
    g = np.random.Generator(np.random.PCG64(77))
    x = g.standard_normal((B, cfg.njoints, cfg.nfeats, cfg.nframes)).astype(np.float32)
    eps = g.standard_normal((2, B, 512)).astype(np.float32)
    noise = g.standard_normal(x.shape).astype(np.float32)
    oracle.prepare(y)
    for name, resp, sampler, idx in (("p", "", L.LS_SAMPLER_DDPM, 40), ("ddim", "ddim100", L.LS_SAMPLER_DDIM, 0)):
        sch = orc.Schedule(1000 if resp else 50, resp)
        eng.set_schedule(sch)
        eng.prepare(y)
        out, x0 = eng.step(sampler, idx, x, eps[0], eps[1], noise)
        w0 = oracle.cfg_forward(x, np.full((B,), sch.timestep_map[idx]), y, eps[0], eps[1])
        want = orc.p_sample_update(sch, x, w0, idx, noise) if name == "p" else orc.ddim_update(sch, x, w0, idx, noise)
        assert max_abs(x0, w0) < TOL and max_abs(out, want) < TOL, name


@pytest.mark.parametrize("B", [3, 8, 32, 40])
def test_one_launch_mixer_agrees_with_the_batch_level_kernels_and_the_oracle(B):
    """The eight blocks of the 150-frame model in ONE launch (csrc/ls_mix_kernel.h: four slice workgroups per (sample, pass), k blocks through
    an LDS ring) against the sixteen batch-level launches it replaces (ls_set_path 2) -- same arithmetic, another summation order -- and, for
    the first and the last sample, against the CPU oracle on the restated Philox noise.  40 clips = 80 groups: two mixer launches per step."""
    from livelyspeaker_amd import _lib
    from oracle import philox_oracle as po
    from oracle import rag_oracle as orc
    cfg = synth.BEAT150
    sd = synth.make_state_dict(cfg)
    steps, seed, off = 6, 4242 + B, 100
    y = synth.make_cond(cfg, B)
    sch = orc.Schedule(steps, "")
    outs = {}
    for path in ("auto", "batch"):
        eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, nframes=cfg.nframes)
        try:
            eng.load_state_dict(sd)
            eng.set_path("batch" if path == "batch" else "coop")      # "coop" on a long-sequence model = the one-launch mixer at any batch size (`auto` takes it from 28 clips)
            eng.set_schedule(sch)
            eng.prepare(y)
            outs[path] = eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=seed, sample_offset=off)
            assert np.array_equal(outs[path], eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=seed, sample_offset=off))       # graph replay, deterministic
        finally:
            eng.close()
    d = max_abs(outs["auto"], outs["batch"])
    assert np.isfinite(outs["auto"]).all() and 0 < d < 1e-4, d
    pick = np.array([0, B - 1])
    oracle = orc.RagOracle(sd, cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, nframes=cfg.nframes)
    eps, noise = po.step_tapes(seed, off + pick, steps, (cfg.njoints, cfg.nfeats, cfg.nframes))
    x_T = po.x_init(seed, off + pick, cfg.njoints * cfg.nfeats, cfg.nframes, (cfg.njoints, cfg.nfeats))
    want = orc.sample_loop(oracle, sch, {k: v[pick] for k, v in y.items()}, x_T, eps, noise)
    do = max_abs(outs["auto"][pick], want)
    print(f"150 frames, B = {B}: one-launch mixer vs batch-level kernels {d:.3e}, vs oracle {do:.3e}")
    assert do < TOL


def test_mixer_forced_on_the_step_by_step_surface(long_ctx):
    """An engine with the one-launch mixer forced (`coop` on a long-sequence model) on the rest of the sampler's surface: the CFG forward and
    single p_sample / ddim_sample steps (per-sample timesteps: these keep the batch-level kernels, whatever the mode), then a taped DDIM loop with
    skipped timesteps and an init image ON the mixer (host-drawn style eps and step noise, poseFinal's per-slice partial products summed by the
    update kernel); graph replay == plain launches."""
    from livelyspeaker_amd import _lib
    cfg, orc, oracle, L = (long_ctx[k] for k in ("cfg", "orc", "oracle", "L"))
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, nframes=cfg.nframes)
    try:
        eng.load_state_dict(synth.make_state_dict(cfg))
        eng.set_path("coop")
        B = 3
        y = synth.make_cond(cfg, B)
        g = np.random.Generator(np.random.PCG64(91))
        x = g.standard_normal((B, cfg.njoints, cfg.nfeats, cfg.nframes)).astype(np.float32)
        eps = g.standard_normal((2, B, 512)).astype(np.float32)
        noise = g.standard_normal(x.shape).astype(np.float32)
        oracle.prepare(y)
        eng.set_schedule(orc.Schedule(50, ""))
        eng.prepare(y)
        oc, ou, og = eng.forward(x, np.full((B,), 17), eps[0], eps[1])
        wc, wu = oracle.forward(x, np.full((B,), 17), y, False, eps[0]), oracle.forward(x, np.full((B,), 17), y, True, eps[1])
        d = max(max_abs(oc, wc), max_abs(ou, wu), max_abs(og, wu + 1.5 * (wc - wu)))
        print(f"150-frame forward on a mixer-forced engine: max|hip - oracle| = {d:.3e}")
        assert d < TOL
        for name, resp, sampler, idx in (("p", "", L.LS_SAMPLER_DDPM, 40), ("ddim", "ddim100", L.LS_SAMPLER_DDIM, 0)):
            sch = orc.Schedule(1000 if resp else 50, resp)
            eng.set_schedule(sch)
            eng.prepare(y)
            out, x0 = eng.step(sampler, idx, x, eps[0], eps[1], noise)
            w0 = oracle.cfg_forward(x, np.full((B,), sch.timestep_map[idx]), y, eps[0], eps[1])
            want = orc.p_sample_update(sch, x, w0, idx, noise) if name == "p" else orc.ddim_update(sch, x, w0, idx, noise)
            assert max_abs(x0, w0) < TOL and max_abs(out, want) < TOL, name
        sch = orc.Schedule(1000, "ddim100")
        skip = 94
        eng.set_schedule(sch)
        eng.prepare(y)
        tape = synth.NoiseTape(cfg, B, sch.num_timesteps - skip)
        init = synth.make_init_image(cfg, B)
        got = eng.sample(sampler=L.LS_SAMPLER_DDIM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise, skip_timesteps=skip, init_image=init)
        assert eng.timing()["coop_slices"] == 4
        want = orc.sample_loop(oracle, sch, y, tape.x_init, tape.eps, tape.noise, ddim=True, skip_timesteps=skip, init_image=init)
        assert max_abs(got, want) < TOL
        again = eng.sample(sampler=L.LS_SAMPLER_DDIM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise, skip_timesteps=skip, init_image=init,
                           use_graph=False)
        assert np.array_equal(got, again)
    finally:
        eng.close()
