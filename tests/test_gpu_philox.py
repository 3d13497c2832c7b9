"""GPU parity of the THROUGHPUT mode (``noise_source='philox'``, the mode bench.py times): the device RNG against its
numpy restatement (oracle/philox_oracle.py, pinned to Random123's known answers), then whole Philox-mode sampling loops on
the HIP path against the CPU oracle fed with the restated noise -- including the headline workload, batch 512 x 1000 DDPM
steps (three samples through the oracle).  Draw order these streams stand in for: gaussian_diffusion.py:700-743."""
import numpy as np
import pytest

from conftest import max_abs
from livelyspeaker_amd import synth

pytestmark = pytest.mark.gpu

TOL_RNG = 4e-6        # v_log_f32 / v_sin_f32 / v_cos_f32 vs float64 libm on |z| <= 6.7
TOL_LOOP = 3e-4


def _engine(ds):
    from livelyspeaker_amd import _lib
    cfg = synth.CONFIGS[ds]
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions)
    eng.load_state_dict(synth.make_state_dict(cfg))
    return cfg, eng


def _oracle(cfg):
    from oracle import rag_oracle as orc
    return orc, orc.RagOracle(synth.make_state_dict(cfg), cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens)


@pytest.mark.parametrize("ds", ["ted", "beat"])
def test_device_x_init_matches_the_numpy_restatement(ds):
    from oracle import philox_oracle as po
    cfg, eng = _engine(ds)
    try:
        for seed, off in ((7, 0), (2 ** 61 + 12345, 300), (1, 2 ** 33 + 17)):
            got = eng.philox_x_init(16, seed=seed, sample_offset=off)
            want = po.x_init(seed, off + np.arange(16), cfg.njoints * cfg.nfeats, cfg.nframes, (cfg.njoints, cfg.nfeats))
            d = max_abs(got, want)
            print(f"{ds} x_T seed={seed} offset={off}: max|d| = {d:.3e}")
            assert d < TOL_RNG
    finally:
        eng.close()


@pytest.mark.parametrize("ds,steps,resp,ddim,skip,off", [("ted", 50, "", False, 0, 0), ("ted", 1000, "ddim100", True, 80, 1000),
                                                          ("beat", 30, "", False, 0, 5), ("beat", 1000, "ddim100", True, 90, 0)])
def test_philox_mode_loop_vs_oracle_on_restated_noise(ds, steps, resp, ddim, skip, off):
    """Every stream id (x_T, cond / uncond style eps, step noise), the step counter and the sample offset enter the result:
    a wrong id or an off-by-one step_id would show up as O(1) differences."""
    from livelyspeaker_amd import _lib
    from oracle import philox_oracle as po
    cfg, eng = _engine(ds)
    orc, oracle = _oracle(cfg)
    try:
        B, seed = 4, 424242 + steps
        y = synth.make_cond(cfg, B)
        sch = orc.Schedule(steps, resp)
        eng.set_schedule(sch)
        eng.prepare(y)
        init = None
        if skip:
            init = np.random.Generator(np.random.PCG64(3)).standard_normal((B, cfg.njoints, cfg.nfeats, cfg.nframes)).astype(np.float32) * 0.3
        got = eng.sample(sampler=_lib.LS_SAMPLER_DDIM if ddim else _lib.LS_SAMPLER_DDPM, philox_seed=seed, sample_offset=off,
                         skip_timesteps=skip, init_image=init)
        gidx = off + np.arange(B)
        n_exec = sch.num_timesteps - skip
        eps, noise = po.step_tapes(seed, gidx, n_exec, (cfg.njoints, cfg.nfeats, cfg.nframes))
        x_T = po.x_init(seed, gidx, cfg.njoints * cfg.nfeats, cfg.nframes, (cfg.njoints, cfg.nfeats))
        want = orc.sample_loop(oracle, sch, y, x_T, eps, noise, ddim=ddim, skip_timesteps=skip, init_image=init)
        d = max_abs(got, want)
        print(f"{ds} philox {n_exec}-step {'DDIM' if ddim else 'DDPM'} offset={off}: max|hip - oracle| = {d:.3e}")
        assert d < TOL_LOOP
        if off:      # and a deliberately wrong offset is far away (the check has teeth)
            bad = orc.sample_loop(oracle, sch, y, po.x_init(seed, gidx - 1, cfg.njoints * cfg.nfeats, cfg.nframes, (cfg.njoints, cfg.nfeats)),
                                  eps, noise, ddim=ddim, skip_timesteps=skip, init_image=init, max_steps=2)
            assert max_abs(bad, want) > 1e-2 or ddim
    finally:
        eng.close()


def test_headline_workload_batch512_1000_steps_spot_check():
    """BASELINE configs[1] end to end in the timed mode: TED, B=512, 1000-step DDPM, CFG 1.5, Philox noise, hipGraph;
    samples 0, 257 and 511 replayed alone through the CPU oracle on the restated noise (the HIP side takes 1.5 s)."""
    from livelyspeaker_amd import _lib
    from oracle import philox_oracle as po
    cfg, eng = _engine("ted")
    orc, oracle = _oracle(cfg)
    try:
        B, steps, seed, off = 512, 1000, 20260928, 4096 - 512
        y = synth.make_cond(cfg, B, scale=1.5)
        sch = orc.Schedule(steps, "")
        eng.set_schedule(sch)
        eng.prepare(y)
        got = eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=seed, sample_offset=off, use_graph=True)
        assert np.isfinite(got).all()
        pick = np.array([0, 257, 511])
        eps, noise = po.step_tapes(seed, off + pick, steps, (cfg.njoints, cfg.nfeats, cfg.nframes))
        x_T = po.x_init(seed, off + pick, cfg.njoints * cfg.nfeats, cfg.nframes, (cfg.njoints, cfg.nfeats))
        want = orc.sample_loop(oracle, sch, {k: v[pick] for k, v in y.items()}, x_T, eps, noise)
        d = max_abs(got[pick], want)
        print(f"B=512 x 1000 DDPM steps (Philox, graph): max|hip - oracle| over 3 samples = {d:.3e}")
        assert d < TOL_LOOP
    finally:
        eng.close()


@pytest.mark.engine_path_auto
@pytest.mark.parametrize("ds,B", [("ted", 160), ("ted", 416), ("beat", 192), ("ted", 72), ("ted", 48), ("beat", 88)])
def test_multi_piece_plans_replayed_through_the_oracle(ds, B):
    """Ragged batches (the last iteration of the reference's loaders, scripts/test_RAG_ted.py:43-82) run on plans of several pieces --
    full rounds on the fused kernel, a chip's worth on the one-pass-per-workgroup kernel, the rest on the sample-split kernel.  The first
    and the last sample of EVERY piece of the plan `auto` makes are replayed alone through the CPU oracle on the restated Philox noise
    (50-step DDPM, CFG 1.5: cfg_sampler.py:24-31, gaussian_diffusion.py:608-743), as the B = 512 check above does for the headline."""
    from livelyspeaker_amd import _lib
    from oracle import philox_oracle as po
    cfg = synth.CONFIGS[ds]
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, path="auto")
    eng.load_state_dict(synth.make_state_dict(cfg))
    orc, oracle = _oracle(cfg)
    try:
        steps, seed, off = 50, 777 + B, 1000
        y = synth.make_cond(cfg, B, scale=1.5)
        sch = orc.Schedule(steps, "")
        eng.set_schedule(sch)
        eng.prepare(y)
        got = eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=seed, sample_offset=off, use_graph=True)
        assert np.isfinite(got).all()
        tm = eng.timing()
        pieces, _ = _lib.plan_query(B, dataset=ds, n_cus=tm["n_cus"])
        assert sum(n for _, _, n in pieces) == B
        assert tm["step_path"] == pieces[0][0] and tm["tail_samples"] == (pieces[1][2] if len(pieces) > 1 else 0)     # the plan that ran
        pick = np.array(sorted({i for _, first, n in pieces for i in (first, first + n - 1)}))
        eps, noise = po.step_tapes(seed, off + pick, steps, (cfg.njoints, cfg.nfeats, cfg.nframes))
        x_T = po.x_init(seed, off + pick, cfg.njoints * cfg.nfeats, cfg.nframes, (cfg.njoints, cfg.nfeats))
        want = orc.sample_loop(oracle, sch, {k: v[pick] for k, v in y.items()}, x_T, eps, noise)
        per = np.abs(got[pick].astype(np.float64) - want).reshape(len(pick), -1).max(axis=1)
        print(f"{ds} B={B}: plan {pieces}; samples {pick.tolist()}: max|hip - oracle| per sample {[float(f'{v:.2e}') for v in per]}")
        assert per.max() < TOL_LOOP
    finally:
        eng.close()


def test_beat_caller_batch_256_1000_steps_spot_check():
    """BASELINE configs[4] at the frame count the reference can run (34): BEAT, B=256, 1000-step DDPM, Philox noise."""
    from livelyspeaker_amd import _lib
    from oracle import philox_oracle as po
    cfg, eng = _engine("beat")
    orc, oracle = _oracle(cfg)
    try:
        B, steps, seed, off = 256, 1000, 31337, 256
        y = synth.make_cond(cfg, B, scale=1.5)
        sch = orc.Schedule(steps, "")
        eng.set_schedule(sch)
        eng.prepare(y)
        got = eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=seed, sample_offset=off)
        assert np.isfinite(got).all()
        pick = np.array([3, 255])
        eps, noise = po.step_tapes(seed, off + pick, steps, (cfg.njoints, cfg.nfeats, cfg.nframes))
        x_T = po.x_init(seed, off + pick, cfg.njoints * cfg.nfeats, cfg.nframes, (cfg.njoints, cfg.nfeats))
        want = orc.sample_loop(oracle, sch, {k: v[pick] for k, v in y.items()}, x_T, eps, noise)
        d = max_abs(got[pick], want)
        print(f"BEAT B=256 x 1000 DDPM steps (Philox): max|hip - oracle| over 2 samples = {d:.3e}")
        assert d < TOL_LOOP
    finally:
        eng.close()


@pytest.mark.parametrize("key,steps,resp,ddim", [("G12_ddpm1000_final", 1000, "", False), ("G13_ddim100_full_final", 1000, "ddim100", True)])
def test_beat_long_loops_vs_reference_fixtures(key, steps, resp, ddim):
    """BEAT at TED's depth: 1000-step DDPM and the full 100-step DDIM loop against fixtures produced by the reference."""
    import os
    from conftest import GOLDEN
    from livelyspeaker_amd import _lib
    g = np.load(os.path.join(GOLDEN, "beat_golden_r2.npz"))
    cfg, eng = _engine("beat")
    orc, _ = _oracle(cfg)
    try:
        sch = orc.Schedule(steps, resp)
        eng.set_schedule(sch)
        eng.prepare(synth.make_cond(cfg, 4))
        tape = synth.NoiseTape(cfg, 4, sch.num_timesteps)
        out = eng.sample(sampler=_lib.LS_SAMPLER_DDIM if ddim else _lib.LS_SAMPLER_DDPM, x_init=tape.x_init, eps_tape=tape.eps,
                         noise_tape=tape.noise)
        d = max_abs(out, g[key])
        print(f"BEAT {key}: max|hip - reference| = {d:.3e}")
        assert d < TOL_LOOP
    finally:
        eng.close()


def test_config4_global_batch_4096_equals_eight_shards_of_512():
    """BASELINE configs[3]: one global batch of 4096 clips == eight shards of 512 generated with sample_offset (what the eight
    ranks of a node run; here one GPU plays every rank in turn), bit for bit -- 12 DDPM steps, Philox noise, hipGraph."""
    from livelyspeaker_amd import _lib
    cfg, eng = _engine("ted")
    orc, _ = _oracle(cfg)
    try:
        G, W, steps, seed = 4096, 8, 12, 8675309
        y = synth.make_cond(cfg, G)
        eng.set_schedule(orc.Schedule(steps, ""))
        eng.prepare(y)
        whole = eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=seed, sample_offset=0)
        assert whole.shape[0] == G and np.isfinite(whole).all()
        for r in range(W):
            sl = slice(r * G // W, (r + 1) * G // W)
            eng.prepare({k: v[sl] for k, v in y.items()})
            part = eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=seed, sample_offset=sl.start)
            assert np.array_equal(part, whole[sl]), f"shard {r} differs from the global batch"
    finally:
        eng.close()
