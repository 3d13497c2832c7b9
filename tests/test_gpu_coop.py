"""The sample-split step kernel (ls_coop_kernel.h, ls_set_path 3 / "coop"): a sample's step spread over 16 workgroups that exchange
LayerNorm partials and rows through L2 inside one launch.  Pinned to the REFERENCE's fixtures like the fused kernel: G2 single
steps, G3 config-1 loop with dumps, G4 ddim100 (skip 80 + init_image, and full), G5 1000 steps, G11 guidance scale 1, BEAT G12 /
G13; plus what is specific to it: determinism under load (a stale hand-off read would show as run-to-run differences), batches
beyond one launch's residency (chunked launches), the single-pass form, per-sample timesteps, concurrency with another handle."""
import os
import threading

import numpy as np
import pytest

from conftest import GOLDEN, max_abs
from livelyspeaker_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.engine_path_auto]
TOL_FWD, TOL_LOOP = 2e-4, 3e-4


def _engine(ds, path="coop"):
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


def _loop(eng, cfg, steps, resp, ddim, skip, use_init, dump=None, use_graph=True, B=4, scale=1.5, two_pass_always=False):
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    sch = orc.Schedule(steps, resp)
    eng.set_schedule(sch)
    eng.prepare(synth.make_cond(cfg, B, scale=scale))
    n_exec = sch.num_timesteps - skip
    tape = synth.NoiseTape(cfg, B, n_exec)
    init = synth.make_init_image(cfg, B) if use_init else None
    return eng.sample(sampler=_lib.LS_SAMPLER_DDIM if ddim else _lib.LS_SAMPLER_DDPM, x_init=tape.x_init, eps_tape=tape.eps,
                      noise_tape=tape.noise, init_image=init, skip_timesteps=skip, dump_steps=dump, use_graph=use_graph,
                      two_pass_always=two_pass_always)


# the slicings of the sample-split kernel (ls_set_path 8 | 6 | 7): 8 | 4 | 2 slice workgroups per (sample, CFG pass); "coop" lets the
# step-time model choose per piece.  Every one is pinned to the reference's fixtures.
FORMS = ["coop8", "coop4", "coop2"]


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("ds", ["ted", "beat"])
def test_reference_fixtures_on_the_sample_split_kernel(ds, form, golden):
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg, eng = _engine(ds, form)
    g = golden[ds]
    try:
        # model(x, t, y) itself (G1): cond / uncond outputs at four timesteps
        x, eps, noise = _g1_inputs(cfg)
        eng.prepare(synth.make_cond(cfg, 4))
        for t in (0, 5, 500, 999):
            oc, ou, _ = eng.forward(x, np.full(4, t, np.int64), eps[0], eps[1])
            assert max_abs(oc, g[f"G1_t{t}_c"]) < TOL_FWD and max_abs(ou, g[f"G1_t{t}_u"]) < TOL_FWD, t
        # single p_sample / ddim_sample steps (G2)
        for name, resp, steps in (("p", "", (0, 7, 999)), ("ddim", "ddim100", (0, 50, 99))):
            eng.set_schedule(orc.Schedule(1000, resp))
            for t in steps:
                s, x0 = eng.step(_lib.LS_SAMPLER_DDPM if name == "p" else _lib.LS_SAMPLER_DDIM, t, x, eps[0], eps[1], noise)
                assert max_abs(s, g[f"G2_{name}_t{t}_sample"]) < TOL_FWD and max_abs(x0, g[f"G2_{name}_t{t}_x0"]) < TOL_FWD, (name, t)
        # config 1: B = 4, 50-step DDPM, CFG 1.5 (G3), with pred_xstart dumps; hipGraph replay == plain launches, bitwise
        out, dumps = _loop(eng, cfg, 50, "", False, 0, False, dump=[0, 25, 49])
        assert eng.timing()["step_path"] == 2
        d3 = max_abs(out, g["G3_ddpm50_final"])
        if ds == "ted":
            for k, dmp in zip((0, 25, 49), dumps):
                assert max_abs(dmp, g[f"G3_ddpm50_dump_x0_step{k}"]) < TOL_LOOP, k
        assert np.array_equal(out, _loop(eng, cfg, 50, "", False, 0, False, use_graph=False))
        assert np.array_equal(out, _loop(eng, cfg, 50, "", False, 0, False))            # the cached graph, replayed
        # the LivelySpeaker refine schedule (G4)
        d4 = max_abs(_loop(eng, cfg, 1000, "ddim100", True, 80, True), g["G4_ddim100_skip80_final"])
        print(f"{ds} [{form}]: G3 {d3:.3e}  G4 skip80 {d4:.3e}")
        assert d3 < TOL_LOOP and d4 < TOL_LOOP
        if ds == "ted":
            d5 = max_abs(_loop(eng, cfg, 1000, "", False, 0, False), g["G5_ddpm1000_final"])
            d4f = max_abs(_loop(eng, cfg, 1000, "ddim100", True, 0, False), g["G4_ddim100_full_final"])
            print(f"      G5 1000 steps {d5:.3e}  G4 full {d4f:.3e}")
            assert d5 < TOL_LOOP and d4f < TOL_LOOP
    finally:
        eng.close()


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("ds", ["ted", "beat"])
def test_round2_fixtures_scale1_and_beat_loops_on_the_sample_split_kernel(ds, form):
    """G11: guidance scale 1 with an odd batch, both as the reference evaluates it (two passes) and in the single-pass form (8
    workgroups per sample); G12 / G13: BEAT 1000-step DDPM and full ddim100."""
    g = np.load(os.path.join(GOLDEN, f"{ds}_golden_r2.npz"))
    cfg, eng = _engine(ds, form)
    try:
        for two in (True, False):
            d1 = max_abs(_loop(eng, cfg, 50, "", False, 0, False, B=5, scale=1.0, two_pass_always=two), g["G11_scale1_ddpm50_B5_final"])
            assert eng.timing()["single_pass"] == (0 if two else 1)
            d2 = max_abs(_loop(eng, cfg, 1000, "ddim100", True, 80, True, B=5, scale=1.0, two_pass_always=two), g["G11_scale1_ddim100_skip80_B5_final"])
            print(f"{ds} [{form}]: G11 ddpm50 {d1:.3e}, ddim100/skip80 {d2:.3e} (two passes: {two})")
            assert d1 < TOL_LOOP and d2 < TOL_LOOP
        if ds == "beat":
            for key, args in (("G12_ddpm1000_final", (1000, "", False, 0, False)), ("G13_ddim100_full_final", (1000, "ddim100", True, 0, False))):
                if key in g:
                    d = max_abs(_loop(eng, cfg, *args), g[key])
                    print(f"beat {key}: {d:.3e}")
                    assert d < TOL_LOOP
    finally:
        eng.close()


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("ds,B", [("ted", 9), ("ted", 40), ("ted", 77), ("beat", 32), ("beat", 70)])
def test_sample_split_agrees_with_the_fused_kernel_across_launch_chunks(ds, B, form):
    """One launch holds what is resident at once (8 / 4 slices: 32 samples with two passes, 64 with one; 2 slices: 64 / 128); larger
    batches run as several launches per step.  Every sample must come out as the fused kernel computes it (to summation order) wherever
    it falls in the chunking."""
    cfg = synth.CONFIGS[ds]
    outs = {}
    for path in ("fused", form):
        _, eng = _engine(ds, path)
        try:
            outs["fused" if path == "fused" else "coop"] = _loop(eng, cfg, 12, "", False, 0, False, B=B)
            if path != "fused":
                assert eng.timing()["step_path"] == 2
                outs["coop1"] = _loop(eng, cfg, 12, "", False, 0, False, B=B, scale=1.0)
                assert eng.timing()["single_pass"] == 1
            else:
                outs["fused1"] = _loop(eng, cfg, 12, "", False, 0, False, B=B, scale=1.0)
        finally:
            eng.close()
    d, d1 = max_abs(outs["fused"], outs["coop"]), max_abs(outs["fused1"], outs["coop1"])
    print(f"{ds} B = {B} [{form}]: fused vs sample-split, 12 steps: {d:.3e}; single pass {d1:.3e}")
    assert 0 < d < 5e-5 and 0 < d1 < 5e-5


def test_determinism_under_load_and_next_to_another_handle():
    """A stale or torn hand-off read would almost surely differ from run to run.  Twenty replays of a 25-step loop on a full chip
    (B = 32: 512 workgroups, two per CU) must be bitwise identical -- alone, and while a second handle (fused kernel, other
    stream, driven from another thread) keeps the chip busy and perturbs which workgroups are resident when."""
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg, eng = _engine("ted")
    _, other = _engine("ted", "fused")
    try:
        sch = orc.Schedule(25, "")
        eng.set_schedule(sch)
        eng.prepare(synth.make_cond(cfg, 32))
        other.set_schedule(orc.Schedule(200, ""))
        other.prepare(synth.make_cond(cfg, 300))
        ref = eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=11)
        for _ in range(10):
            assert np.array_equal(ref, eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=11))
        stop = threading.Event()
        errs = []

        def hammer():
            try:
                while not stop.is_set():
                    other.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=3)
            except Exception as e:      # noqa: BLE001
                errs.append(e)

        th = threading.Thread(target=hammer)
        th.start()
        try:
            for _ in range(10):
                assert np.array_equal(ref, eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=11))
        finally:
            stop.set()
            th.join()
        assert not errs, errs
    finally:
        eng.close()
        other.close()


def test_two_sample_split_handles_at_once_do_not_starve_each_other():
    """Two handles whose step kernels BOTH wait on their own workgroups (512 + 512 workgroups competing for the chip's 512 slots): the
    slices of a group have consecutive block ids and are dispatched together, so neither kernel can hold slots while waiting for
    blocks that cannot start.  Results bitwise the sequential ones; a starved hand-off would end in the bounded spin's error."""
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg = synth.TED
    engs = []
    for B in (32, 30):
        _, e = _engine("ted")
        e.set_schedule(orc.Schedule(30, ""))
        e.prepare(synth.make_cond(cfg, B, seed=B))
        engs.append(e)
    try:
        seq = [e.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=21 + i) for i, e in enumerate(engs)]
        got, errs = [None, None], []
        bar = threading.Barrier(2)

        def work(i):
            try:
                bar.wait()
                for _ in range(6):
                    got[i] = engs[i].sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=21 + i)
            except Exception as ex:     # noqa: BLE001
                errs.append(ex)
        ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        assert not errs, errs
        assert np.array_equal(got[0], seq[0]) and np.array_equal(got[1], seq[1])
    finally:
        for e in engs:
            e.close()


def test_auto_takes_the_sample_split_kernel_for_small_batches_only():
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg, eng = _engine("ted", "auto")
    try:
        eng.set_schedule(orc.Schedule(4, ""))
        for B, want in ((4, 2), (32, 2), (512, 0)):
            eng.prepare(synth.make_cond(cfg, B))
            out = eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=5)
            assert np.isfinite(out).all() and eng.timing()["step_path"] == want, B
        # the split-precision mode exists in the fused and the one-pass-per-workgroup kernels only: never the sample-split kernel
        eng.set_precision("bf16x3")
        for B, want in ((6, 3), (512, 0)):
            eng.prepare(synth.make_cond(cfg, B))
            eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=5)
            assert eng.timing()["step_path"] == want, B
    finally:
        eng.close()


@pytest.mark.parametrize("ds,B,want_tail", [("ted", 300, (44, 2, 0, 0)), ("ted", 400, (128, 3, 16, 2)), ("beat", 288, (32, 2, 0, 0))])
def test_partial_last_round_goes_to_the_small_batch_kernels(ds, B, want_tail):
    """256 k + r samples on the fused kernel pay k + 1 full rounds.  `auto` runs the k full rounds on it and the r samples on the
    sample-split, the one-pass-per-workgroup or the batch-level kernels, or on two of them (whatever the step-time model says is
    cheapest): every sample as the fused kernel alone computes it (to summation order), in TAPE and in PHILOX mode (the tail's Philox
    streams are keyed by its global sample index)."""
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg = synth.CONFIGS[ds]
    outs = {}
    for path in ("fused", "auto"):
        _, eng = _engine(ds, path)
        try:
            outs[path] = _loop(eng, cfg, 6, "", False, 0, False, B=B)
            t = eng.timing()
            assert t["step_path"] == 0 and (t["tail_samples"], t["tail_path"], t["tail2_samples"], t["tail2_path"]) == (want_tail if path == "auto" else (0, 0, 0, 0)), t
            outs[path + "_philox"] = eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=17, sample_offset=1000)
        finally:
            eng.close()
    d, dp = max_abs(outs["fused"], outs["auto"]), max_abs(outs["fused_philox"], outs["auto_philox"])
    nf = B - want_tail[0] - want_tail[2]
    print(f"{ds} B = {B}: fused vs fused + tail {want_tail}: tape {d:.3e}, philox {dp:.3e}")
    assert np.array_equal(outs["fused"][:nf], outs["auto"][:nf]) and np.array_equal(outs["fused_philox"][:nf], outs["auto_philox"][:nf])
    assert 0 < d < 5e-5 and 0 < dp < 5e-5
