"""GPU edge cases of the sampling path through the C-ABI: ragged batch sizes, const_noise, clip_denoised, DDIM eta,
zero init_image with skip, BEAT at its caller batch, device-resident arguments, and the ABI's error behaviour."""
import ctypes

import os

import numpy as np
import pytest

from conftest import max_abs
from livelyspeaker_amd import synth

pytestmark = pytest.mark.gpu
TOL = 3e-4


@pytest.fixture(scope="module")
def ted():
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg = synth.TED
    sd = synth.make_state_dict(cfg)
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len)
    eng.load_state_dict(sd)
    oracle = orc.RagOracle(sd, cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens)
    yield dict(cfg=cfg, eng=eng, orc=orc, oracle=oracle, lib=_lib)
    eng.close()


@pytest.mark.parametrize("B", [1, 3, 17])
def test_ragged_batch_sizes_vs_oracle(ted, B):
    cfg, eng, orc, oracle, L = (ted[k] for k in ("cfg", "eng", "orc", "oracle", "lib"))
    sch = orc.Schedule(6, "")
    y = synth.make_cond(cfg, B, seed=11 + B)
    tape = synth.NoiseTape(cfg, B, 6, seed=5 + B)
    eng.set_schedule(sch)
    eng.prepare(y)
    got = eng.sample(sampler=L.LS_SAMPLER_DDPM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise)
    want = orc.sample_loop(oracle, sch, y, tape.x_init, tape.eps, tape.noise)
    assert max_abs(got, want) < TOL


def test_clip_denoised_and_ddim_eta_vs_oracle(ted):
    cfg, eng, orc, oracle, L = (ted[k] for k in ("cfg", "eng", "orc", "oracle", "lib"))
    B = 4
    y = synth.make_cond(cfg, B)
    sch = orc.Schedule(1000, "ddim100")
    tape = synth.NoiseTape(cfg, B, 100)
    eng.set_schedule(sch)
    eng.prepare(y)
    got = eng.sample(sampler=L.LS_SAMPLER_DDIM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise, eta=0.5,
                     clip_denoised=True)
    want = orc.sample_loop(oracle, sch, y, tape.x_init, tape.eps, tape.noise, ddim=True, eta=0.5, clip_denoised=True)
    d = max_abs(got, want)
    print(f"ddim100 eta=0.5 clip: max|d|={d:.3e}")
    assert d < TOL and float(np.abs(got).max()) < 3.5      # clamped x0 keeps samples bounded
    # clip really changes the result (the unclipped run is what every other test covers)
    unclipped = eng.sample(sampler=L.LS_SAMPLER_DDIM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise, eta=0.5)
    assert max_abs(got, unclipped) > 0.1


def test_const_noise_uses_sample_zero_noise(ted):
    """const_noise=True (gaussian_diffusion.py:545-546, 706-707): every sample gets sample 0's x_T and step noise."""
    cfg, eng, orc, L = ted["cfg"], ted["eng"], ted["orc"], ted["lib"]
    B = 4
    eng.set_schedule(orc.Schedule(8, ""))
    eng.prepare(synth.make_cond(cfg, B))
    tape = synth.NoiseTape(cfg, B, 8)
    x0 = np.repeat(tape.x_init[:1], B, axis=0)
    a = eng.sample(sampler=L.LS_SAMPLER_DDPM, x_init=x0, eps_tape=tape.eps, noise_tape=tape.noise, const_noise=True)
    nz = np.repeat(tape.noise[:, :1], B, axis=1)
    b = eng.sample(sampler=L.LS_SAMPLER_DDPM, x_init=x0, eps_tape=tape.eps, noise_tape=nz)
    assert np.array_equal(a, b)


def test_skip_without_init_image_is_zero_init(ted):
    """skip_timesteps>0 and init_image None -> zeros (gaussian_diffusion.py:709-710)."""
    cfg, eng, orc, oracle, L = (ted[k] for k in ("cfg", "eng", "orc", "oracle", "lib"))
    B = 2
    sch = orc.Schedule(20, "")
    y = synth.make_cond(cfg, B)
    eng.set_schedule(sch)
    eng.prepare(y)
    tape = synth.NoiseTape(cfg, B, 5)
    got = eng.sample(sampler=L.LS_SAMPLER_DDPM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise, skip_timesteps=15)
    want = orc.sample_loop(oracle, sch, y, tape.x_init, tape.eps, tape.noise, skip_timesteps=15)
    also = eng.sample(sampler=L.LS_SAMPLER_DDPM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise, skip_timesteps=15,
                      init_image=np.zeros_like(tape.x_init))
    assert max_abs(got, want) < TOL and np.array_equal(got, also)


def test_device_resident_arguments_match_host_arguments(ted):
    import torch
    cfg, eng, orc, L = ted["cfg"], ted["eng"], ted["orc"], ted["lib"]
    B = 5
    y = synth.make_cond(cfg, B)
    eng.set_schedule(orc.Schedule(6, ""))
    tape = synth.NoiseTape(cfg, B, 6)
    eng.prepare(y)
    host = eng.sample(sampler=L.LS_SAMPLER_DDPM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise)
    dev = torch.device("cuda", 0)
    eng.prepare({k: torch.from_numpy(v).to(dev) for k, v in y.items()})
    out = eng.sample(sampler=L.LS_SAMPLER_DDPM, x_init=torch.from_numpy(tape.x_init).to(dev),
                     eps_tape=torch.from_numpy(tape.eps).to(dev), noise_tape=torch.from_numpy(tape.noise).to(dev))
    assert isinstance(out, torch.Tensor) and out.is_cuda
    assert np.array_equal(out.cpu().numpy(), host)


def test_beat_caller_batch_spot_check():
    """BEAT RAG at the callers' batch (256, scripts_beat/test_RAG_beat.py:186): spot-check 3 clips against the oracle."""
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg = synth.BEAT
    sd = synth.make_state_dict(cfg)
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions)
    try:
        eng.load_state_dict(sd)
        B, steps = 256, 8
        y = synth.make_cond(cfg, B)
        sch = orc.Schedule(1000, "ddim100")
        eng.set_schedule(sch)
        eng.prepare(y)
        tape = synth.NoiseTape(cfg, B, steps)
        got = eng.sample(sampler=_lib.LS_SAMPLER_DDIM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise,
                         skip_timesteps=100 - steps, init_image=synth.make_init_image(cfg, B))
        pick = [0, 100, 255]
        oracle = orc.RagOracle(sd, cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens)
        want = orc.sample_loop(oracle, sch, {k: v[pick] for k, v in y.items()}, tape.x_init[pick], tape.eps[:, :, pick],
                               tape.noise[:, pick], ddim=True, skip_timesteps=100 - steps,
                               init_image=synth.make_init_image(cfg, B)[pick])
        d = max_abs(got[pick], want)
        print(f"BEAT B=256 spot check max|d|={d:.3e}")
        assert d < TOL and np.isfinite(got).all()
    finally:
        eng.close()


def test_shipped_library_ignores_debug_environment(ted):
    """LS_ABLATE / LS_PROF (honoured only by the -DLS_DEBUG profiling variant) must not change the product's results."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = ("import sys, zlib, numpy as np; sys.path.insert(0, %r)\n"
             "from livelyspeaker_amd import _lib, synth\n"
             "from oracle import rag_oracle as orc\n"
             "cfg = synth.TED\n"
             "eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len)\n"
             "eng.load_state_dict(synth.make_state_dict(cfg)); eng.set_schedule(orc.Schedule(6, '')); eng.prepare(synth.make_cond(cfg, 3))\n"
             "print('CRC', zlib.crc32(np.ascontiguousarray(eng.sample(sampler=0, philox_seed=5)).tobytes()))\n") % root
    crcs = []
    for extra in ({}, {"LS_ABLATE": "7", "LS_PROF": "0", "LS_LIB": "/nonexistent.so"}):
        env = dict(os.environ, **extra)
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        crcs.append([l for l in out.stdout.splitlines() if l.startswith("CRC")][-1])
    assert crcs[0] == crcs[1], crcs


def test_abi_error_behaviour(ted):
    """Negative codes + messages, never a crash: call-order and argument errors (include/ls_hip.h conventions)."""
    L, cfg = ted["lib"], ted["cfg"]
    lib = L.load_library()
    c = L.LsConfig(cfg.njoints, cfg.nfeats, 34, 1, 4, 512, 8, cfg.audio_len, 1400, 0, 0, 0)
    h = ctypes.c_void_p()
    assert lib.ls_create(ctypes.byref(c), ctypes.byref(h)) == 0
    try:
        assert lib.ls_commit_weights(h) == -2 and b"missing weight" in lib.ls_last_error(h)
        cond = L.LsCond(1, 0, None, None, None, None, None)
        assert lib.ls_prepare(h, ctypes.byref(cond)) == -2                 # before weights
        assert lib.ls_prepare_async(h, ctypes.byref(cond)) == -2           # the asynchronous form validates the same way
        w = np.zeros(7, np.float32)
        assert lib.ls_set_weight(h, b"input_mapping.bias", w.ctypes.data_as(L.c_f32p), w.size) == 0
        assert lib.ls_commit_weights(h) != 0                                # wrong size / still missing keys
        a = L.LsSampleArgs()
        assert lib.ls_sample(h, ctypes.byref(a)) == -2 and b"before" in lib.ls_last_error(h)
        assert lib.ls_set_schedule(h, None) == -1
    finally:
        lib.ls_destroy(h)
    eng, orc = ted["eng"], ted["orc"]
    eng.set_schedule(orc.Schedule(4, ""))
    eng.prepare(synth.make_cond(cfg, 2))
    tape = synth.NoiseTape(cfg, 2, 4)
    with pytest.raises(L.EngineError, match="DDPM only"):
        eng.sample(sampler=L.LS_SAMPLER_DDIM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise, dump_steps=[0])
    with pytest.raises(L.EngineError, match="skip_timesteps"):
        eng.sample(sampler=L.LS_SAMPLER_DDPM, x_init=tape.x_init, eps_tape=tape.eps[:0], noise_tape=tape.noise[:0], skip_timesteps=4)
    with pytest.raises(L.EngineError):
        eng.step(L.LS_SAMPLER_DDPM, 9, tape.x_init, tape.eps[0, 0], tape.eps[0, 1], tape.noise[0])


# ------------------------------------------------------------------------------------------------
# Opt-in bf16x3 split-precision mode: same contract (1e-3 max-abs vs the reference), looser than fp32 noise
@pytest.mark.parametrize("mode", ["bf16x3"])
@pytest.mark.parametrize("ds", ["ted", "beat"])
def test_bf16x3_mode_meets_the_parity_contract(ds, mode, golden):
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg = synth.CONFIGS[ds]
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions)
    try:
        eng.load_state_dict(synth.make_state_dict(cfg))
        eng.set_precision(mode)
        g = golden[ds]
        B = 4
        y = synth.make_cond(cfg, B)
        eng.prepare(y)
        gen = np.random.Generator(np.random.PCG64(1234))
        x = gen.standard_normal((B, cfg.njoints, cfg.nfeats, cfg.nframes)).astype(np.float32)
        eps = gen.standard_normal((2, B, 512)).astype(np.float32)
        worst = 0.0
        for t in (0, 500, 999):
            oc, ou, _ = eng.forward(x, np.full((B,), t), eps[0], eps[1])
            worst = max(worst, max_abs(oc, g[f"G1_t{t}_c"]), max_abs(ou, g[f"G1_t{t}_u"]))
        print(f"{ds} {mode} single forward vs reference: {worst:.3e}")
        assert worst < 1e-3
        runs = [("G3_ddpm50_final", 50, "", False, 0, False), ("G4_ddim100_skip80_final", 1000, "ddim100", True, 80, True)]
        if ds == "ted":
            runs.append(("G5_ddpm1000_final", 1000, "", False, 0, False))
        for key, steps, resp, ddim, skip, use_init in runs:
            sch = orc.Schedule(steps, resp)
            eng.set_schedule(sch)
            tape = synth.NoiseTape(cfg, B, sch.num_timesteps - skip)
            out = eng.sample(sampler=_lib.LS_SAMPLER_DDIM if ddim else _lib.LS_SAMPLER_DDPM, x_init=tape.x_init,
                             eps_tape=tape.eps, noise_tape=tape.noise, skip_timesteps=skip,
                             init_image=synth.make_init_image(cfg, B) if use_init else None)
            d = max_abs(out, g[key])
            print(f"{ds} {mode} {key}: max|d| = {d:.3e} (contract 1e-3)")
            assert d < 1e-3
        eng.set_precision("fp32")
        sch = orc.Schedule(50, "")
        eng.set_schedule(sch)
        tape = synth.NoiseTape(cfg, B, 50)
        out = eng.sample(sampler=_lib.LS_SAMPLER_DDPM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise)
        assert max_abs(out, g["G3_ddpm50_final"]) < 3e-4        # switching back restores the exact path
    finally:
        eng.close()


# ------------------------------------------------------------------------------------------------
# BEAT edge cases (round 1 covered them for TED only) and the single-pass variant at degenerate sizes
@pytest.mark.parametrize("B,scale", [(1, 1.5), (3, 1.5), (1, 1.0), (7, 1.0)])
def test_beat_ragged_batches_both_step_variants_vs_oracle(B, scale):
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg = synth.BEAT
    sd = synth.make_state_dict(cfg)
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions)
    try:
        eng.load_state_dict(sd)
        oracle = orc.RagOracle(sd, cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens)
        sch = orc.Schedule(1000, "ddim100")
        skip = 94
        y = synth.make_cond(cfg, B, seed=21 + B, scale=scale)
        tape = synth.NoiseTape(cfg, B, sch.num_timesteps - skip, seed=9 + B)
        init = synth.make_init_image(cfg, B)
        eng.set_schedule(sch)
        eng.prepare(y)
        got = eng.sample(sampler=_lib.LS_SAMPLER_DDIM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise, skip_timesteps=skip,
                         init_image=init, eta=0.3)
        assert eng.timing()["single_pass"] == int(scale == 1.0)
        want = orc.sample_loop(oracle, sch, y, tape.x_init, tape.eps, tape.noise, ddim=True, skip_timesteps=skip, init_image=init, eta=0.3)
        d = max_abs(got, want)
        print(f"BEAT B={B} scale={scale}: max|hip - oracle| = {d:.3e}")
        assert d < TOL
    finally:
        eng.close()


def test_emo_accepts_the_callers_frame_tensor_and_a_bare_id_vector():
    """y['emo'] is [B, T] in the reference's callers (frame 0 is read, scripts_beat/model/RAG.py:125); the ABI takes that shape for
    sampling and training alike.  A bare [B] vector is broadcast by the Python layer; a strided CUDA view takes the device path."""
    import torch
    from livelyspeaker_amd import _lib
    cfg = synth.BEAT
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions)
    try:
        eng.load_state_dict(synth.make_state_dict(cfg))
        from oracle import rag_oracle as orc
        eng.set_schedule(orc.Schedule(3, ""))
        y = synth.make_cond(cfg, 4)
        tape = synth.NoiseTape(cfg, 4, 3)
        kw = dict(sampler=_lib.LS_SAMPLER_DDPM, x_init=tape.x_init, eps_tape=tape.eps, noise_tape=tape.noise)
        eng.prepare(y)
        a = eng.sample(**kw)
        eng.prepare(dict(y, emo=y["emo"][:, 0].copy()))                                   # bare [B]
        assert np.array_equal(a, eng.sample(**kw))
        wide = torch.from_numpy(np.repeat(y["emo"], 2, axis=1)).cuda()[:, ::2]            # non-contiguous CUDA view of the same ids
        yc = {k: torch.from_numpy(v).cuda() for k, v in y.items()}
        yc["emo"] = wide
        eng.prepare(yc)
        assert np.array_equal(a, eng.sample(**kw))
        for width in (1, 50):                                                             # [B, 1] and a padded [B, 50]: column 0 is all RAG.py:125 reads
            eng.prepare(dict(y, emo=np.repeat(y["emo"][:, :1], width, axis=1)))
            assert np.array_equal(a, eng.sample(**kw)), width
        with pytest.raises(_lib.EngineError):
            eng.prepare(dict(y, emo=y["emo"][:, :, None]))
        y2 = dict(y, emo=(y["emo"] + 1) % 8)
        eng.prepare(y2)
        assert not np.array_equal(a, eng.sample(**kw))                                    # the token really depends on it
    finally:
        eng.close()
