"""The one-pass-per-workgroup step kernel (ls_pass_kernel.h, ls_set_path 4 / "pass"): a workgroup of four waves per (sample, CFG pass),
two independent workgroups per CU, the passes combined by whichever finishes later.  Pinned to the REFERENCE's fixtures like the fused
kernel: G1 forwards, G2 single steps, G3 config-1 loop with dumps, G4 ddim100 (skip 80 + init_image, and full), G5 1000 steps, G11
guidance scale 1, BEAT G12 / G13; plus what is specific to it: run-to-run determinism under load (a stale hand-off read would show),
grids beyond the chip's residency, the single-pass form, the tail of a fused main part."""
import os
import threading

import numpy as np
import pytest

from conftest import GOLDEN, max_abs
from livelyspeaker_amd import synth
from test_gpu_coop import _engine as _engine_any, _g1_inputs, _loop

pytestmark = [pytest.mark.gpu, pytest.mark.engine_path_auto]
TOL_FWD, TOL_LOOP = 2e-4, 3e-4


def _engine(ds, path="pass"):
    return _engine_any(ds, path)


# "pass": 8-wave workgroups at these batch sizes (the grid fits the chip once); "pass4" (ls_set_path 5): the 4-wave / two-per-CU form,
# which `auto` and "pass" reach only beyond one workgroup per CU -- pinned to the same reference fixtures here
FORMS = ["pass", "pass4"]


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("ds", ["ted", "beat"])
def test_reference_fixtures_on_the_pass_kernel(ds, form, golden):
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg, eng = _engine(ds, form)
    g = golden[ds]
    try:
        x, eps, noise = _g1_inputs(cfg)
        eng.prepare(synth.make_cond(cfg, 4))
        for t in (0, 5, 500, 999):
            oc, ou, _ = eng.forward(x, np.full(4, t, np.int64), eps[0], eps[1])
            assert max_abs(oc, g[f"G1_t{t}_c"]) < TOL_FWD and max_abs(ou, g[f"G1_t{t}_u"]) < TOL_FWD, t
        for name, resp, steps in (("p", "", (0, 7, 999)), ("ddim", "ddim100", (0, 50, 99))):
            eng.set_schedule(orc.Schedule(1000, resp))
            for t in steps:
                s, x0 = eng.step(_lib.LS_SAMPLER_DDPM if name == "p" else _lib.LS_SAMPLER_DDIM, t, x, eps[0], eps[1], noise)
                assert max_abs(s, g[f"G2_{name}_t{t}_sample"]) < TOL_FWD and max_abs(x0, g[f"G2_{name}_t{t}_x0"]) < TOL_FWD, (name, t)
        out, dumps = _loop(eng, cfg, 50, "", False, 0, False, dump=[0, 25, 49])
        assert eng.timing()["step_path"] == 3
        d3 = max_abs(out, g["G3_ddpm50_final"])
        if ds == "ted":
            for k, dmp in zip((0, 25, 49), dumps):
                assert max_abs(dmp, g[f"G3_ddpm50_dump_x0_step{k}"]) < TOL_LOOP, k
        assert np.array_equal(out, _loop(eng, cfg, 50, "", False, 0, False, use_graph=False))      # hipGraph replay == plain launches
        assert np.array_equal(out, _loop(eng, cfg, 50, "", False, 0, False))
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
def test_round2_fixtures_scale1_and_beat_loops_on_the_pass_kernel(ds, form):
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


@pytest.mark.parametrize("ds,B", [("ted", 9), ("ted", 130), ("ted", 300), ("beat", 70), ("beat", 257)])
def test_pass_kernel_agrees_with_the_fused_kernel_at_any_grid_size(ds, B):
    """2 B workgroups against 512 resident slots: from a fraction of the chip to more than one residency round (the later workgroups
    start as slots free up; nobody waits for a partner, so any dispatch order is fine)."""
    cfg = synth.CONFIGS[ds]
    outs = {}
    for path in ("fused", "pass"):
        _, eng = _engine(ds, path)
        try:
            outs[path] = _loop(eng, cfg, 8, "", False, 0, False, B=B)
            if path == "pass":
                assert eng.timing()["step_path"] == 3
            outs[path + "1"] = _loop(eng, cfg, 8, "", False, 0, False, B=B, scale=1.0)
            assert eng.timing()["single_pass"] == 1
        finally:
            eng.close()
    d, d1 = max_abs(outs["fused"], outs["pass"]), max_abs(outs["fused1"], outs["pass1"])
    print(f"{ds} B = {B}: fused vs one-pass-per-workgroup, 8 steps: {d:.3e}; single pass {d1:.3e}")
    assert 0 < d < 5e-5 and 0 < d1 < 5e-5


def test_pass_kernel_determinism_under_load_and_next_to_another_handle():
    """A stale read of the other pass's output would almost surely differ from run to run.  Replays of a 25-step loop at B = 300 (600
    workgroups: more than the chip holds) must be bitwise identical -- alone, and while a second handle keeps the chip busy."""
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg, eng = _engine("ted")
    _, other = _engine("ted", "fused")
    try:
        eng.set_schedule(orc.Schedule(25, ""))
        eng.prepare(synth.make_cond(cfg, 300))
        other.set_schedule(orc.Schedule(100, ""))
        other.prepare(synth.make_cond(cfg, 200))
        ref = eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=11)
        for _ in range(8):
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
            for _ in range(8):
                assert np.array_equal(ref, eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=11))
        finally:
            stop.set()
            th.join()
        assert not errs, errs
    finally:
        eng.close()
        other.close()


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("ds", ["ted", "beat"])
def test_bf16x3_on_the_pass_kernel_meets_the_parity_contract(ds, form, golden):
    """The opt-in split-precision arithmetic (three bf16 MFMAs per product) inside the one-pass-per-workgroup kernel, both forms:
    contract 1e-3 against the reference's fixtures (G1 forwards, G3 / G4 / G5 loops, G11 scale 1 single-pass)."""
    from livelyspeaker_amd import _lib
    from oracle import rag_oracle as orc
    cfg, eng = _engine(ds, form)
    g = golden[ds]
    g2 = np.load(os.path.join(GOLDEN, f"{ds}_golden_r2.npz"))
    try:
        eng.set_precision("bf16x3")
        x, eps, noise = _g1_inputs(cfg)
        eng.prepare(synth.make_cond(cfg, 4))
        worst = 0.0
        for t in (0, 500, 999):
            oc, ou, _ = eng.forward(x, np.full(4, t, np.int64), eps[0], eps[1])
            worst = max(worst, max_abs(oc, g[f"G1_t{t}_c"]), max_abs(ou, g[f"G1_t{t}_u"]))
        d3 = max_abs(_loop(eng, cfg, 50, "", False, 0, False), g["G3_ddpm50_final"])
        assert eng.timing()["step_path"] == 3
        d4 = max_abs(_loop(eng, cfg, 1000, "ddim100", True, 80, True), g["G4_ddim100_skip80_final"])
        d11 = max_abs(_loop(eng, cfg, 50, "", False, 0, False, B=5, scale=1.0), g2["G11_scale1_ddpm50_B5_final"])
        assert eng.timing()["single_pass"] == 1 and eng.timing()["step_path"] == 3
        print(f"{ds} bf16x3 [{form}]: forward {worst:.3e}  G3 {d3:.3e}  G4 skip80 {d4:.3e}  G11 {d11:.3e}")
        assert max(worst, d3, d4, d11) < 1e-3
        if ds == "ted":
            d5 = max_abs(_loop(eng, cfg, 1000, "", False, 0, False), g["G5_ddpm1000_final"])
            print(f"      G5 1000 steps {d5:.3e}")
            assert d5 < 1e-3
    finally:
        eng.close()


def test_replayed_graph_through_the_torch_mirror_stays_correct_on_the_pass_kernel():
    """Regression: the arrival tickets used to be zeroed by a memset node at the head of the captured loop; REPLAYED under the torch mirror
    (device-resident outputs) that node left garbage in the ticket words and every second sample's passes were combined in the wrong
    order from the second call on.  Four calls through p_sample_loop (same Philox key, graph replay from the second) must all agree with
    the fused kernel's."""
    import torch
    from types import SimpleNamespace
    from livelyspeaker_amd.cfg_sampler import ClassifierFreeSampleModel
    from livelyspeaker_amd.model_util import create_model_and_diffusion
    cfg = synth.TED
    dev = torch.device("cuda", 0)
    B, steps = 128, 40
    args = SimpleNamespace(mdm_condm='text', latent_dim=512, ff_size=1024, layers=8, cond_mask_prob=0.1, arch='trans_enc', emb_trans_dec=False,
                           dataset='humanml', lang_model=None, mlpact='silu', diffusion_steps=steps, noise_schedule='cosine', sigma_small=True,
                           lambda_vel=1.0, lambda_rcxyz=0.0, lambda_fc=0.0, njoints=cfg.njoints)
    outs = {}
    for path in ("fused", "pass"):
        model, diffusion = create_model_and_diffusion(args, "", dataset="ted")
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()}, strict=False)
        model.to(dev)
        model.eval()
        model.step_path = path
        cfgm = ClassifierFreeSampleModel(model)
        diffusion.noise_source, diffusion.use_graph, diffusion.sample_offset = "philox", True, 0
        y = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_cond(cfg, B, scale=1.5, seed=3).items()}
        res = []
        for i in range(4):
            diffusion.philox_seed = 12345
            o = diffusion.p_sample_loop(cfgm, (B, cfg.njoints, cfg.nfeats, cfg.nframes), clip_denoised=False, model_kwargs={"y": y},
                                        skip_timesteps=0, init_image=None, progress=False, dump_steps=None, noise=None, const_noise=False)
            torch.cuda.synchronize()
            res.append(o.clone())
            if i:
                assert model.engine().timing()["graph_replayed"] == 1
        assert model.engine().timing()["step_path"] == (3 if path == "pass" else 0)
        outs[path] = res
        model.engine().close()
    for i in range(4):
        d = float((outs["fused"][i] - outs["pass"][i]).abs().max())
        assert d < 5e-5, (i, d)
        assert torch.equal(outs["pass"][i], outs["pass"][0])
