"""CPU: host-side logic of the drop-in (schedule mirror, init order, RNG draw order, ABI surface).
No compute calls are made: there is no GPU here and the product has no CPU path."""
import ctypes
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import ROOT
from livelyspeaker_amd import _lib, gaussian_diffusion as gd, synth
from livelyspeaker_amd.cfg_sampler import ClassifierFreeSampleModel
from livelyspeaker_amd.model_util import create_model_and_diffusion, load_model_wo_clip
from livelyspeaker_amd.respace import SpacedDiffusion, space_timesteps


def mk_args(steps=1000, njoints=47):
    return SimpleNamespace(mdm_condm="text", latent_dim=512, ff_size=1024, layers=8, cond_mask_prob=0.1,
                           arch="trans_enc", emb_trans_dec=False, dataset="humanml", lang_model=None, mlpact="silu",
                           diffusion_steps=steps, noise_schedule="cosine", sigma_small=True, lambda_vel=1.0,
                           lambda_rcxyz=0.0, lambda_fc=0.0, njoints=njoints)


@pytest.mark.parametrize("steps,resp", [(1000, ""), (1000, "ddim100"), (50, "")])
def test_product_schedule_tables_bit_identical_to_reference(golden, steps, resp):
    _, diff = create_model_and_diffusion(mk_args(steps), resp)
    g = golden["ted"]
    tag = f"G0_{steps}_{resp or 'full'}"
    for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
                 "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                 "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                 "posterior_mean_coef1", "posterior_mean_coef2"):
        assert np.array_equal(getattr(diff, name), g[f"{tag}_{name}"]), name
    assert np.array_equal(np.asarray(diff.timestep_map), g[f"{tag}_timestep_map"])
    assert isinstance(diff, SpacedDiffusion) and diff.num_timesteps == len(g[f"{tag}_betas"])


def test_space_timesteps_matches_reference_semantics():
    assert space_timesteps(1000, "ddim100") == set(range(0, 1000, 10))
    assert space_timesteps(1000, [1000]) == set(range(1000))
    assert len(space_timesteps(300, "10,15,20")) == 45
    with pytest.raises(ValueError):
        space_timesteps(1000, "ddim999")


def test_random_init_replays_reference_order(golden):
    """torch.manual_seed(s); RAG(...) must consume torch's RNG like scripts/model/RAG.py:56-77 does."""
    torch.manual_seed(5)
    model, _ = create_model_and_diffusion(mk_args(), "")
    sd = {k: v for k, v in model.state_dict().items() if not k.endswith(".pe")}
    cs = np.array([float(np.abs(v.numpy()).sum()) for _, v in sorted(sd.items())])
    assert np.array_equal(cs, golden["ted"]["G8_refinit_checksum"])
    # the reference's degenerate init is reproduced too (mlp_module.py:63-65, RAG.py:67)
    assert float(sd["backbone.mlps.0.block2.1.weight"].abs().max()) < 1e-8
    assert torch.all(sd["speaker_embedding.weight"] == 1e-6)


@pytest.mark.parametrize("ds", ["ted", "beat"])
def test_state_dict_contract(ds):
    cfg = synth.CONFIGS[ds]
    model, _ = create_model_and_diffusion(mk_args(njoints=cfg.njoints), "", dataset=ds)
    want = synth.make_state_dict(cfg)
    have = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith(".pe")}
    assert set(have) == set(want)
    for k, v in want.items():
        assert have[k] == v.shape, k
    pe_keys = sorted(k for k in model.state_dict() if k.endswith(".pe"))
    assert pe_keys == ["backbone.embed_timestep.sequence_pos_encoder.pe", "backbone.sequence_pos_encoder.pe",
                       "sequence_pos_encoder.pe"]
    load_model_wo_clip(model, {k: torch.from_numpy(v) for k, v in want.items()})
    assert model.eval() is None                      # RAG.train() returns None in the reference (RAG.py:136-137)
    assert (model.njoints, model.nfeats) == (cfg.njoints, cfg.nfeats)
    w = ClassifierFreeSampleModel(model)
    assert (w.njoints, w.nfeats, w.cond_mode, w.translation, w.data_rep) == (cfg.njoints, cfg.nfeats, "text", True, "vec_dir")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ls_hip.h")).read()
    declared = set(re.findall(r"\b(ls_[a-z_]+)\s*\(", hdr)) - {"ls_handle"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = ctypes.CDLL(_lib.library_path())
    for name in declared:
        assert hasattr(lib, name), name
    lib.ls_abi_version.restype = ctypes.c_int
    assert lib.ls_abi_version() == 5


def test_shipped_library_has_no_debug_switches():
    """Ablation / phase-stamp code exists only in -DLS_DEBUG variants built by tools/: the shipped binary reads no LS_*
    environment variable (an env var must never be able to make the product fast and wrong), and the Python loader has
    no environment override of which binary it binds."""
    blob = open(_lib.library_path(), "rb").read()
    for name in (b"LS_ABLATE", b"LS_PROF", b"LS_LIB"):
        assert name not in blob, name
    assert b"getenv" not in blob
    src = open(os.path.join(ROOT, "livelyspeaker_amd", "_lib.py")).read()
    assert "os.environ" not in src
    for f in ("ls_api.cpp", "ls_sag_api.cpp", "ls_train_api.cpp"):
        text = open(os.path.join(ROOT, "livelyspeaker_amd", "csrc", f)).read()
        for m in re.finditer(r"getenv", text):
            before = text[:m.start()]
            assert before.rfind("#ifdef LS_DEBUG") > before.rfind("#endif"), f"{f}: getenv outside an LS_DEBUG block"


def test_abi_shard_range_matches_the_python_shard_layer():
    from livelyspeaker_amd import shard
    lib = ctypes.CDLL(_lib.library_path())
    lib.ls_shard_range.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    f, c = ctypes.c_int64(), ctypes.c_int64()
    for total, world in ((4096, 8), (513, 8), (5, 8), (0, 3), (7, 1)):
        seen = 0
        for r in range(world):
            assert lib.ls_shard_range(total, world, r, ctypes.byref(f), ctypes.byref(c)) == 0
            assert (f.value, c.value) == shard.shard_range(total, world, r) and f.value == seen
            seen += c.value
        assert seen == total
    assert lib.ls_shard_range(8, 2, 2, ctypes.byref(f), ctypes.byref(c)) < 0 and lib.ls_shard_range(8, 0, 0, ctypes.byref(f), ctypes.byref(c)) < 0


def test_abi_struct_sizes_match_header_layout(tmp_path):
    """The ctypes mirrors against what a C compiler makes of include/ls_hip.h itself: size of every struct and the offset of its
    last field (gcc is in the image; the header is plain C)."""
    import subprocess
    pairs = [("ls_config", _lib.LsConfig), ("ls_schedule", _lib.LsSchedule), ("ls_cond", _lib.LsCond), ("ls_sample_args", _lib.LsSampleArgs),
             ("ls_forward_args", _lib.LsForwardArgs), ("ls_step_args", _lib.LsStepArgs), ("ls_timing", _lib.LsTiming),
             ("ls_sag_config", _lib.LsSagConfig), ("ls_post_config", _lib.LsPostConfig), ("ls_train_config", _lib.LsTrainConfig),
             ("ls_train_batch", _lib.LsTrainBatch), ("ls_train_terms", _lib.LsTrainTerms), ("ls_eval_config", _lib.LsEvalConfig)]
    src = tmp_path / "sizes.c"
    body = "".join(f'    printf("{c} %zu %zu\\n", sizeof({c}), offsetof({c}, {py._fields_[-1][0]}));\n' for c, py in pairs)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ls_hip.h"\nint main(void) {\n' + body + "    return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = {l.split()[0]: (int(l.split()[1]), int(l.split()[2])) for l in out.splitlines()}
    for c, py in pairs:
        assert got[c] == (ctypes.sizeof(py), getattr(py, py._fields_[-1][0]).offset), (c, got[c])
    assert ctypes.sizeof(_lib.LsSampleArgs) == 40 + 2 * 8 + 4 * 8 + 2 * 8 + 8 + 8 + 3 * 8 + 8
    assert ctypes.sizeof(_lib.LsStepArgs) == 24 + 6 * 8 + 8 + 8 + 3 * 8


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly_not_silently():
    model, diff = create_model_and_diffusion(mk_args(), "")
    cfgm = ClassifierFreeSampleModel(model)
    cfg = synth.TED
    y = {k: torch.from_numpy(v) for k, v in synth.make_cond(cfg, 2).items()}
    with pytest.raises(_lib.EngineError):
        diff.p_sample_loop(cfgm, (2, 9, 3, 34), clip_denoised=False, model_kwargs={"y": y})
    lib = _lib.load_library()
    bad = _lib.LsConfig(9, 3, 34, 1, 4, 256, 8, 36267, 1400, 0, 0, 0)       # latent_dim 256 is not built
    h = ctypes.c_void_p()
    assert lib.ls_create(ctypes.byref(bad), ctypes.byref(h)) == -5
    assert b"latent_dim" in lib.ls_last_error(None)
    bad = _lib.LsConfig(9, 3, 34, 1, 4, 512, 8, 30000, 1400, 0, 0, 0)       # audio length that is not 34 frames
    assert lib.ls_create(ctypes.byref(bad), ctypes.byref(h)) == -1


class _FakeEngine:
    """Records what the sampler hands to the engine; stands in for libls_hip.so in RNG-order tests."""
    J, F, T, D, batch, n_steps = 9, 3, 34, 512, 2, 0

    def set_schedule(self, sched):
        self.n_steps = sched.num_timesteps

    def sample(self, **kw):
        self.kw = kw
        return np.zeros((self.batch, self.J, self.F, self.T), np.float32)


def test_sampler_draws_noise_in_the_reference_order(monkeypatch):
    """randn(B,J,F,T) once, then per step randn(B,1,512) x2 and randn(B,J,F,T)."""
    model, diff = create_model_and_diffusion(mk_args(steps=5), "")
    cfgm = ClassifierFreeSampleModel(model)
    fake = _FakeEngine()
    monkeypatch.setattr(type(model), "_engine_prepared", lambda self, y: fake)
    y = {"dummy": torch.zeros(2)}
    torch.manual_seed(99)
    diff.p_sample_loop(cfgm, (2, 9, 3, 34), clip_denoised=False, model_kwargs={"y": y}, device="cpu", skip_timesteps=2)
    torch.manual_seed(99)
    x_init = torch.randn(2, 9, 3, 34)
    assert torch.equal(torch.as_tensor(fake.kw["x_init"]), x_init)
    assert tuple(fake.kw["eps_tape"].shape) == (3, 2, 2, 512) and tuple(fake.kw["noise_tape"].shape) == (3, 2, 9, 3, 34)
    later = torch.empty(34, 2, 9, 3).permute(1, 2, 3, 0)      # layout of x after the first step (RAG.py:209-210)
    for k in range(3):
        assert torch.equal(fake.kw["eps_tape"][k, 0], torch.randn(2, 1, 512)[:, 0])
        assert torch.equal(fake.kw["eps_tape"][k, 1], torch.randn(2, 1, 512)[:, 0])
        want = torch.randn(2, 9, 3, 34) if k == 0 else torch.randn_like(later)
        assert torch.equal(fake.kw["noise_tape"][k], want)
    assert fake.kw["skip_timesteps"] == 2 and fake.kw["clip_denoised"] is False
    # explicit noise= replaces the initial draw (gaussian_diffusion.py:701-704)
    given = torch.full((2, 9, 3, 34), 0.5)
    diff.ddim_sample_loop(cfgm, (2, 9, 3, 34), noise=given, model_kwargs={"y": y}, device="cpu")
    assert torch.equal(fake.kw["x_init"], given) and fake.kw["sampler"] == _lib.LS_SAMPLER_DDIM


def test_sampler_error_behaviour(monkeypatch):
    model, diff = create_model_and_diffusion(mk_args(steps=5), "")
    cfgm = ClassifierFreeSampleModel(model)
    fake = _FakeEngine()
    monkeypatch.setattr(type(model), "_engine_prepared", lambda self, y: fake)
    y = {"y": {"d": torch.zeros(2)}}
    with pytest.raises(NotImplementedError):        # gaussian_diffusion.py:919-920
        diff.ddim_sample_loop(cfgm, (2, 9, 3, 34), model_kwargs=y, dump_steps=[0])
    with pytest.raises(NotImplementedError):
        diff.p_sample_loop(cfgm, (2, 9, 3, 34), model_kwargs=y, cond_fn=lambda *a: None)
    with pytest.raises(TypeError):
        diff.p_sample_loop(model, (2, 9, 3, 34), model_kwargs=y)
    with pytest.raises(ValueError):
        diff.p_sample_loop(cfgm, (3, 9, 3, 34), model_kwargs=y, device="cpu")
    model.cond_mask_prob = 0.0
    assert cfgm(torch.zeros(2, 9, 3, 34), torch.zeros(2, dtype=torch.long), y=y["y"]) is None   # cfg_sampler.py:24-31


def test_q_sample_matches_tables():
    _, diff = create_model_and_diffusion(mk_args(), "ddim100")
    x0, nz = torch.randn(3, 9, 3, 34), torch.randn(3, 9, 3, 34)
    t = torch.tensor([0, 50, 99])
    got = diff.q_sample(x0, t, nz)
    a = torch.tensor(diff.sqrt_alphas_cumprod[[0, 50, 99]]).float().view(3, 1, 1, 1)
    b = torch.tensor(diff.sqrt_one_minus_alphas_cumprod[[0, 50, 99]]).float().view(3, 1, 1, 1)
    assert torch.equal(got, a * x0 + b * nz)


def test_train_abi_struct_sizes_match_header_layout():
    import ctypes as C
    from livelyspeaker_amd import _lib
    # ls_train_config = ls_config (12 x int32) + 2 floats + 2 int32; ls_train_batch = 2 x int32 + 9 pointers; terms = 8 floats
    assert C.sizeof(_lib.LsTrainConfig) == 12 * 4 + 16
    assert C.sizeof(_lib.LsTrainBatch) == 8 + 9 * 8
    assert C.sizeof(_lib.LsTrainTerms) == 32
    assert _lib.LsTrainBatch.t.offset == 16 and _lib.LsTrainBatch.emo.offset == 72


def test_uniform_sampler_replays_numpy_choice_stream():
    import numpy as np
    from types import SimpleNamespace
    from livelyspeaker_amd.resample import create_named_schedule_sampler
    import pytest
    s = create_named_schedule_sampler("uniform", SimpleNamespace(num_timesteps=1000))
    np.random.seed(5)
    t, w = s.sample(16, "cpu")
    np.random.seed(5)
    want = np.random.choice(1000, size=(16,), p=np.ones(1000) / 1000)
    assert np.array_equal(t.numpy(), want) and float(w.min()) == 1.0 == float(w.max())
    with pytest.raises(NotImplementedError):
        create_named_schedule_sampler("loss-second-moment", SimpleNamespace(num_timesteps=10))


def test_parse_resume_step_from_filename():
    from livelyspeaker_amd.train_loop import parse_resume_step_from_filename as f
    assert f("save/x/model000012345.pt") == 12345 and f("nothing.pt") == 0 and f("modelabc.pt") == 0


def test_bench_stdout_carries_only_the_result_line():
    """bench.py promises ONE JSON line on stdout; libraries under it (RCCL's version banner) write to fd 1 too, so the
    run points fd 1 at stderr and keeps a private duplicate for the result."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent(f'''
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import bench
        out = bench._reserve_stdout()
        os.write(1, b"banner written to fd 1 by a library\\n")
        print("python-level noise")
        out.write('{{"ok": 1}}\\n'); out.flush()
    ''')
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"ok": 1}\n'
    assert "banner written to fd 1" in r.stderr and "python-level noise" in r.stderr


def test_build_staleness_is_content_based(tmp_path, monkeypatch):
    """The library is rebuilt when its sources changed, judged by the content hash the build leaves beside it (file times do not
    survive every copy of the tree); without the stamp the check falls back to modification times."""
    from livelyspeaker_amd import build as b
    if not os.path.exists(b.LIB):
        pytest.skip("library not built")
    stamp = tmp_path / "libls_hip.so.srchash"
    monkeypatch.setattr(b, "STAMP", str(stamp))
    stamp.write_text(b.source_hash() + "\n")
    assert not b.is_stale()
    stamp.write_text("0" * 64 + "\n")
    assert b.is_stale()
    stamp.unlink()
    assert b.is_stale() == any(os.path.getmtime(d) > os.path.getmtime(b.LIB) for d in [os.path.join(b.CSRC, s) for s in b.SOURCES] + b.HEADERS)


def test_checkpoint_files_interchange_with_the_reference(tmp_path):
    """model%09d.pt must pass the reference's `load_model_wo_clip` (no unexpected keys, only clip_model.* may be missing) and
    opt%09d.pt must load into a `torch.optim.AdamW` over the reference model's parameters; and a file written by the reference's
    optimizer must load back into the engine's layout.  Runs the reference's own loader when /root/reference is present (build
    container); the layout checks against the mirror model run everywhere."""
    import subprocess
    import sys
    from livelyspeaker_amd import synth, train_loop as tl
    from livelyspeaker_amd.model_util import create_model_and_diffusion
    cfg = synth.TED
    model, _ = create_model_and_diffusion(_mk_args_local(cfg), "")
    trained = {k: v + 1.0 for k, v in synth.make_state_dict(cfg).items()}
    keys = tl._param_keys(model)
    g = np.random.Generator(np.random.PCG64(11))
    opt = {"step": 7, "exp_avg": {k: g.standard_normal(dict(model.named_parameters())[k].numel()).astype(np.float32) for k in keys},
           "exp_avg_sq": {k: np.abs(g.standard_normal(dict(model.named_parameters())[k].numel())).astype(np.float32) for k in keys}}
    msd = tl.pack_model_checkpoint(model, trained)
    assert list(msd) == [k for k in model.state_dict() if not k.startswith("clip_model.")]
    assert sum(k.endswith(".pe") for k in msd) == 3                      # the three PositionalEncoding buffers travel too
    assert torch.equal(msd["input_mapping.weight"], torch.from_numpy(trained["input_mapping.weight"]))
    ost = tl.pack_optimizer_state(model, opt, lr=3e-5, weight_decay=0.01)
    ref_opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    ref_opt.load_state_dict(ost)                                          # torch's own loader accepts the layout
    assert ref_opt.param_groups[0]["lr"] == 3e-5 and ref_opt.param_groups[0]["weight_decay"] == 0.01
    back, lr = tl.unpack_optimizer_state(model, ref_opt.state_dict())     # ... and what torch writes loads back
    assert lr == 3e-5 and back["step"] == 7
    for k in keys:
        assert np.array_equal(np.asarray(back["exp_avg"][k]).ravel(), opt["exp_avg"][k])
        assert np.array_equal(np.asarray(back["exp_avg_sq"][k]).ravel(), opt["exp_avg_sq"][k])
    old, lr_old = tl.unpack_optimizer_state(model, {"step": 3, "exp_avg": {"a": 1}, "exp_avg_sq": {"a": 2}})   # round-1 layout
    assert old["step"] == 3 and lr_old is None
    if not os.path.isdir("/root/reference/scripts"):
        return
    torch.save(msd, tmp_path / "model000000007.pt")
    torch.save(ost, tmp_path / "opt000000007.pt")
    code = f'''
import sys, types, torch
sys.dont_write_bytecode = True
sys.modules["clip"] = types.ModuleType("clip")
sys.path.insert(0, "/root/reference/scripts")
from types import SimpleNamespace
from mdm_utils.model_util import create_model_and_diffusion, load_model_wo_clip
args = SimpleNamespace(mdm_condm="text", latent_dim=512, ff_size=1024, layers=8, cond_mask_prob=0.1, arch="trans_enc", emb_trans_dec=False,
    dataset="humanml", lang_model=None, mlpact="silu", diffusion_steps=1000, noise_schedule="cosine", sigma_small=True, lambda_vel=1.0,
    lambda_rcxyz=0.0, lambda_fc=0.0)
model, _ = create_model_and_diffusion(args, "")
load_model_wo_clip(model, torch.load(r"{tmp_path}/model000000007.pt", map_location="cpu"))        # asserts on unexpected / missing keys
opt = torch.optim.AdamW(model.parameters_wo_clip() if hasattr(model, "parameters_wo_clip") else model.parameters(), lr=1e-4, weight_decay=0.0)
opt.load_state_dict(torch.load(r"{tmp_path}/opt000000007.pt", map_location="cpu"))
names = [k for k, _ in model.named_parameters()]
assert float(opt.state[model.get_parameter(names[0])]["step"]) == 7.0
assert tuple(opt.state[model.get_parameter("input_mapping.weight")]["exp_avg"].shape) == (512, 311)
torch.save(opt.state_dict(), r"{tmp_path}/ref_opt.pt")
print("REFERENCE-LOADER-OK", len(names))
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "REFERENCE-LOADER-OK" in r.stdout, r.stderr[-2000:]
    assert int(r.stdout.split("REFERENCE-LOADER-OK")[1].split()[0]) == len(keys)       # same parameter list, same order
    back2, _ = tl.unpack_optimizer_state(model, torch.load(tmp_path / "ref_opt.pt", map_location="cpu"))
    assert np.array_equal(np.asarray(back2["exp_avg"]["speaker_mu.bias"]).ravel(), opt["exp_avg"]["speaker_mu.bias"])


def _mk_args_local(cfg):
    from types import SimpleNamespace
    return SimpleNamespace(mdm_condm="text", latent_dim=512, ff_size=1024, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                           emb_trans_dec=False, dataset="humanml", lang_model=None, mlpact="silu", diffusion_steps=1000,
                           noise_schedule="cosine", sigma_small=True, lambda_vel=1.0, lambda_rcxyz=0.0, lambda_fc=0.0, njoints=cfg.njoints)


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher must re-execute itself under torch.distributed.run (one rank per GPU), not exit
    asking for one.  No GPU here, so each of the two ranks stops at its device check -- which is the evidence that two ranks ran."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""                 # nothing but the JSON line may ever reach stdout
    assert "must be launched with" not in r.stderr
    # (the launcher tears the other rank down as soon as one fails, so either rank's message may be the only one)
    assert "rank 0: local rank 0 but only 0 GPU(s) visible" in r.stderr or "rank 1: local rank 1 but only 0 GPU(s) visible" in r.stderr
    assert "torch/distributed" in r.stderr                              # the elastic launcher's failure report: it was the launcher that ran them
    # a launcher / --gpus mismatch is named, not silently accepted
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=600,
                        cwd=root, env=dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"))
    assert r2.returncode != 0 and "--gpus 1 but the launcher started 2 rank(s)" in r2.stderr


def test_identical_seeds_tapes_in_segments_are_the_one_piece_tapes(monkeypatch):
    """The chunked "identical seeds" mode (GaussianDiffusion._loop, noise_source='torch_cpu') must hand the engine exactly the tapes of
    the one-piece mode, cut into consecutive (first step, count) segments -- same draws from torch's CPU generator in the same order
    (randn(*shape); per step randn(B,1,512) x 2 and randn_like(x) with x's strides).  The engine is a recording stand-in here."""
    import torch as th
    from livelyspeaker_amd import gaussian_diffusion as gd
    from livelyspeaker_amd.model_util import create_gaussian_diffusion
    diff = create_gaussian_diffusion(mk_args(steps=23), "")

    class FakeEngine:
        batch, J, F, T, D = 3, 9, 3, 34, 512
        def __init__(self):
            self.calls = []
        def sample(self, **kw):
            self.calls.append({k: (v.clone() if th.is_tensor(v) else v) for k, v in kw.items()})
            last = kw.get("segment") is None or kw["segment"][0] + kw["segment"][1] == 23
            return th.zeros(3, 9, 3, 34).numpy() if last else None

    def run(segment_bytes):
        eng = FakeEngine()
        monkeypatch.setattr(diff, "_engine_for", lambda *a, **k: eng)
        diff.tape_segment_bytes = segment_bytes
        th.manual_seed(77)
        diff.p_sample_loop(object(), (3, 9, 3, 34), clip_denoised=False, model_kwargs={"y": {}}, device="cpu")
        return eng.calls

    monkeypatch.setattr(th.cuda, "is_available", lambda: True)                       # the chunked path is for the GPU build only
    monkeypatch.setattr(gd.GaussianDiffusion, "_tape_ring", lambda self, K, B, D, shape:
                        [(th.empty(K, 2, B, D), th.empty((K,) + tuple(shape))) for _ in range(2)])    # (no page-locked memory here)
    one = run(1 << 40)
    assert len(one) == 1 and one[0].get("segment") is None and diff.last_tape_segments == 1
    per_step = (2 * 3 * 512 + 3 * 9 * 3 * 34) * 4
    seg = run(2 * 5 * per_step)                                                       # room for two 5-step segments
    assert [c["segment"] for c in seg] == [(0, 5), (5, 5), (10, 5), (15, 5), (20, 3)] and diff.last_tape_segments == 5
    assert th.equal(th.cat([c["eps_tape"] for c in seg]), one[0]["eps_tape"])
    assert th.equal(th.cat([c["noise_tape"] for c in seg]), one[0]["noise_tape"])
    assert all(th.equal(c["x_init"], one[0]["x_init"]) for c in seg)
    assert all("use_graph" not in c for c in seg) and "use_graph" in one[0]


def test_sampler_outputs_carry_the_reference_strides():
    """pred_xstart / sample of the reference are permuted views (OutputProcess, RAG.py:209-210): memory order [T][B][J][F].  randn_like
    of the next step consumes the generator in that order, so the mirror returns the same strides."""
    import torch as th
    from livelyspeaker_amd.gaussian_diffusion import _ref_strides
    x = th.arange(4 * 9 * 3 * 34, dtype=th.float32).reshape(4, 9, 3, 34)
    r = _ref_strides(x)
    assert th.equal(r, x) and not r.is_contiguous()
    want = th.empty(34, 4, 9, 3).permute(1, 2, 3, 0)
    assert r.stride() == want.stride()
    th.manual_seed(1); a = th.randn_like(r)
    th.manual_seed(1); b = th.randn_like(want)
    assert th.equal(a, b)


def test_committed_traffic_profile_was_taken_on_the_shipped_step_kernel():
    """profiles/k_step_traffic.json (rocprofv3 PMC passes, the figure bench.py quotes as roofline.traffic_from_committed_profile) is
    keyed to the SHA-256 of ls_step_kernel.h it was measured on: a kernel edit without re-profiling (tools/traffic_measure.sh) fails here
    instead of silently leaving a stale profile behind the bench line."""
    import hashlib
    import json
    doc = json.load(open(os.path.join(ROOT, "profiles", "k_step_traffic.json")))
    now = hashlib.sha256(open(os.path.join(ROOT, "livelyspeaker_amd", "csrc", "ls_step_kernel.h"), "rb").read()).hexdigest()
    stale = [k for k, e in doc["entries"].items() if e["kernel_source_sha256"] != now]
    assert not stale, f"re-take profiles/k_step_traffic.json (tools/traffic_measure.sh on the GPU box): stale entries {stale}"


def test_progressive_generators_refuse_bad_arguments_when_requested_not_at_first_next():
    """A generator body runs at the first next(): the argument checks of p_sample_loop_progressive / ddim_sample_loop_progressive are made
    by the call itself (unbuilt hooks, a shape that is not this model's, missing conditioning); the draws stay inside the generator."""
    model, diff = create_model_and_diffusion(mk_args(steps=5), "")
    cfgm = ClassifierFreeSampleModel(model)
    y = {"dummy": torch.zeros(2)}
    for fn in (diff.p_sample_loop_progressive, diff.ddim_sample_loop_progressive):
        with pytest.raises(NotImplementedError):
            fn(cfgm, (2, 9, 3, 34), cond_fn=lambda *a: None, model_kwargs={"y": y})
        with pytest.raises(ValueError, match="njoints"):
            fn(cfgm, (2, 9, 3, 30), model_kwargs={"y": y})
        with pytest.raises(ValueError, match="shape must be"):
            fn(cfgm, (2, 27, 34), model_kwargs={"y": y})
        with pytest.raises(ValueError, match="model_kwargs"):
            fn(cfgm, (2, 9, 3, 34), model_kwargs=None)
        state = torch.get_rng_state()
        gen = fn(cfgm, [2, 9, 3, 34], model_kwargs={"y": y}, device="cpu")          # accepted: nothing has run or been drawn yet
        assert torch.equal(state, torch.get_rng_state()) and hasattr(gen, "__next__")


def test_step_plans_cover_the_batch_and_follow_the_chip():
    """ls_plan_query (no GPU): the pieces `auto` splits a batch into are contiguous and cover it, a fused piece comes first and holds whole
    rounds of the chip, bf16x3 plans use the two families that have that mode, and the round / unit sizes follow the CU count."""
    from livelyspeaker_amd import _lib as L
    for ds in ("ted", "beat"):
        for sp in (False, True):
            for prec in ("fp32", "bf16x3"):
                for n_cus in (256, 64, 304):
                    for B in list(range(1, 70, 7)) + [96, 128, 129, 160, 255, 256, 257, 300, 352, 384, 416, 511, 512, 513, 1000, 4096]:
                        segs, ms = L.plan_query(B, ds, sp, prec, n_cus)
                        assert 1 <= len(segs) <= 3 and ms > 0, (ds, sp, prec, n_cus, B, segs)
                        nxt = 0
                        for i, (path, first, n) in enumerate(segs):
                            assert first == nxt and n > 0 and path in (0, 1, 2, 3), (B, segs)
                            nxt += n
                            if path == 0:
                                assert i == 0                                            # the fused kernel indexes clips from 0
                                if len(segs) > 1:
                                    assert n % (n_cus * (2 if sp else 1)) == 0, (B, segs)   # whole rounds in front of a remainder
                            if prec == "bf16x3":
                                assert path in (0, 3), (B, segs)
                        assert nxt == B, (B, segs)
    # the measured table's landmarks (TED, fp32, CFG, 256 CUs)
    want = {4: [2], 32: [2], 64: [2], 128: [3], 160: [3, 2], 256: [0], 300: [0, 2], 384: [0, 3], 416: [0, 3, 2], 512: [0], 4096: [0]}
    for B, fam in want.items():
        assert [p for p, _, _ in L.plan_query(B)[0]] == fam, (B, L.plan_query(B))
    # the sample-split kernel's slicing: 8 slices of 64 channels up to 30 clips under CFG, 4 slices of 128 channels (one workgroup per CU, the
    # slices of a group on one XCD) at 32, two slices of 256 channels (64 clips per launch) from there to where the one-pass-per-workgroup
    # kernel takes over
    for ds in ("ted", "beat"):
        assert [L.plan_coop_slices(2 * b, ds) for b in (1, 4, 16, 28, 32, 40, 48, 64)] == [8, 8, 8, 8, 4, 2, 2, 2], ds
        assert L.plan_coop_slices(32, ds) == 8 and L.plan_coop_slices(64, ds) == 4 and L.plan_coop_slices(80, ds) == 2     # single-pass form: groups = clips
    assert [p for p, _, _ in L.plan_query(48)[0]] == [2] and [p for p, _, _ in L.plan_query(80)[0]] == [2] and [p for p, _, _ in L.plan_query(96)[0]] == [3]      # 80 clips: 64 on two slices + 16 on eight, two launches
    # model time never falls when clips are added by whole rounds, and a batch never costs more than the next multiple of the chip
    for B in range(1, 513):
        assert L.plan_query(B)[1] <= L.plan_query(-(-B // 256) * 256)[1] + 1e-6, B
    with pytest.raises(L.EngineError):
        L.plan_query(0)
