"""Training step on the GPU, through the C-ABI (ls_train_*), against the torch-CPU oracle
(oracle/train_oracle.py, itself pinned to fixtures generated from the reference) and against those fixtures directly.

Tolerances (fp32 everywhere, only the summation order differs):
  * loss terms: 2e-5 relative;  model output: 1e-3 max-abs (the path's contract), measured ~2e-5
  * gradients: max|g_hip - g_oracle| <= 2e-4 * max|g_oracle| per parameter tensor, measured ~1e-6..2e-5
  * AdamW kernel, fed the oracle's gradients: parameters within 1e-6 (a few ulp at |p| ~ 1)
  * three conv biases feed an InstanceNorm: their true gradient is 0, both sides return rounding noise (< 1e-5), and Adam
    turns that noise into +-lr steps -- excluded from parameter comparisons (the function does not depend on them).
"""
import os

import numpy as np
import pytest

from livelyspeaker_amd import synth
from oracle import rag_oracle as orc
from oracle import train_oracle as tro

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
NULL_GRAD = tuple(f"audio_encoder.feat_extractor.{i}.bias" for i in (0, 3, 6))
B = 6


def make(dataset):
    from livelyspeaker_amd import _lib
    cfg = synth.CONFIGS[dataset]
    sd = synth.make_state_dict(cfg)
    tr = _lib.Trainer(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions)
    tr.load_state_dict(sd)
    tr.set_schedule(orc.Schedule(1000, ""))
    return cfg, sd, tr


@pytest.mark.parametrize("dataset", ["ted", "beat"])
def test_forward_backward_matches_oracle_and_fixture(dataset):
    cfg, sd, tr = make(dataset)
    gold = np.load(os.path.join(GOLD, f"train_{dataset}_golden.npz"))
    oracle = tro.TrainOracle(sd, cfg.n_prefix_tokens)
    x_start, y, noise, drop, eps = synth.make_train_batch(cfg, B, 0)
    t = gold["s0_t"]
    terms = tr.forward_backward(x_start, t, noise, y, drop, eps)
    oterms, ototal, ograds, oout = oracle.forward_backward(x_start, t, noise, y, drop, eps)
    for k in ("rot_mse", "vel_mse", "kld", "loss"):
        assert abs(terms[k] - float(gold[f"s0_{k}"])) <= 2e-5 * max(1.0, abs(float(gold[f"s0_{k}"]))), (k, terms[k])
        assert abs(terms[k] - oterms[k]) <= 2e-5 * max(1.0, abs(oterms[k])), k
    assert abs(terms["total"] - float(gold["s0_total"])) <= 2e-5 * max(1.0, abs(ototal))
    out = tr.read("out", (B, cfg.nframes, cfg.jf)).reshape(B, cfg.nframes, cfg.njoints, cfg.nfeats).transpose(0, 2, 3, 1)
    assert np.abs(out - oout).max() < 1e-3
    g = tr.grads()
    assert sorted(g) == sorted(ograds)
    # LeakyReLU' is discontinuous: an InstanceNorm output within a few ulp of 0 may get either slope depending on the last
    # bit of the channel mean.  Conv weights upstream of such an element (layer j and below) are compared in relative L2.
    import torch
    margins = tro.leaky_kink_margins(oracle.P, torch.from_numpy(y["audio_input"]))
    exposed = {f"audio_encoder.feat_extractor.{idx}.weight" for j, idx in enumerate((0, 3, 6)) if min(margins[j:]) < 2e-6}
    print(f"[{dataset}] min |normalised activation| per layer: {margins}; kink-exposed tensors: {sorted(exposed)}")
    worst = 0.0
    for k in g:
        if k in NULL_GRAD:
            assert np.abs(g[k]).max() < 1e-5, k
            continue
        if k in exposed:
            assert np.linalg.norm(g[k] - ograds[k]) <= 2e-2 * np.linalg.norm(ograds[k]), k
            continue
        rel = np.abs(g[k] - ograds[k]).max() / (np.abs(ograds[k]).max() + 1e-12)
        worst = max(worst, rel)
        assert rel < 2e-4, (k, rel)
        if f"g_{k}_full" in gold and k not in exposed:   # reference gradients stored in full for the small tensors
            ref = gold[f"g_{k}_full"]
            assert np.abs(g[k] - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-9, k
    print(f"[{dataset}] worst relative gradient error {worst:.2e}")


def test_adamw_kernel_matches_oracle_given_same_gradients():
    import torch
    cfg, sd, tr = make("ted")
    oracle = tro.TrainOracle(sd, cfg.n_prefix_tokens, lr=3e-4, weight_decay=0.02)
    g = np.random.Generator(np.random.PCG64(11))
    for step in range(3):
        grads = {k: (g.standard_normal(v.shape) * 10.0 ** g.integers(-6, 0)).astype(np.float32) for k, v in sd.items()}
        flat = np.zeros(tr.flat_size, np.float32)
        for k, (o, n) in tr.params.items():
            flat[o:o + n] = grads[k].ravel()
        tr.grad.copy_(torch.from_numpy(flat))
        tr.adamw(lr=3e-4, weight_decay=0.02)
        oracle.optimizer_step(grads)
    psd, osd = tr.state_dict(), oracle.state_dict()
    for k in psd:
        assert psd[k].shape == osd[k].shape
        assert np.abs(psd[k] - osd[k]).max() < 1e-6, k      # a few ulp of parameters ~1 after three steps


def test_five_training_steps_track_the_oracle():
    """End to end: loss trajectory and parameters.  Adam divides by sqrt(v): where |g| is at rounding-noise level the
    update direction is ill-conditioned in ANY fp32 implementation, so parameters are compared in bulk (99.9 % of the
    elements within 2e-6) and bounded everywhere by the largest possible drift 2 * lr * steps."""
    cfg, sd, tr = make("ted")
    oracle = tro.TrainOracle(sd, cfg.n_prefix_tokens)
    rng = np.random.Generator(np.random.PCG64(77))
    for step in range(5):
        x_start, y, noise, drop, eps = synth.make_train_batch(cfg, B, step)
        t = rng.integers(0, 1000, size=(B,))
        terms = tr.forward_backward(x_start, t, noise, y, drop, eps)
        oterms, ototal, ograds, _ = oracle.forward_backward(x_start, t, noise, y, drop, eps)
        assert abs(terms["total"] - ototal) <= 1e-4 * max(1.0, abs(ototal)), (step, terms["total"], ototal)
        tr.adamw()
        oracle.optimizer_step(ograds)
    psd, osd = tr.state_dict(), oracle.state_dict()
    n_all = n_bad = 0
    for k in psd:
        if k in NULL_GRAD:
            continue
        d = np.abs(psd[k] - osd[k])
        assert d.max() <= 2 * 1e-4 * 5 + 1e-7, k
        n_all += d.size
        n_bad += int((d > 2e-6).sum())
    assert n_bad <= 1e-3 * n_all, (n_bad, n_all)


def test_train_argument_errors():
    from livelyspeaker_amd import _lib
    cfg, sd, tr = make("ted")
    x_start, y, noise, drop, eps = synth.make_train_batch(cfg, 2, 0)
    with pytest.raises(_lib.EngineError):
        tr.forward_backward(x_start, np.array([0, 1000]), noise, y, drop, eps)          # t out of range
    tr2 = _lib.Trainer(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len)
    with pytest.raises(_lib.EngineError):
        tr2.forward_backward(x_start, np.array([0, 1]), noise, y, drop, eps)             # no schedule yet
    with pytest.raises(_lib.EngineError):
        tr.load_state_dict({"not.a.key": np.zeros(3, np.float32)})
    with pytest.raises(_lib.EngineError):
        tr.load_state_dict({"input_mapping.bias": np.zeros(3, np.float32)})              # wrong size


# ---------------------------------------------------------------------------------------------------------------
# Host mirror: TrainLoop drop-in (scripts/train_utils/train_loop.py) driving the engine
def _loop_fixture(tmp_path, steps_data, resume="", lr_anneal_steps=0):
    import torch
    from types import SimpleNamespace
    from livelyspeaker_amd.model_util import create_model_and_diffusion
    from livelyspeaker_amd.train_loop import TrainLoop
    cfg = synth.CONFIGS["ted"]
    margs = SimpleNamespace(mdm_condm="text", latent_dim=512, ff_size=1024, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                            emb_trans_dec=False, dataset="humanml", lang_model=None, mlpact="silu", diffusion_steps=1000,
                            noise_schedule="cosine", sigma_small=True, lambda_vel=1.0, lambda_rcxyz=0.0, lambda_fc=0.0, njoints=9)
    model, diffusion = create_model_and_diffusion(margs, "")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()}, strict=False)
    model.to("cuda:0")
    model.train()
    targs = SimpleNamespace(batch_size=B, lr=1e-4, weight_decay=0.0, lr_anneal_steps=lr_anneal_steps, log_interval=1000, save_interval=1000,
                            resume_checkpoint=resume, epochs=1, save_dir=str(tmp_path), overwrite=True, dataset="ted")
    return cfg, model, diffusion, TrainLoop(targs, None, model, diffusion, steps_data)


def _batches(cfg, n, start=0):
    import torch
    out = []
    for s in range(start, start + n):
        x_start, y, _, _, _ = synth.make_train_batch(cfg, B, s)
        cond = {"y": {k: torch.from_numpy(v) for k, v in y.items()}}
        out.append((torch.from_numpy(x_start), cond))
    return out


def test_trainloop_reproduces_oracle_trajectory_with_reference_draw_order():
    """Seeds -> the loop's own draws (np.random.choice, randn, bernoulli, randn) are replayed for the oracle in the
    reference's order; losses must agree step by step and the trained weights must land back in the model."""
    import torch
    cfg = synth.CONFIGS["ted"]
    data = _batches(cfg, 3)
    _, model, diffusion, loop = _loop_fixture("/tmp/ls_train_a", data)
    np.random.seed(21)
    torch.manual_seed(21)
    losses = []
    for motion, cond in data:
        loop.run_step(motion, cond)
        losses.append(loop.last_losses["total"])
        loop.step += 1
    # oracle with the same stream
    oracle = tro.TrainOracle(synth.make_state_dict(cfg), cfg.n_prefix_tokens)
    np.random.seed(21)
    torch.manual_seed(21)
    for i, (motion, cond) in enumerate(_batches(cfg, 3)):
        t = np.random.choice(1000, size=(B,), p=np.ones(1000) / 1000)
        noise = torch.randn(tuple(motion.shape)).numpy()
        drop = torch.bernoulli(torch.ones(B) * 0.1).numpy()
        eps = torch.randn(B, 1, 512).numpy().reshape(B, 512)
        y = {k: v.numpy() for k, v in cond["y"].items()}
        _, ototal, ograds, _ = oracle.forward_backward(motion.numpy(), t, noise, y, drop, eps)
        oracle.optimizer_step(ograds)
        assert abs(losses[i] - ototal) <= 1e-4 * max(1.0, abs(ototal)), (i, losses[i], ototal)
    loop.sync_model()
    w = model.state_dict()["output_process.poseFinal.weight"].cpu().numpy()
    assert np.abs(w - oracle.state_dict()["output_process.poseFinal.weight"]).max() < 1e-3
    assert np.abs(w - synth.make_state_dict(cfg)["output_process.poseFinal.weight"]).max() > 1e-5     # it did train


@pytest.mark.parametrize("anneal", [0, 10])
def test_trainloop_checkpoint_resume_is_bit_identical(tmp_path, anneal):
    """Also with lr annealing on: the resumed run must continue with the annealed lr of the last executed step (restored from
    the optimizer file's param_groups like opt.load_state_dict does, train_loop.py:96-104), not restart at args.lr."""
    import torch
    cfg = synth.CONFIGS["ted"]

    def run(loop, data):
        for motion, cond in data:
            loop.run_step(motion, cond)
            loop.step += 1

    np.random.seed(3); torch.manual_seed(3)
    _, _, _, full = _loop_fixture(tmp_path / "full", None, lr_anneal_steps=anneal)
    run(full, _batches(cfg, 4))
    ref = full.trainer.state_dict()

    np.random.seed(3); torch.manual_seed(3)
    _, _, _, first = _loop_fixture(tmp_path / "ck", None, lr_anneal_steps=anneal)
    run(first, _batches(cfg, 2))
    first.save()
    ck = os.path.join(str(tmp_path / "ck"), "model000000002.pt")
    assert os.path.exists(ck) and os.path.exists(os.path.join(str(tmp_path / "ck"), "opt000000002.pt"))
    rng_np, rng_t = np.random.get_state(), torch.get_rng_state()
    saved = torch.load(ck, map_location="cpu")
    assert sum(k.endswith(".pe") for k in saved) == 3                 # the reference's loader requires the buffers
    osd = torch.load(os.path.join(str(tmp_path / "ck"), "opt000000002.pt"), map_location="cpu")
    assert set(osd) == {"state", "param_groups"} and len(osd["state"]) == len(osd["param_groups"][0]["params"])
    _, _, _, second = _loop_fixture(tmp_path / "ck", None, resume=ck, lr_anneal_steps=anneal)
    assert second.resume_step == 2 and second.trainer.optimizer_state()["step"] == 2
    assert second.cur_lr == first.cur_lr
    np.random.set_state(rng_np); torch.set_rng_state(rng_t)
    run(second, _batches(cfg, 2, start=2))
    got = second.trainer.state_dict()
    for k in ref:
        assert np.array_equal(ref[k], got[k]), k


def test_trainloop_resume_without_optimizer_file_starts_at_args_lr(tmp_path):
    """train_loop.py:60-66, 96-104: with no opt%09d.pt beside the checkpoint the reference keeps the AdamW it has just built at
    args.lr, so the FIRST resumed step runs un-annealed and _anneal_lr takes over after it."""
    import torch
    cfg = synth.CONFIGS["ted"]
    np.random.seed(4); torch.manual_seed(4)
    _, _, _, first = _loop_fixture(tmp_path / "ck", None, lr_anneal_steps=10)
    for motion, cond in _batches(cfg, 2):
        first.run_step(motion, cond)
        first.step += 1
    first.save()
    os.remove(os.path.join(str(tmp_path / "ck"), "opt000000002.pt"))
    ck = os.path.join(str(tmp_path / "ck"), "model000000002.pt")
    _, _, _, second = _loop_fixture(tmp_path / "ck", None, resume=ck, lr_anneal_steps=10)
    assert second.resume_step == 2 and second.cur_lr == 1e-4                      # fresh optimizer: args.lr, not the annealed 0.9e-4
    assert second.trainer.optimizer_state()["step"] in (0, 1)                      # and fresh moments / step counter
    motion, cond = _batches(cfg, 1, start=2)[0]
    second.run_step(motion, cond)
    assert abs(second.cur_lr - 1e-4 * (1 - 2 / 10)) < 1e-12                        # _anneal_lr after the step: (step 0 + resume 2) / 10


# ---------------------------------------------------------------------------------------------------------------
# Full training batch (B=512, the reference's default -b 512): size-independent properties instead of the CPU oracle
def test_full_batch_gradient_is_deterministic_and_equals_mean_of_half_batches():
    """Every loss term is a mean over the batch, so grad(full) == (grad(first half) + grad(second half)) / 2; and the
    step has no float atomics, so repeating it gives bit-identical gradients."""
    import torch
    cfg, sd, tr = make("ted")
    Bf = 512
    x_start, y, noise, drop, eps = synth.make_train_batch(cfg, Bf, 3)
    t = np.random.Generator(np.random.PCG64(9)).integers(0, 1000, size=(Bf,))
    terms = tr.forward_backward(x_start, t, noise, y, drop, eps)
    g_full = tr.grad.clone()
    terms2 = tr.forward_backward(x_start, t, noise, y, drop, eps)
    assert torch.equal(g_full, tr.grad) and terms["total"] == terms2["total"]
    halves, losses = [], []
    for sl in (slice(0, Bf // 2), slice(Bf // 2, Bf)):
        yy = {k: v[sl] for k, v in y.items()}
        tt = tr.forward_backward(x_start[sl], t[sl], noise[sl], yy, drop[sl], eps[sl])
        halves.append(tr.grad.clone())
        losses.append(tt["total"])
    g_mean = (halves[0] + halves[1]) / 2
    assert abs(terms["total"] - (losses[0] + losses[1]) / 2) < 1e-5 * abs(terms["total"])
    g_full_np, g_mean_np = g_full.cpu().numpy(), g_mean.cpu().numpy()
    for k, (o, n) in tr.params.items():
        if k in NULL_GRAD:
            continue
        a, b = g_full_np[o:o + n], g_mean_np[o:o + n]
        assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max() + 1e-9, k
    assert np.isfinite(g_full_np).all()


def test_gradients_strict_on_a_batch_without_kink_proximal_activations():
    """Same comparison with EVERY tensor held to the strict max-abs bound, on a (deterministically searched) 2-sample batch
    whose normalised activations all stay >= 3e-6 away from the LeakyReLU kink, so that no slope is ambiguous."""
    import torch
    cfg, sd, tr = make("ted")
    oracle = tro.TrainOracle(sd, cfg.n_prefix_tokens)
    Bs = 2
    chosen = None
    for step in range(40):
        x_start, y, noise, drop, eps = synth.make_train_batch(cfg, Bs, step)
        if min(tro.leaky_kink_margins(oracle.P, torch.from_numpy(y["audio_input"]))) >= 3e-6:
            chosen = step
            break
    assert chosen is not None, "no kink-free batch among 40 candidates"
    t = np.array([17, 803])
    tr.forward_backward(x_start, t, noise, y, drop, eps)
    _, _, ograds, _ = oracle.forward_backward(x_start, t, noise, y, drop, eps)
    g = tr.grads()
    worst = 0.0
    for k in g:
        if k in NULL_GRAD:
            continue
        rel = np.abs(g[k] - ograds[k]).max() / (np.abs(ograds[k]).max() + 1e-12)
        worst = max(worst, rel)
        assert rel < 2e-4, (k, rel, chosen)
    print(f"kink-free batch = step {chosen}; worst relative gradient error {worst:.2e}")


def test_batch_size_may_change_between_steps():
    """The last batch of an epoch is smaller and buffers grow on demand: results must not depend on the call history."""
    import torch
    cfg, sd, tr = make("ted")
    _, _, fresh = make("ted")
    x, y, noise, drop, eps = synth.make_train_batch(cfg, 6, 2)
    t = np.array([3, 999, 500, 0, 42, 77])
    sub = lambda s: (x[s], t[s], noise[s], {k: v[s] for k, v in y.items()}, drop[s], eps[s])
    tr.forward_backward(*sub(slice(0, 2)))             # small first (allocates for 2) ...
    tr.forward_backward(*sub(slice(0, 6)))             # ... grow to 6 ...
    tr.forward_backward(*sub(slice(2, 5)))             # ... shrink to 3
    a = tr.grad.clone()
    fresh.forward_backward(*sub(slice(2, 5)))
    assert torch.equal(a, fresh.grad)


def test_trainloop_beat_variant_runs_and_matches_oracle_loss():
    """BEAT (47x6 pose features, style + emotion prefix tokens) through the TrainLoop drop-in: first-step loss vs the oracle."""
    import torch
    from types import SimpleNamespace
    from livelyspeaker_amd.model_util import create_model_and_diffusion
    from livelyspeaker_amd.train_loop import TrainLoop
    cfg = synth.CONFIGS["beat"]
    margs = SimpleNamespace(mdm_condm="text", latent_dim=512, ff_size=1024, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                            emb_trans_dec=False, dataset="humanml", lang_model=None, mlpact="silu", diffusion_steps=1000,
                            noise_schedule="cosine", sigma_small=True, lambda_vel=1.0, lambda_rcxyz=0.0, lambda_fc=0.0, njoints=cfg.njoints)
    model, diffusion = create_model_and_diffusion(margs, "", dataset="beat")
    sd = synth.make_state_dict(cfg)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model.to("cuda:0")
    model.train()
    targs = SimpleNamespace(batch_size=4, lr=1e-4, weight_decay=0.0, lr_anneal_steps=0, log_interval=1000, save_interval=1000,
                            resume_checkpoint="", epochs=1, save_dir="/tmp/ls_train_beat", overwrite=True, dataset="beat")
    loop = TrainLoop(targs, None, model, diffusion, None)
    x_start, y, _, _, _ = synth.make_train_batch(cfg, 4, 0)
    cond = {"y": {k: torch.from_numpy(v) for k, v in y.items()}}
    np.random.seed(8); torch.manual_seed(8)
    loop.run_step(torch.from_numpy(x_start), cond)
    np.random.seed(8); torch.manual_seed(8)
    t = np.random.choice(1000, size=(4,), p=np.ones(1000) / 1000)
    noise = torch.randn(tuple(x_start.shape)).numpy()
    drop = torch.bernoulli(torch.ones(4) * 0.1).numpy()
    eps = torch.randn(4, 1, 512).numpy().reshape(4, 512)
    oracle = tro.TrainOracle(sd, cfg.n_prefix_tokens)
    _, ototal, _, _ = oracle.forward_backward(x_start, t, noise, y, drop, eps)
    assert abs(loop.last_losses["total"] - ototal) <= 1e-4 * max(1.0, abs(ototal))
