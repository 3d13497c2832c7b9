"""Training-step oracle (oracle/train_oracle.py) against the fixtures generated from the imported reference
(tests/golden/make_golden_train.py): loss terms, gradients (statistics + samples + full small tensors) and the
parameters after two AdamW steps.  CPU only."""
import os
import zlib

import numpy as np
import pytest

from livelyspeaker_amd import synth
from oracle import train_oracle as tro

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NULL_GRAD = tuple(f"audio_encoder.feat_extractor.{i}.bias" for i in (0, 3, 6))   # true gradient == 0 (feeds InstanceNorm)
B = 6


def sample_idx(name, n, k=48):
    g = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode())))
    return g.integers(0, n, size=(k,))


def check_against_fixture(gold, prefix, name, arr, rel):
    """Compare one tensor with the fixture's (sum, abs-sum, L2) + 48 samples (+ full copy when small)."""
    a = np.asarray(arr, dtype=np.float64).ravel()
    st, sm = gold[f"{prefix}_{name}_stat"], gold[f"{prefix}_{name}_samp"]
    scale = max(float(np.abs(sm).max()), st[2] / np.sqrt(a.size), 1e-12)
    assert np.abs(a[sample_idx(name, a.size)] - sm).max() <= rel * scale + 1e-9, name
    assert abs(np.abs(a).sum() - st[1]) <= rel * st[1] + 1e-9, name
    assert abs(np.sqrt((a * a).sum()) - st[2]) <= rel * st[2] + 1e-9, name
    full = f"{prefix}_{name}_full"
    if full in gold:
        assert np.abs(np.asarray(arr) - gold[full]).max() <= rel * float(np.abs(gold[full]).max()) + 1e-9, name


@pytest.mark.parametrize("dataset", ["ted", "beat"])
def test_train_oracle_matches_reference_fixture(dataset):
    cfg = synth.CONFIGS[dataset]
    gold = np.load(os.path.join(GOLD, f"train_{dataset}_golden.npz"))
    names = [str(n) for n in gold["param_names"]]
    oracle = tro.TrainOracle(synth.make_state_dict(cfg), cfg.n_prefix_tokens)
    assert sorted(names) == sorted(oracle.P.keys())
    for step in range(2):
        x_start, y, noise, drop, eps = synth.make_train_batch(cfg, B, step)
        terms, total, grads, _ = oracle.forward_backward(x_start, gold[f"s{step}_t"], noise, y, drop, eps)
        for k in ("rot_mse", "vel_mse", "kld", "loss"):
            assert abs(terms[k] - float(gold[f"s{step}_{k}"])) <= 2e-6 * max(1.0, abs(float(gold[f"s{step}_{k}"]))), k
        assert abs(total - float(gold[f"s{step}_total"])) <= 2e-6 * max(1.0, abs(total))
        if step == 0:
            for name in names:
                if name in NULL_GRAD:
                    assert np.abs(grads[name]).max() < 1e-5
                    continue
                check_against_fixture(gold, "g", name, grads[name], 2e-4)
        oracle.optimizer_step(grads)
    sd = oracle.state_dict()
    for name in names:
        if name not in NULL_GRAD:
            check_against_fixture(gold, "p", name, sd[name], 2e-6)


def test_adamw_restatement_matches_torch_optim():
    import torch
    g = np.random.Generator(np.random.PCG64(1))
    p0 = g.standard_normal((37, 5)).astype(np.float32)
    pt = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.AdamW([pt], lr=3e-3, weight_decay=0.05)
    mine, state = {"w": torch.from_numpy(p0.copy())}, {}
    for step in range(1, 6):
        gr = g.standard_normal(p0.shape).astype(np.float32)
        pt.grad = torch.from_numpy(gr.copy())
        opt.step()
        tro.adamw_step(mine, {"w": torch.from_numpy(gr)}, state, step, lr=3e-3, weight_decay=0.05)
    assert float((pt.detach() - mine["w"]).abs().max()) < 1e-6
