#!/usr/bin/env python3
"""Debug helper (test infrastructure, run by hand on the GPU box): per-parameter gradient error of the HIP training step vs the
torch-CPU oracle.  Lives under tests/ because only tests/, smoke() and bench.py's checker legs may use oracle/."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from livelyspeaker_amd import _lib, synth
from oracle import train_oracle as tro, rag_oracle as orc

ds = sys.argv[1] if len(sys.argv) > 1 else "ted"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = synth.CONFIGS[ds]
sd = synth.make_state_dict(cfg)
tr = _lib.Trainer(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions)
tr.load_state_dict(sd)
tr.set_schedule(orc.Schedule(1000, ""))
oracle = tro.TrainOracle(sd, cfg.n_prefix_tokens)
for step in range(2):
    x_start, y, noise, drop, eps = synth.make_train_batch(cfg, B, step)
    t = np.random.Generator(np.random.PCG64(step)).integers(0, 1000, size=(B,))
    t[0], t[1] = 0, 999
    terms = tr.forward_backward(x_start, t, noise, y, drop, eps)
    oterms, ototal, ograds, oout = oracle.forward_backward(x_start, t, noise, y, drop, eps)
    print("terms hip", {k: round(v, 6) for k, v in terms.items()})
    print("terms orc", {k: round(v, 6) for k, v in oterms.items()}, round(ototal, 6))
    out = tr.read("out", (B, cfg.nframes, cfg.jf)).reshape(B, cfg.nframes, cfg.njoints, cfg.nfeats).transpose(0, 2, 3, 1)
    print("out max|d|", np.abs(out - oout).max())
    g = tr.grads()
    for k in g:
        sc = np.abs(ograds[k]).max() + 1e-12
        d = np.abs(g[k] - ograds[k]).max()
        flag = "" if d / sc < 2e-4 or k in ("audio_encoder.feat_extractor.0.bias", "audio_encoder.feat_extractor.3.bias", "audio_encoder.feat_extractor.6.bias") else "   <<<<<<"
        print(f"  {k:55s} max|g| {sc:.3e}  err {d:.3e}  rel {d/sc:.2e}{flag}")
    tr.adamw()
    oracle.optimizer_step(ograds)
psd, osd = tr.state_dict(), oracle.state_dict()
print("params after 2 steps: max|d| =", max(np.abs(psd[k] - osd[k]).max() for k in psd if not k.endswith((".0.bias", ".3.bias", ".6.bias")) or "audio" not in k))
