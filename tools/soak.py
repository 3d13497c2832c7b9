#!/usr/bin/env python3
"""Shake-out on the GPU box: every sampling path (fused TED / BEAT, long-sequence) and the SAG decoder at odd and tile-aligned batch sizes,
back to back in one process (allocation sizes, tile paths and buffer re-use all vary); finite outputs only -- parity is tests/."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livelyspeaker_amd import _lib, synth
# long path at several batch sizes (different allocation sizes / tile paths), TED + BEAT fused at odd batches, SAG at odd batches
for ds, Bs in (("beat150", (1, 5, 8, 24, 48, 100)), ("ted", (1, 3, 37, 130, 513)), ("beat", (2, 65, 255))):
    cfg = synth.CONFIGS[ds]
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, nframes=cfg.nframes)
    eng.load_state_dict(synth.make_state_dict(cfg))
    eng.set_schedule(synth.schedule(6))
    for B in Bs:
        for scale in (1.5, 1.0):                      # 1.0: the single-pass (PAIR) kernels where the path has them
            y = synth.make_cond(cfg, B)
            if "scale" in y:
                y["scale"] = np.full_like(y["scale"], scale)
            eng.prepare(y)
            out = eng.sample(sampler=0, philox_seed=3)
        assert np.isfinite(out).all(), (ds, B)
        print(ds, B, "ok", float(np.abs(out).mean()))
    eng.close()
sag = _lib.SagEngine()
sag.load_state_dict(synth.make_sag_state_dict())
for B in (1, 2, 7, 33, 64, 129, 500):
    xb = torch.from_numpy(synth.make_cond(synth.TED, B)["origin_x"]).cuda()
    zb = torch.from_numpy(synth.make_text_features(B)).cuda()
    o = sag.decode(xb, zb)
    o = o.cpu().numpy() if hasattr(o, "cpu") else np.asarray(o)
    assert np.isfinite(o).all(), B
    print("sag", B, "ok", float(np.abs(o).mean()))
print("SOAK OK")
