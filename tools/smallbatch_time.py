#!/usr/bin/env python
"""ms per diffusion step of the two step implementations of a 34-frame model over the batch size (GPU box):
    python tools/smallbatch_time.py [ted|beat]
fused = one workgroup per sample (ls_step_kernel.h), batch = batch-level kernels (ls_long.hip); Philox noise, hipGraph, 60 steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livelyspeaker_amd import _lib, synth

ds = sys.argv[1] if len(sys.argv) > 1 else "ted"
if len(sys.argv) > 2:
    _lib.use_library(sys.argv[2])
cfg = synth.CONFIGS[ds]
print(f"{ds}: B | fused ms/step | batch-level ms/step | ratio")
for B in (1, 2, 4, 8, 16, 32, 48, 64, 96, 128, 144, 160, 176, 192, 256):
    res = {}
    for path in ("fused", "batch"):
        eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, path=path)
        eng.load_state_dict(synth.make_state_dict(cfg))
        eng.set_schedule(synth.schedule(60))
        eng.prepare(synth.make_cond(cfg, B, scale=1.5))
        for _ in range(3):
            eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=3)
        tm = eng.timing()
        res[path] = tm["loop_ms"] / tm["n_step_launches"]
        eng.close()
    print(f"{B:4d} | {res['fused']:.4f} | {res['batch']:.4f} | {res['fused'] / res['batch']:.2f}")
