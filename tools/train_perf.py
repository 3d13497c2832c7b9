#!/usr/bin/env python3
"""Time the HIP training step (GPU box): python tools/train_perf.py [dataset] [B] [steps] [library.so]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from livelyspeaker_amd import _lib, synth

ds = sys.argv[1] if len(sys.argv) > 1 else "ted"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
if len(sys.argv) > 4:
    _lib.use_library(sys.argv[4])            # an A/B variant of the library (livelyspeaker_amd.build.build_library(defines=..., out=...))
cfg = synth.CONFIGS[ds]
tr = _lib.Trainer(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions)
tr.load_state_dict(synth.make_state_dict(cfg))
tr.set_schedule(synth.schedule(1000))
x_start, y, noise, drop, eps = synth.make_train_batch(cfg, B, 0)
dev = torch.device("cuda", 0)
tod = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
x_start, noise, drop, eps = tod(x_start), tod(noise), tod(drop), tod(eps)
y = {k: tod(v) for k, v in y.items()}
t = np.random.Generator(np.random.PCG64(0)).integers(0, 1000, size=(B,))
for i in range(steps + 1):
    t0 = time.perf_counter()
    terms = tr.forward_backward(x_start, t, noise, y, drop, eps)
    t1 = time.perf_counter()
    tr.adamw()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"step {i}: total {terms['total']:.5f}  fwd {terms['fwd_ms']:.2f} ms  bwd {terms['bwd_ms']:.2f} ms  fb wall {(t1-t0)*1e3:.2f} ms  adamw {(t2-t1)*1e3:.2f} ms", flush=True)
print("max mem GB", torch.cuda.mem_get_info())
