#!/usr/bin/env python3
"""GPU time of ls_prepare at B = 512, steady state (HIP events on the engine's stream): python tools/prepare_only.py [library.so]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from livelyspeaker_amd import _lib, synth

if len(sys.argv) > 1:
    _lib.use_library(sys.argv[1])
cfg = synth.TED
eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len)
eng.load_state_dict(synth.make_state_dict(cfg))
import torch
y = {k: torch.from_numpy(v).cuda() for k, v in synth.make_cond(cfg, 512).items()}
# The shader clock idles at ~100 MHz and needs a few hundred ms of load to reach its 2.4 GHz ceiling (rocm-smi while bench.py runs);
# 30 one-millisecond calls with a host sync each never get there.  Heat it with a GEMM loop first, then time calls back to back.
heat = torch.randn(4096, 4096, device="cuda")
for _ in range(60):
    heat = torch.mm(heat, heat) * 1e-3
torch.cuda.synchronize()
ts = []
for _ in range(30):
    eng.prepare(y)
    ts.append(eng.timing()["prepare_ms"])
print(os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "in-tree", "prepare ms: median of last 15 =", round(float(np.median(ts[15:])), 4))
