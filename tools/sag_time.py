#!/usr/bin/env python3
"""GPU time of ls_sag_decode at B = 512 (HIP events on the handle's stream), steady state: python tools/sag_time.py [library.so]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from livelyspeaker_amd import _lib, synth

if len(sys.argv) > 1:
    _lib.use_library(sys.argv[1])
eng = _lib.SagEngine()
eng.load_state_dict(synth.make_sag_state_dict())
B = 512
xb = torch.from_numpy(synth.make_cond(synth.TED, B)["origin_x"]).cuda()
zb = torch.from_numpy(synth.make_text_features(B)).cuda()
# The shader clock idles at ~100 MHz and needs a few hundred ms of load to reach its 2.4 GHz ceiling (rocm-smi while bench.py runs);
# 30 one-millisecond calls with a host sync each never get there.  Heat it with a GEMM loop first, then time calls back to back.
heat = torch.randn(4096, 4096, device="cuda")
for _ in range(60):
    heat = torch.mm(heat, heat) * 1e-3
torch.cuda.synchronize()
ts = []
for _ in range(40):
    eng.decode(xb, zb)
    ts.append(eng.last_decode_ms())
print(os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "in-tree", "sag decode ms: median of last 20 =", round(float(np.median(ts[20:])), 4),
      "min", round(min(ts), 4))
