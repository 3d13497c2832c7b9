#!/usr/bin/env python3
"""In-kernel phase profile of ls::k_mix (-DLS_DEBUG build, variants/debug.so: `python tools/phase_profile.py build`).  LS_PROF=<workgroup>.
usage: python tools/mix_profile.py [B]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LS_PROF", "0")
from livelyspeaker_amd import _lib, synth
from livelyspeaker_amd import build as _build
_lib.use_library(os.path.join(_build.ROOT, "variants", "debug.so"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = synth.BEAT150
eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, nframes=cfg.nframes)
eng.load_state_dict(synth.make_state_dict(cfg))
eng.set_schedule(synth.schedule(6))
eng.prepare(synth.make_cond(cfg, B))
for _ in range(2):
    eng.sample(sampler=0, philox_seed=1)
raw = np.empty(8 * 96 * 2, np.float32)
eng.lib.ls_read(eng.h, b"prof", raw.ctypes.data_as(_lib.c_f32p), raw.size)
st = raw.view(np.uint64).reshape(8, 96).astype(np.float64)
L = 8
names = ["temb + LN1 partials / publish + weight requests", "LN1 gather (hop)", "LN1 apply + barrier", "token mixing + SiLU", "rows publish + LN2 partials + ring fill",
         "own blocks 0-3 (+ flag poll, refills)", "own blocks 4-7 (+ LN2 gather)", "other slices' 24 blocks", "epilogue + barrier"]
acc = np.zeros(len(names))
for l in range(L):
    b = 10 * l
    seq = [(1 if l == 0 else b + 1, b + 2), (b + 2, b + 3), (b + 3, b + 4), (b + 4, b + 5), (b + 5, b + 6), (b + 6, b + 7), (b + 7, b + 8), (b + 8, b + 9), (b + 9, b + 11)]
    for i, (x, y) in enumerate(seq):
        acc[i] += np.mean(st[:, y] - st[:, x])
print(f"beat150 B={B} workgroup {os.environ['LS_PROF']}: load {np.mean(st[:, 1] - st[:, 0]):.0f}; per layer (wave mean, s_memtime ticks):")
for i, n in enumerate(names):
    print(f"  {n:48s}: {acc[i] / L:9.0f}")
print(f"  per-layer total {acc.sum() / L:.0f}; wave 0 / wave 4 layer ends: {[int(st[0, 10 * l + 11] - st[0, 1]) for l in range(L)]}")
