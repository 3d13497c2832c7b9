#!/usr/bin/env python3
"""In-kernel phase profile of ls::k_coop (debug aid; -DLS_DEBUG build of the library, variants/debug.so: build it in the container
with `python tools/phase_profile.py build`).  LS_PROF=<workgroup>: lane 0 of each wave of that workgroup records s_memtime at
phase boundaries.  usage: python tools/coop_profile.py [ted|beat] [B] [coop|coop8|coop4|coop2]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LS_PROF", "0")
from livelyspeaker_amd import _lib, synth          # noqa: E402
from livelyspeaker_amd import build as _build      # noqa: E402

_lib.use_library(os.path.join(_build.ROOT, "variants", "debug.so"))
ds = sys.argv[1] if len(sys.argv) > 1 else "ted"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = synth.CONFIGS[ds]
path = sys.argv[3] if len(sys.argv) > 3 else "coop"
eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, path=path)
eng.load_state_dict(synth.make_state_dict(cfg))
eng.set_schedule(synth.schedule(8))
eng.prepare(synth.make_cond(cfg, B))
for _ in range(2):
    eng.sample(sampler=0, philox_seed=1)
raw = np.empty(8 * 96 * 2, np.float32)
eng.lib.ls_read(eng.h, b"prof", raw.ctypes.data_as(_lib.c_f32p), raw.size)
st = raw.view(np.uint64).reshape(8, 96).astype(np.float64)[:4]
L = 8
E = 2 + 8 * L
print(f"{ds} B={B} {path} workgroup {os.environ['LS_PROF']}: total (wave mean) = {np.mean(st[:, E + 3] - st[:, 0]):.0f}  [s_memtime ticks]")
print(f"  embed                      : {np.mean(st[:, 1] - st[:, 0]):9.0f}")
names = ["temb + LN1 partials/publish", "SYNC1 wait + merge", "LN1 store + token mixing", "rows publish + LN2 partials", "ready-flag wait + pulls issued",
         "own k blocks", "LN2 gather + pulls landed", "remaining k blocks", "partial-sum swap + epilogue"]
acc = np.zeros(len(names))
prev = st[:, 1].copy()
for l in range(L):
    b = 2 + 8 * l
    seq = [b + 5, b + 0, b + 1, b + 2, 70 + 2 * l, b + 6, 71 + 2 * l, b + 3, b + 4]
    for i, pnt in enumerate(seq):
        acc[i] += np.mean(st[:, pnt] - prev)
        prev = st[:, pnt].copy()
for i in range(len(names)):
    print(f"  {names[i]:31s}: {acc[i] / L:9.0f} /layer")
print(f"  partial poseFinal + publish  : {np.mean(st[:, E] - prev):9.0f}")
print(f"  flags wait                   : {np.mean(st[:, E + 1] - st[:, E]):9.0f}")
print(f"  reduce + CFG + sampler update: {np.mean(st[:, E + 3] - st[:, E + 1]):9.0f}")
print("  per-layer totals (wave 0):", [int(st[0, 2 + 8 * l + 4] - (st[0, 2 + 8 * (l - 1) + 4] if l else st[0, 1])) for l in range(L)])
