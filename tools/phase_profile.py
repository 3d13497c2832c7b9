#!/usr/bin/env python3
"""In-kernel phase profile of ls::k_step (debug aid).  Uses a -DLS_DEBUG build of the library (variants/debug.so; build it in
the container with `python tools/phase_profile.py build`, it travels to the GPU box): there LS_PROF=<workgroup> makes lane 0
of each wave of that workgroup record s_memtime at phase boundaries.  The shipped library has no such code.
Prints mean cycles per phase (over waves and layers)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LS_PROF", "300")
from livelyspeaker_amd import _lib, synth          # noqa: E402
from livelyspeaker_amd import build as _build      # noqa: E402

DEBUG_LIB = os.path.join(_build.ROOT, "variants", "debug.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.dirname(DEBUG_LIB), exist_ok=True)
    print(_build.build_library(defines=["LS_DEBUG"], out=DEBUG_LIB))
    sys.exit(0)
_lib.use_library(DEBUG_LIB)

ds = sys.argv[1] if len(sys.argv) > 1 else "ted"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = synth.CONFIGS[ds]
PATH = os.environ.get("LS_PROF_PATH", "fused")          # "fused" (k_step, 8 waves) or "pass" (k_pass, 4 waves)
NWV = 4 if PATH == "pass" else 8
eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, path=PATH)
eng.load_state_dict(synth.make_state_dict(cfg))
if os.environ.get("LS_PROF_PRECISION"):
    eng.set_precision(os.environ["LS_PROF_PRECISION"])
eng.set_schedule(synth.schedule(8))
eng.prepare(synth.make_cond(cfg, B))
for _ in range(2):
    eng.sample(sampler=0, philox_seed=1)
raw = np.empty(8 * 96 * 2, np.float32)
n = eng.lib.ls_read(eng.h, b"prof", raw.ctypes.data_as(_lib.c_f32p), raw.size)
st = raw.view(np.uint64).reshape(8, 96).astype(np.float64)[:NWV]
L = 8
names = ["LN1 stats(+temb)", "LN1 store+bar", "token-mix", "LN2 stats", "LN2 store+bar", "GEMM pass0+epi", "GEMM pass1+epi", "(trace)"]
print(f"{ds} B={B} path={PATH}: total cycles (wave mean) = {np.mean(st[:, 4 + 8 * L] - st[:, 0]):.0f}")
print(f"  embed                : {np.mean(st[:, 1] - st[:, 0]):9.0f}")
prev = st[:, 1].copy()
acc = np.zeros(7)
for l in range(L):
    pts = [2, 3, 4, 5, 6, 7, 9]
    for i, pnt in enumerate(pts):
        cur = st[:, pnt + 8 * l]
        acc[i] += np.mean(cur - prev)
        prev = cur
for i in range(7):
    print(f"  {names[i]:21s}: {acc[i] / L:9.0f} /layer")
print(f"  X->U + barrier       : {np.mean(st[:, 2 + 8 * L] - prev):9.0f}")
print(f"  out-proj MFMA        : {np.mean(st[:, 3 + 8 * L] - st[:, 2 + 8 * L]):9.0f}")
print(f"  OUT + combine        : {np.mean(st[:, 4 + 8 * L] - st[:, 3 + 8 * L]):9.0f}")
print("  per-wave totals:", (st[:, 4 + 8 * L] - st[:, 0]).astype(int).tolist())
if os.environ.get("LS_PROF_RAW"):
    l = 3
    base = st[:, 1 + 8 * l].min()
    print("per-wave stamps of layer 3 (cycles since earliest previous-layer end):")
    for wv in range(NWV):
        print(f"  wave {wv}:", [int(st[wv, p + 8 * l] - base) for p in (1, 2, 3, 4, 5, 6, 7, 9)])
