#!/usr/bin/env python3
"""Duration of every k_step workgroup of one launch (debug aid; -DLS_DEBUG build, variants/debug.so from `python
tools/phase_profile.py build`): how uniform the workgroups are -- an upper bound on what a kernel without the per-step boundary
could recover from skew.  python tools/wg_timeline.py [ted|beat] [B]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LS_PROF", "0")
from livelyspeaker_amd import _lib, synth          # noqa: E402
from livelyspeaker_amd import build as _build      # noqa: E402

_lib.use_library(os.path.join(_build.ROOT, "variants", "debug.so"))
ds = sys.argv[1] if len(sys.argv) > 1 else "ted"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = synth.CONFIGS[ds]
PATH = os.environ.get("LS_PROF_PATH", "fused")          # "fused" (k_step: B workgroups) or "pass" (k_pass: 2 B workgroups)
eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, path=PATH)
eng.load_state_dict(synth.make_state_dict(cfg))
if os.environ.get("LS_PROF_PRECISION"):
    eng.set_precision(os.environ["LS_PROF_PRECISION"])
eng.set_schedule(synth.schedule(8))
eng.prepare(synth.make_cond(cfg, B))
for _ in range(2):
    eng.sample(sampler=0, philox_seed=1)
raw = np.empty(2048 * 2, np.float32)
eng.lib.ls_read(eng.h, b"wgt", raw.ctypes.data_as(_lib.c_f32p), raw.size)
t = raw.view(np.uint64).reshape(1024, 2)[:min(1024, 2 * B if PATH == "pass" else B)].astype(np.float64)
# s_memtime has no common time base across the chip (the stamps of different workgroups differ by far more than a launch lasts), so
# only each workgroup's own duration is meaningful
dur = t[:, 1] - t[:, 0]
print(f"{ds} B={B}, last launch of the run: workgroup duration in s_memtime ticks: mean {dur.mean():.0f}  min {dur.min():.0f}  max {dur.max():.0f}  "
      f"(spread {100 * (dur.max() - dur.min()) / dur.mean():.2f} %, std {100 * dur.std() / dur.mean():.2f} %)")
hist, edges = np.histogram(dur, bins=12)
print("  histogram:", ", ".join(f"{int(lo / 1000)}k: {n}" for lo, n in zip(edges[:-1], hist)))
