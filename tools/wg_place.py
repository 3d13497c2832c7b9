#!/usr/bin/env python3
"""Where and when the workgroups of one k_pass launch ran (debug aid; -DLS_DEBUG build): CU of every workgroup (HW_ID / XCC_ID), and
for the workgroups that shared a CU the offsets between their start / end stamps (s_memtime of one CU is one clock).
python tools/wg_place.py [ted|beat] [B]; LS_PROF_PRECISION=bf16x3"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("LS_PROF", "0")
from livelyspeaker_amd import _lib, synth          # noqa: E402
from livelyspeaker_amd import build as _build      # noqa: E402
_lib.use_library(os.path.join(_build.ROOT, "variants", "debug.so"))
ds = sys.argv[1] if len(sys.argv) > 1 else "ted"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = synth.CONFIGS[ds]
eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, path="pass")
eng.load_state_dict(synth.make_state_dict(cfg))
if os.environ.get("LS_PROF_PRECISION"):
    eng.set_precision(os.environ["LS_PROF_PRECISION"])
eng.set_schedule(synth.schedule(8))
eng.prepare(synth.make_cond(cfg, B))
for _ in range(2):
    eng.sample(sampler=0, philox_seed=1)
n = min(1024, 2 * B)
raw = np.empty(2048 * 2, np.float32)
eng.lib.ls_read(eng.h, b"wgt", raw.ctypes.data_as(_lib.c_f32p), raw.size)
t = raw.view(np.uint64).reshape(1024, 2)[:n].astype(np.int64)
eng.lib.ls_read(eng.h, b"wgt_hw", raw.ctypes.data_as(_lib.c_f32p), raw.size)
hw = raw.view(np.uint64)[:n]
hwid, xcc = (hw & 0xffffffff).astype(np.int64), (hw >> 32).astype(np.int64) & 0xf
cu, sh, se = (hwid >> 8) & 0xf, (hwid >> 12) & 1, (hwid >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print(f"{ds} B={B}: {n} workgroups on {len(np.unique(key))} distinct CUs; workgroups per CU: {np.unique(np.unique(key, return_counts=True)[1], return_counts=True)}")
print("xcc of workgroups 0..15:", xcc[:16].tolist())
d_start, d_end, overlap = [], [], []
for k in np.unique(key):
    idx = np.nonzero(key == k)[0]
    if len(idx) >= 2:
        idx = idx[np.argsort(t[idx, 0])]
        d_start.append(t[idx[1], 0] - t[idx[0], 0]); d_end.append(t[idx[-1], 1] - t[idx[0], 1])
        if len(idx) > 2:
            overlap.append(len(idx))
ds_, de_ = np.array(d_start), np.array(d_end)
print(f"second workgroup of a CU starts {ds_.mean():.0f} ticks after the first (min {ds_.min()}, max {ds_.max()}); last ends {de_.mean():.0f} after the first ends (max {de_.max()})")
span = t[:, 1].max() - t[:, 0].min()
print(f"durations: mean {np.mean(t[:, 1] - t[:, 0]):.0f}; (unsynchronised) global span {span}")
