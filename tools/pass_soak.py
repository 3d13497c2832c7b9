#!/usr/bin/env python3
"""Hand-off soak of the one-pass-per-workgroup kernel on the GPU box: for several (data set, batch, guidance) cases -- grids below, at and
beyond the chip's residency, the 8-wave and the 4-wave form -- replay a 20-step loop N times while two other handles (fused kernel,
sample-split kernel) and a torch matmul loop keep the chip and the memory system busy from other threads, and require every replay to be
BITWISE the first (a stale or torn read of the other pass's output, a ticket taken twice, a lost reset would show as a difference).
    python tools/pass_soak.py [replays per case, default 150]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livelyspeaker_amd import _lib, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
stop = threading.Event()
errs = []


def hammer(path, ds, B):
    try:
        cfg = synth.CONFIGS[ds]
        e = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, path=path)
        e.load_state_dict(synth.make_state_dict(cfg))
        e.set_schedule(synth.schedule(40))
        e.prepare(synth.make_cond(cfg, B, seed=9))
        while not stop.is_set():
            e.sample(sampler=0, philox_seed=3)
        e.close()
    except Exception as ex:      # noqa: BLE001
        errs.append(repr(ex))


def torch_load():
    try:
        import torch
        a = torch.randn(4096, 4096, device="cuda")
        while not stop.is_set():
            (a @ a).sum().item()
    except Exception as ex:      # noqa: BLE001
        errs.append(repr(ex))


ths = [threading.Thread(target=hammer, args=("fused", "ted", 100)), threading.Thread(target=hammer, args=("coop", "ted", 24)), threading.Thread(target=torch_load)]
for t in ths:
    t.start()
bad = 0
try:
    for ds, B, scale in (("ted", 128, 1.5), ("ted", 37, 1.5), ("ted", 300, 1.5), ("ted", 513, 1.5), ("beat", 128, 1.5), ("beat", 260, 1.5), ("ted", 300, 1.0), ("beat", 64, 1.0)):
        cfg = synth.CONFIGS[ds]
        eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, path="pass")
        eng.load_state_dict(synth.make_state_dict(cfg))
        eng.set_schedule(synth.schedule(20))
        eng.prepare(synth.make_cond(cfg, B, scale=scale))
        ref = eng.sample(sampler=0, philox_seed=11)
        t0, diff = time.time(), 0
        for i in range(N):
            if not np.array_equal(ref, eng.sample(sampler=0, philox_seed=11)):
                diff += 1
        bad += diff
        print(f"{ds} B={B} scale={scale}: {N} replays, {diff} differ, {time.time() - t0:.1f} s, single_pass={eng.timing()['single_pass']}", flush=True)
        eng.close()
finally:
    stop.set()
    for t in ths:
        t.join()
print("errors in load threads:", errs)
print("PASS SOAK", "OK" if not bad and not errs else "FAILED")
sys.exit(1 if bad or errs else 0)
