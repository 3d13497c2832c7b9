#!/usr/bin/env python3
"""SAG decode + the refinement's once-per-call stage at B = 512, back to back (the reference's order) vs ls_prepare_async enqueued under
the decode (RAG.prefetch_condition): wall time of the pair and each stream's own span.  Run on the GPU box."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import livelyspeaker_ted as ex
B = 512
cfg, model, diffusion, sag, _ = ex.build()
_, batch, cond = ex.make_inputs(cfg, B)
inner = model.model if hasattr(model, "model") else model
inner.cache_conditioning = False
eng = inner.engine(); seng = sag.engine()
def serial():
    inner._cond_key = None
    d = sag(batch)["output"]
    inner._engine_prepared(cond["y"])
    eng.synchronize()
def overlap():
    inner._cond_key = None
    inner.prefetch_condition(cond["y"])
    d = sag(batch)["output"]
    inner._engine_prepared(cond["y"])
    eng.synchronize()
for f in (serial, overlap, serial, overlap):
    for _ in range(5): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    print(f.__name__, "wall ms median", round(1e3 * float(np.median(ts)), 3), "sag", round(seng.last_decode_ms(), 3), "prepare", round(eng.timing()["prepare_ms"], 3))
