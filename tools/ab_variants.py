#!/usr/bin/env python3
"""A/B harness for kernel schedule variants.

  python tools/ab_variants.py build NAME=DEF1,DEF2 ...   # here (hipcc cross-compiles): variants/NAME.so
  python tools/ab_variants.py run [B] [steps]            # on the GPU box: times ls::k_step for every variants/*.so

Each variant is timed in its own subprocess (one HIP module per process), `rounds` times interleaved, and the
median kernel time from HIP events on the engine's stream is reported.  Variants live under variants/
(git-ignored; they travel to the GPU box with the snapshot).
"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "variants")
sys.path.insert(0, ROOT)

CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
from livelyspeaker_amd import _lib, synth
ds, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
_lib.use_library(sys.argv[4])
cfg = synth.CONFIGS[ds]
eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, nframes=cfg.nframes, path=os.environ.get('LS_AB_PATH') or None)
eng.load_state_dict(synth.make_state_dict(cfg))
if os.environ.get("LS_PRECISION"): eng.set_precision(os.environ["LS_PRECISION"])
eng.set_schedule(synth.schedule(steps))
eng.prepare(synth.make_cond(cfg, B))
ts = []
for i in range(4):
    out = eng.sample(sampler=0, philox_seed=1)
    t = eng.timing()
    ts.append(t["loop_ms"] / t["n_step_launches"])
import numpy as np
print(json.dumps({"ms": sorted(ts[1:])[1], "chk": float(np.abs(out).sum())}))
''' % ROOT


def build(specs):
    from livelyspeaker_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    for spec in specs:
        name, _, defs = spec.partition("=")
        out = os.path.join(VDIR, name + ".so")
        items = [d for d in defs.split(",") if d]          # NAME=DEF1,DEF2,@-mllvm,@-some-flag : '@' items are raw compiler flags
        b.build_library(force=True, defines=[d for d in items if not d.startswith("@")], out=out,
                        extra_flags=[d[1:] for d in items if d.startswith("@")])
        print("built", out)


def run(ds="ted", B=512, steps=40, rounds=3):
    libs = sorted(glob.glob(os.path.join(VDIR, "*.so")))
    res = {os.path.basename(l)[:-3]: [] for l in libs}
    chk = {}
    for _ in range(rounds):
        for lib in libs:
            out = subprocess.run([sys.executable, "-c", CHILD, ds, str(B), str(steps), lib], capture_output=True, text=True)
            name = os.path.basename(lib)[:-3]
            try:
                r = json.loads(out.stdout.strip().splitlines()[-1])
                res[name].append(r["ms"])
                chk[name] = r["chk"]
            except Exception:
                res[name].append(float("nan"))
                print(name, "FAILED:", out.stderr[-400:])
    for name, v in sorted(res.items(), key=lambda kv: sorted(kv[1])[len(kv[1]) // 2]):
        print(f"{name:24s} median {sorted(v)[len(v) // 2]:.4f} ms   all {['%.4f' % x for x in v]}   checksum {chk.get(name)}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(*(sys.argv[2:3] or ["ted"]), *[int(x) for x in sys.argv[3:]])
