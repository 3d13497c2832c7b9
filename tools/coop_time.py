"""Step time of the sampling loop by kernel path and batch: python tools/coop_time.py [ted|beat] [steps]
Prints ms/step (hipGraph replay, Philox noise) for each (path, B) -- the throughput-vs-batch table of profiles/."""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from livelyspeaker_amd import _lib, synth  # noqa: E402

if os.environ.get("LS_LIB"):
    _lib.use_library(os.environ["LS_LIB"])


def main():
    ds = sys.argv[1] if len(sys.argv) > 1 else "ted"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    batches = [int(b) for b in sys.argv[3].split(",")] if len(sys.argv) > 3 else [4, 16, 32, 64, 128, 192, 256, 320, 384, 512]
    paths = sys.argv[4].split(",") if len(sys.argv) > 4 else ["coop", "batch", "fused", "auto"]
    cfg = synth.CONFIGS[ds]
    sd = synth.make_state_dict(cfg)
    print(f"# {ds}, {steps}-step DDPM, CFG 1.5, Philox; ms per step | pose-frames/s at 1000 steps")
    for path in paths:
        eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, path=path)
        eng.load_state_dict(sd)
        eng.set_schedule(synth.schedule(steps))
        for B in batches:
            eng.prepare(synth.make_cond(cfg, B))
            eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=1)          # capture
            best = 1e9
            for _ in range(3):
                eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=1)
                best = min(best, eng.timing()["loop_ms"] / steps)
            t = eng.timing()
            print(f"{path:6s} B={B:4d}  {best:8.4f} ms/step  {B * 34 / best:10.0f} frames/s  path={t['step_path']}" + (f" +{t['tail_samples']}@{t['tail_path']}" if t['tail_samples'] else "")
                  + (f" +{t['tail2_samples']}@{t['tail2_path']}" if t['tail2_samples'] else ""), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
