"""Step time in the opt-in bf16x3 mode by kernel family (fused: one workgroup per clip; pass: one per (clip, CFG pass), two per CU) and
batch; hipGraph replay, Philox noise.  python tools/bf16x3_time.py [paths] [ted batches] [beat batches]; LS_LIB=<library.so> for an A/B."""
import sys, os
sys.path.insert(0, ".")
from livelyspeaker_amd import _lib, synth
if os.environ.get("LS_LIB"): _lib.use_library(os.environ["LS_LIB"])
paths = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fused", "pass"]
bt = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [512]
bb = [int(b) for b in sys.argv[3].split(",")] if len(sys.argv) > 3 else [256]
for ds, Bs in (("ted", bt), ("beat", bb)):
    cfg = synth.CONFIGS[ds]
    for path in paths:
        eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, path=path)
        eng.load_state_dict(synth.make_state_dict(cfg))
        eng.set_precision("bf16x3")
        eng.set_schedule(synth.schedule(50))
        for B in Bs:
            eng.prepare(synth.make_cond(cfg, B))
            eng.sample(sampler=0, philox_seed=1)
            best = 1e9
            for _ in range(3):
                eng.sample(sampler=0, philox_seed=1)
                best = min(best, eng.timing()["loop_ms"] / 50)
            print(f"{ds} {path:5s} B={B:4d} bf16x3: {best:.4f} ms/step  {B * 34 / best:8.0f} pose-frames/s  (path {eng.timing()['step_path']})", flush=True)
        eng.close()
