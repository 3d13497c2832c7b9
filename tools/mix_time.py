"""Step time of the 150-frame model (one-launch mixer or batch-level kernels): python tools/mix_time.py [B] [steps] [auto|batch]; LS_LIB=<library.so> for an A/B."""
import os, sys
sys.path.insert(0, ".")
from livelyspeaker_amd import _lib, synth
if os.environ.get("LS_LIB"): _lib.use_library(os.environ["LS_LIB"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
path = sys.argv[3] if len(sys.argv) > 3 else "auto"
cfg = synth.BEAT150
eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, nframes=cfg.nframes)
eng.load_state_dict(synth.make_state_dict(cfg))
if path == "batch": eng.set_path("batch")
eng.set_schedule(synth.schedule(steps))
eng.prepare(synth.make_cond(cfg, B))
eng.sample(sampler=0, philox_seed=1)
best = 1e9
for _ in range(3):
    eng.sample(sampler=0, philox_seed=1)
    best = min(best, eng.timing()["loop_ms"] / steps)
print(f"beat150 B={B} {path}: {best:.4f} ms/step  {B * 150 / best:.0f} pose-frames/s  frac {2 * 913432576 * B / (best * 1e-3) / 1e12 / 157.3:.4f}", flush=True)
eng.close()
