"""Keep one step-kernel family busy for a few seconds (telemetry / counter runs):  python tools/busy_loop.py <ted|beat> <B> <path> <fp32|bf16x3> <seconds>
Prints the best ms per step of the 1000-step hipGraph replays it made."""
import sys
import time

sys.path.insert(0, ".")
from livelyspeaker_amd import _lib, synth  # noqa: E402

ds, B, path, prec, secs = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], float(sys.argv[5])
cfg = synth.CONFIGS[ds]
eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, path=path)
eng.load_state_dict(synth.make_state_dict(cfg))
if prec != "fp32":
    eng.set_precision(prec)
steps = 1000
eng.set_schedule(synth.schedule(steps))
eng.prepare(synth.make_cond(cfg, B))
eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=1)
best, t0 = 1e9, time.time()
while time.time() - t0 < secs:
    eng.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=1)
    best = min(best, eng.timing()["loop_ms"] / steps)
t = eng.timing()
print(f"{ds} B={B} {path} {prec}: {best:.4f} ms/step (path {t['step_path']}, slices {t['coop_slices']})")
eng.close()
