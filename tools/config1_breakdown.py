"""Where a call of the reference's own test shape (4 clips, 50 steps) spends its time: prepare / loop / wall (GPU box)."""
import sys, time
sys.path.insert(0, ".")
import torch
from livelyspeaker_amd import _lib, synth
cfg = synth.TED
eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len)
eng.load_state_dict(synth.make_state_dict(cfg))
eng.set_schedule(synth.schedule(50))
y = synth.make_cond(cfg, 4)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.prepare(y)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out = eng.sample(sampler=0, philox_seed=1)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    t = eng.timing()
    print(f"prepare wall {1e3*(t1-t0):.3f} ms, sample wall {1e3*(t2-t1):.3f} ms;", {k: round(v, 3) if isinstance(v, float) else v for k, v in t.items() if k.endswith("_ms") or k in ("n_step_launches", "step_path")})
