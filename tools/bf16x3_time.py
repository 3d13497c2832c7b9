"""Step time of the fused kernel in the opt-in bf16x3 mode (TED B = 512, BEAT B = 256; hipGraph replay); LS_LIB=<library.so> for an A/B."""
import sys, os
sys.path.insert(0, ".")
from livelyspeaker_amd import _lib, synth
if os.environ.get("LS_LIB"): _lib.use_library(os.environ["LS_LIB"])
for ds, B in (("ted", 512), ("beat", 256)):
    cfg = synth.CONFIGS[ds]
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, path="fused")
    eng.load_state_dict(synth.make_state_dict(cfg))
    eng.set_precision("bf16x3")
    eng.set_schedule(synth.schedule(50))
    eng.prepare(synth.make_cond(cfg, B))
    eng.sample(sampler=0, philox_seed=1)
    best = 1e9
    for _ in range(3):
        eng.sample(sampler=0, philox_seed=1)
        best = min(best, eng.timing()["loop_ms"] / 50)
    print(f"{ds} B={B} bf16x3: {best:.4f} ms/step")
    eng.close()
