import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from livelyspeaker_amd import _lib, synth
cfg = synth.TED
eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len)
eng.load_state_dict(synth.make_state_dict(cfg))
y = synth.make_cond(cfg, 512)
for _ in range(3):
    eng.prepare(y)
print(eng.timing()["prepare_ms"])
