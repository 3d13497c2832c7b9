import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from livelyspeaker_amd import _lib, synth
from oracle import rag_oracle as orc
cfg = synth.TED; B = 6
tr = _lib.Trainer(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len)
tr.load_state_dict(synth.make_state_dict(cfg)); tr.set_schedule(orc.Schedule(1000, ""))
x_start, y, noise, drop, eps = synth.make_train_batch(cfg, B, 0)
tr.forward_backward(x_start, np.arange(B) * 100, noise, y, drop, eps)
g = tr.grads()
np.savez(sys.argv[1], **{k: v for k, v in g.items() if "audio" in k})
