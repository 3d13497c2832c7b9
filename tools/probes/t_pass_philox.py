import sys, numpy as np
sys.path.insert(0, ".")
from livelyspeaker_amd import _lib, synth
cfg = synth.TED
outs = {}
for path in ("fused", "pass"):
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, path=path)
    eng.load_state_dict(synth.make_state_dict(cfg))
    eng.set_schedule(synth.schedule(20))
    for B, off in ((128, 0), (128, 128), (128, 0)):
        eng.prepare(synth.make_cond(cfg, B, seed=off + 1))
        o = eng.sample(sampler=0, philox_seed=77, sample_offset=off)
        outs[(path, B, off, len([k for k in outs if k[0] == path]))] = o
        print(path, B, off, eng.timing()["step_path"], float(np.abs(o).sum()))
    eng.close()
ks = [k for k in outs if k[0] == "fused"]
for k in ks:
    k2 = ("pass",) + k[1:]
    print(k[1:], np.abs(outs[k] - outs[k2]).max())
