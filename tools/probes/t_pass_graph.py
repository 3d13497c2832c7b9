import sys, numpy as np
sys.path.insert(0, ".")
from livelyspeaker_amd import _lib, synth
cfg = synth.TED
B, steps = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 20
DO = len(sys.argv) > 2
outs = {}
import os
PP = os.environ.get("PP", "pass")
for path in ("fused", PP):
    eng = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, path=path)
    eng.load_state_dict(synth.make_state_dict(cfg))
    eng.set_schedule(synth.schedule(steps))
    eng.prepare(synth.make_cond(cfg, B, seed=1))
    for i in range(4):
        o = eng.sample(sampler=0, philox_seed=77, device_out=DO)
        if DO: o = o.cpu().numpy()
        outs[(path, i)] = o
        if path == PP:
            raw = np.empty(B, np.float32)
            eng.lib.ls_read(eng.h, b"pass_tickets", raw.ctypes.data_as(_lib.c_f32p), raw.size)
            tk = raw.view(np.uint32)
            print("tickets:", np.unique(tk, return_counts=True), tk[:12])
        print(path, i, eng.timing()["graph_replayed"], float(np.abs(o).sum()))
    eng.close()
for i in range(4):
    d = np.abs(outs[("fused", i)] - outs[(PP, i)]).max(axis=(1, 2, 3))
    print(i, d.max(), np.nonzero(d > 1e-3)[0][:16])
