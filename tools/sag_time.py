import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from livelyspeaker_amd import _lib, synth
cfg = synth.TED
B = 512
eng = _lib.SagEngine()
eng.load_state_dict(synth.make_sag_state_dict(cfg))
x = torch.from_numpy(synth.make_init_image(cfg, B)).cuda()
z = torch.from_numpy(synth.make_text_features(B)).cuda()
for i in range(3): out = eng.decode(x, z)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10): out = eng.decode(x, z)
torch.cuda.synchronize(); print("sag decode ms", (time.perf_counter() - t0) * 100)
