import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from livelyspeaker_amd import _lib, synth
from oracle import rag_oracle as orc
cfg = synth.TED; B = 6
tr = _lib.Trainer(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len)
tr.load_state_dict(synth.make_state_dict(cfg)); tr.set_schedule(orc.Schedule(1000, ""))
x_start, y, noise, drop, eps = synth.make_train_batch(cfg, B, 0)
tr.forward_backward(x_start, np.arange(B) * 100, noise, y, drop, eps)
Ls = synth.audio_lengths(cfg.audio_len); C = [32, 64, 128]
for i in range(3):
    c = tr.read(f"c{i+1}", (B, C[i], Ls[i])).astype(np.float64)
    st = tr.read(f"st{i+1}", (B, C[i], 2))
    mean = c.mean(-1); rstd = 1 / np.sqrt(c.var(-1) + 1e-5)
    print(i, "mean err", np.abs(st[..., 0] - mean).max(), "rstd rel err", (np.abs(st[..., 1] - rstd) / rstd).max())
import torch, torch.nn.functional as F
sd = synth.make_state_dict(cfg)
h = torch.from_numpy(y["audio_input"]).unsqueeze(1)
for i, (idx, s, p) in enumerate(zip((0, 3, 6, 9), (5, 6, 6, 6), (1600, 0, 0, 0))):
    h = F.conv1d(h, torch.from_numpy(sd[f"audio_encoder.feat_extractor.{idx}.weight"]), torch.from_numpy(sd[f"audio_encoder.feat_extractor.{idx}.bias"]), stride=s, padding=p)
    Lc = h.shape[-1]
    got = tr.read(f"c{i+1}", (B, h.shape[1], Lc))
    d = np.abs(got - h.numpy())
    print("conv", i + 1, "max|d|", d.max(), "argmax", np.unravel_index(d.argmax(), d.shape))
    if i < 3:
        h = F.leaky_relu(F.instance_norm(h), 0.3)
