import sys, numpy as np, torch
sys.path.insert(0, ".")
from livelyspeaker_amd import synth
from livelyspeaker_amd.cfg_sampler import ClassifierFreeSampleModel
from livelyspeaker_amd.model_util import create_model_and_diffusion
import bench
cfg = synth.TED
dev = torch.device("cuda", 0)
B, steps = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 60
res = {}
import os
UG = os.environ.get("UG", "1") == "1"; CC = os.environ.get("CC", "0") == "1"
for path in ("fused", "auto"):
    model, diffusion = create_model_and_diffusion(bench.mk_args(cfg, steps), "", dataset="ted")
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()}
    model.load_state_dict(sd, strict=False); model.to(dev); model.eval()
    model.step_path = path
    model.cache_conditioning = CC
    cfgm = ClassifierFreeSampleModel(model)
    diffusion.noise_source = "philox"; diffusion.use_graph = UG; diffusion.sample_offset = 0
    shape = (B, cfg.njoints, cfg.nfeats, cfg.nframes)
    outs = []
    for i in range(4):
        y = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_cond(cfg, B, scale=1.5, seed=3).items()}
        diffusion.philox_seed = 12345
        o = diffusion.p_sample_loop(cfgm, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=0, init_image=None, progress=False, dump_steps=None, noise=None, const_noise=False)
        torch.cuda.synchronize()
        outs.append(o.clone())
        print(path, i, model.engine().timing()["step_path"], float(o.abs().sum()))
        if path != "fused":
            from livelyspeaker_amd import _lib
            eng = model.engine()
            raw = np.empty(B, np.float32)
            eng.lib.ls_read(eng.h, b"pass_tickets", raw.ctypes.data_as(_lib.c_f32p), raw.size)
            tk = raw.view(np.uint32)
            print("tickets:", np.unique(tk, return_counts=True), tk[:12])
    res[path] = outs
for i in range(4):
    d = (res["fused"][i] - res["auto"][i]).abs().amax(dim=(1, 2, 3))
    print(i, float(d.max()), (d > 1e-3).nonzero().flatten().tolist()[:20])
