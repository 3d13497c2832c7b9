"""The identical-seeds mode (noise_source='torch_cpu') end to end at BEAT B = 256 / TED B = 512 for a few hundred steps: ms per step of the loop and of
the host draws.  python tools/seeds_time.py [beat|ted] [B] [steps]; LS_TRNG_JUMP=0|1 and LS_TRNG_THREADS=n select the native stream's form."""
import sys, time
sys.path.insert(0, ".")
import torch
import os
import bench
from livelyspeaker_amd import gaussian_diffusion as gd
if os.environ.get("LS_TAPE_MB"):
    gd.GaussianDiffusion.tape_segment_bytes = int(os.environ["LS_TAPE_MB"]) << 20
ds = sys.argv[1] if len(sys.argv) > 1 else "beat"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda", 0)
r = bench.other_config_leg(ds, B, dev, torch.cuda.synchronize, steps=steps, noise="torch_cpu")
i = r.get("identical_seeds", {})
print(f"{ds} B={B} {steps} steps: {r['value']:.0f} pose-frames/s, loop {i.get('loop_ms_per_step')} ms/step, host RNG {i.get('host_rng_ms_per_step')} ms/step, upload {i.get('upload_ms_per_step')}, segments {i.get('segments')}")
