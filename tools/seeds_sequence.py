import sys, time
sys.path.insert(0, ".")
import torch, bench
dev = torch.device("cuda", 0)
def seeds(tag):
    r = bench.other_config_leg("beat", 256, dev, torch.cuda.synchronize, steps=400, noise="torch_cpu")
    i = r["identical_seeds"]; print(tag, "host RNG", i["host_rng_ms_per_step"], "loop", i["loop_ms_per_step"], flush=True)
seeds("fresh")
bench.other_config_leg("ted", 64, dev, torch.cuda.synchronize, steps=200)
seeds("after ted64 philox")
bench.other_config_leg("ted", 384, dev, torch.cuda.synchronize, steps=200)
seeds("after ted384")
import numpy as np
a = np.random.rand(2000, 2000); 
for _ in range(5): a @ a
seeds("after numpy matmul")
x = torch.randn(4000, 4000); 
for _ in range(5): x @ x
seeds("after torch cpu matmul")
