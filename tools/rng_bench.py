"""ms per step of ls_trng_fill_steps (the native torch-RNG stream) at TED B = 512 and BEAT B = 256 by worker-thread count: python tools/rng_bench.py [library.so]"""
import sys, time, ctypes, numpy as np, torch
sys.path.insert(0, ".")
from livelyspeaker_amd import _lib
if len(sys.argv) > 1: _lib.use_library(sys.argv[1])
from livelyspeaker_amd import torch_rng
lib = _lib.load_library()
v = torch_rng.variant()
for (B, D, J, F, T, n), label in (((512, 512, 9, 3, 34, 12), "TED B=512"), ((256, 512, 47, 6, 34, 8), "BEAT B=256")):
  for nt in (int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else "4,8,12,16,24,32,48".split(","))):
    torch.manual_seed(1)
    st = torch.get_rng_state().numpy().copy()
    eps = np.empty((n, 2, B, D), np.float32); nz = np.empty((n, B, J, F, T), np.float32)
    best = 1e9
    for rep in range(3):
        s2 = st.copy()
        t0 = time.perf_counter()
        rc = lib.ls_trng_fill_steps(s2.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), s2.size, B, D, J, F, T, n, 0, eps.ctypes.data_as(_lib.c_f32p), nz.ctypes.data_as(_lib.c_f32p), v, nt)
        best = min(best, time.perf_counter() - t0)
    print(f"{label}: variant {v} threads {nt}: {best / n * 1e3:.2f} ms/step rc={rc} checksum {float(eps.sum()):.4f} {float(nz.sum()):.4f}", flush=True)
