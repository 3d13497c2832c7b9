import sys, time, ctypes, numpy as np, torch
sys.path.insert(0, ".")
from livelyspeaker_amd import _lib
_lib.use_library("variants/trngt.so")
from livelyspeaker_amd import torch_rng
lib = _lib.load_library()
v = torch_rng.variant()
B, D, J, F, T, n = 256, 512, 47, 6, 34, 8
for on in (0, 1):
    lib.ls_trng_set_jump(on)
    for nt in (16, 24):
        torch.manual_seed(1)
        st = torch.get_rng_state().numpy().copy()
        eps = np.empty((n, 2, B, D), np.float32); nz = np.empty((n, B, J, F, T), np.float32)
        for rep in range(3):
            s2 = st.copy()
            t0 = time.perf_counter()
            rc = lib.ls_trng_fill_steps(s2.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), s2.size, B, D, J, F, T, n, 0, eps.ctypes.data_as(_lib.c_f32p), nz.ctypes.data_as(_lib.c_f32p), v, nt)
            print(f"jump {on} threads {nt}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step", file=sys.stderr, flush=True)
