import sys, time, ctypes, numpy as np, torch
sys.path.insert(0, ".")
from livelyspeaker_amd import _lib, torch_rng
lib = _lib.load_library()
v = torch_rng.variant()
for (B, D, J, F, T, n), label in (((512, 512, 9, 3, 34, 12), "TED B=512 x12"), ((256, 512, 47, 6, 34, 4), "BEAT B=256 x4 steps per call"), ((256, 512, 47, 6, 34, 16), "BEAT B=256 x16")):
  for on in (0, 1):
    lib.ls_trng_set_jump(on)
    for nt in (12, 16, 24, 32, 48):
        torch.manual_seed(1)
        st = torch.get_rng_state().numpy().copy()
        eps = np.empty((n, 2, B, D), np.float32); nz = np.empty((n, B, J, F, T), np.float32)
        best = 1e9
        for rep in range(6):
            s2 = st.copy()
            t0 = time.perf_counter()
            rc = lib.ls_trng_fill_steps(s2.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), s2.size, B, D, J, F, T, n, 0, eps.ctypes.data_as(_lib.c_f32p), nz.ctypes.data_as(_lib.c_f32p), v, nt)
            best = min(best, time.perf_counter() - t0)
        print(f"{label}: jump {on} threads {nt}: {best / n * 1e3:.3f} ms/step rc={rc} checksum {float(eps.sum()):.4f} {float(nz.sum()):.4f}", flush=True)
