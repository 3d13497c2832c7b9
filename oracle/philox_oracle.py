"""TEST INFRASTRUCTURE -- numpy restatement of the engine's device RNG (``livelyspeaker_amd/csrc/ls_philox.h``).

In throughput mode (``noise_source='philox'``) the engine replaces the reference's torch-CPU draws
(``scripts/diffusion/gaussian_diffusion.py:700-707`` x_T, ``:543`` / ``:787`` per-step ``randn_like(x)``,
``scripts/model/RAG.py:12`` the style ``randn_like`` of each CFG pass) by counter-based streams:

    Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 constants)
    counter = (e >> 2, 4 * step_id + stream, gidx_lo, gidx_hi),  key = (seed_lo, seed_hi)
    element e uses words (0,1) if e & 2 == 0 else (2,3):  u0 = (float(a) + 0.5) * 2^-32,  u1 likewise from b
    value = sqrt(-2 ln u0) * (cos(2 pi u1) if e & 1 == 0 else sin(2 pi u1))                      (Box-Muller)

with ``gidx`` the GLOBAL sample index (shard-invariant), ``stream`` 0 = x_T (step_id 0xFFFFFF), 1 / 2 = style eps of the
cond / uncond pass, 3 = the sampler's step noise, and ``step_id`` the executed-step counter k (0 = first executed step).
The reference has no such RNG: this file pins the ENGINE's streams so that a Philox-mode run can be replayed through the
CPU oracle (tests/test_gpu_philox.py).  The block function is pinned to Random123's published known-answer vectors in
tests/test_philox_oracle.py.  Only tests/, __graft_entry__.smoke() and bench.py's checker legs may import this.
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
X_T_STEP = 0xFFFFFF


def philox4x32_10(ctr: np.ndarray, key) -> np.ndarray:
    """ctr: [..., 4] uint32 counters; key: (k0, k1).  Returns [..., 4] uint32."""
    c = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack([x.astype(np.uint32) for x in c], axis=-1)


def normals(seed: int, gidx, step_id: int, stream: int, n: int) -> np.ndarray:
    """N(0,1) elements e = 0..n-1 of stream (step_id, stream) for every global sample index in ``gidx`` -> [len(gidx), n] fp32."""
    gidx = np.atleast_1d(np.asarray(gidx, dtype=np.uint64))
    nblk = (n + 3) // 4
    ctr = np.empty((len(gidx), nblk, 4), np.uint32)
    ctr[..., 0] = np.arange(nblk, dtype=np.uint32)[None, :]
    ctr[..., 1] = np.uint32((4 * step_id + stream) & 0xFFFFFFFF)
    ctr[..., 2] = (gidx & MASK).astype(np.uint32)[:, None]
    ctr[..., 3] = (gidx >> np.uint64(32)).astype(np.uint32)[:, None]
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    # the device forms u in fp32: float(a) is RNE to 24 bits, + 0.5f, * 2^-32
    u = (r.astype(np.float32) + np.float32(0.5)) * np.float32(2.3283064365386963e-10)
    u0, u1 = u[..., 0::2].astype(np.float64), u[..., 1::2].astype(np.float64)         # [G, nblk, 2 pairs]
    rad = np.sqrt(-2.0 * np.log(u0))
    out = np.stack([rad * np.cos(2 * np.pi * u1), rad * np.sin(2 * np.pi * u1)], axis=-1)   # [G, nblk, pair, (cos, sin)]
    return out.reshape(len(gidx), nblk * 4)[:, :n].astype(np.float32)


def x_init(seed: int, gidx, JF: int, T: int, shape_jf) -> np.ndarray:
    """x_T of ``ls_sample`` when no x_init is given: [G, J, F, T]; element index = flat index of [J*F][T]."""
    J, F = shape_jf
    return normals(seed, gidx, X_T_STEP, 0, JF * T).reshape(-1, J, F, T)


def step_tapes(seed: int, gidx, n_exec: int, shape_jft, D: int = 512):
    """The tapes the oracle's sample loop consumes, equal to what the kernels draw in PHILOX mode:
    eps [n_exec, 2, G, D] (cond, uncond) and noise [n_exec, G, J, F, T]."""
    J, F, T = shape_jft
    G = len(np.atleast_1d(gidx))
    eps = np.empty((n_exec, 2, G, D), np.float32)
    noise = np.empty((n_exec, G, J, F, T), np.float32)
    for k in range(n_exec):
        eps[k, 0] = normals(seed, gidx, k, 1, D)
        eps[k, 1] = normals(seed, gidx, k, 2, D)
        noise[k] = normals(seed, gidx, k, 3, J * F * T).reshape(G, J, F, T)
    return eps, noise
