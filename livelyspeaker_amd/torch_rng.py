"""torch's CPU normal stream drawn natively (``csrc/ls_torch_rng.cpp`` behind ``ls_trng_*``): the host draws of the "identical
seeds" mode (``GaussianDiffusion.noise_source = 'torch_cpu'``) without a Python-level torch call per tensor.

The reference's contract is torch's *global CPU generator*: ``torch.manual_seed(s)`` fixes every normal of a sampling loop
(scripts/diffusion/gaussian_diffusion.py:700-743, scripts/model/RAG.py:10-13, 120).  The native code continues exactly that stream:
it takes ``torch.get_rng_state()``, makes the draws torch would make (mt19937 words, the 16-block float Box-Muller of contiguous
tensors, the per-element double Box-Muller -- with its cached second sample -- of everything else), and hands the advanced state back
through ``torch.set_rng_state()``.  How torch's *float* transform rounds depends on how torch was compiled (libm in the DEFAULT
kernel, Cephes polynomials with compiler-contracted FMAs in the AVX2 / AVX512 ones), so ``variant()`` looks once per process for the
restated variant that reproduces this torch build bit for bit; if none does, callers keep using torch's own generator.
"""
from __future__ import annotations

import os

import torch as th

from . import _lib

_VARIANT = None         # None: not probed yet; -1: no variant reproduces this torch (or no library); 0..4: the one that does


def n_threads() -> int:
    """Transform workers of a draw call (the word generators of long fills come on top: 6 behind the sequential scout, 12 with jump-ahead).
    LS_TRNG_THREADS overrides (tools / A-B runs)."""
    env = os.environ.get("LS_TRNG_THREADS")
    if env:
        return max(1, int(env))
    return max(1, min(16, (os.cpu_count() or 1)))      # measured on the 256-thread GPU hosts, rounds 4 and 6: 16 workers; 24 / 32 are slower inside the sampling loop


def fill_steps(eps: th.Tensor, noise: th.Tensor, first_contiguous: bool, variant_: int) -> None:
    """The per-step draws of ``eps.shape[0]`` sampling steps, in the reference's order, from torch's global CPU generator:
    eps [n, 2, B, D] (randn(B,1,D) of the cond then the uncond pass), noise [n, B, J, F, T] (randn_like(x); x has the memory order
    [T][B][J][F] of the model output except at a first step with a contiguous x).  Advances torch's generator accordingly."""
    assert eps.is_contiguous() and noise.is_contiguous() and eps.dtype == noise.dtype == th.float32 and not eps.is_cuda
    n, two, B, D = eps.shape
    assert two == 2 and noise.shape[0] == n and noise.shape[1] == B
    _, _, J, F, T = noise.shape
    lib = _lib.load_library()
    if os.environ.get("LS_TRNG_JUMP") is not None:      # A/B switch: 0 = the sequential scout (default), 1 = mt19937 jump-ahead (csrc/ls_mt_jump.h)
        lib.ls_trng_set_jump(int(os.environ["LS_TRNG_JUMP"]))
    st = th.get_rng_state()
    rc = lib.ls_trng_fill_steps(st.data_ptr(), st.numel(), B, D, J, F, T, n, int(bool(first_contiguous)), eps.data_ptr(), noise.data_ptr(),
                                int(variant_), n_threads())
    if rc != 0:
        raise _lib.EngineError(f"ls_trng_fill_steps failed ({rc})")
    th.set_rng_state(st)


def _torch_steps(n, B, D, J, F, T, first_contiguous):
    eps, nz = th.empty(n, 2, B, D), th.empty(n, B, J, F, T)
    later = th.empty(T, B, J, F).permute(1, 2, 3, 0)
    for k in range(n):
        eps[k, 0] = th.randn(B, 1, D)[:, 0]
        eps[k, 1] = th.randn(B, 1, D)[:, 0]
        nz[k].copy_(th.randn_like(th.empty(B, J, F, T) if (k == 0 and first_contiguous) else later))
    return eps, nz


def _probe() -> int:
    try:
        _lib.load_library()
    except Exception:           # noqa: BLE001  (no library: nothing to probe; the engine itself will complain where it matters)
        return -1
    keep = th.get_rng_state()
    try:
        # the CPU generator only: th.manual_seed would also reseed every CUDA generator, i.e. silently reset the caller's device-side
        # RNG streams the first time this mode is used in a process
        th.default_generator.manual_seed(0x5EED)
        th.randn(3)                                     # leave a cached double sample behind: the hand-over is part of the contract
        s0 = th.get_rng_state()
        shape = dict(n=3, B=3, D=512, J=3, F=3, T=5)    # odd element counts: the cached sample crosses step boundaries
        want_e, want_n = _torch_steps(first_contiguous=True, **shape)
        want_after = th.randn(7, dtype=th.float64)
        for v in (1, 0, 2, 3, 4):
            th.set_rng_state(s0)
            eps, nz = th.empty_like(want_e), th.empty_like(want_n)
            try:
                fill_steps(eps, nz, True, v)
            except _lib.EngineError:
                continue
            if th.equal(eps, want_e) and th.equal(nz, want_n) and th.equal(th.randn(7, dtype=th.float64), want_after):
                return v
        return -1
    finally:
        th.set_rng_state(keep)


def variant() -> int:
    """The restated float transform that reproduces this torch build (see the module docstring), or -1."""
    global _VARIANT
    if _VARIANT is None:
        _VARIANT = _probe()
    return _VARIANT
