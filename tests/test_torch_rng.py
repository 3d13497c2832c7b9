"""CPU: torch's CPU normal stream restated natively (csrc/ls_torch_rng.cpp, livelyspeaker_amd/torch_rng.py) -- the "identical seeds"
contract (torch.manual_seed fixes every draw of the reference's loop: gaussian_diffusion.py:700-743, RAG.py:10-13, 120) without
torch's generator on the critical path.  Bitwise against torch itself: values, draw order, and the generator state left behind."""
import ctypes as C

import numpy as np
import pytest
import torch

from livelyspeaker_amd import _lib, torch_rng


def _native_randn(n, variant):
    lib = _lib.load_library()
    st = torch.get_rng_state()
    out = torch.empty(n)
    assert lib.ls_trng_randn(st.data_ptr(), st.numel(), out.data_ptr(), n, variant, 2) == 0
    torch.set_rng_state(st)
    return out


def test_state_blob_layout_is_the_one_the_native_code_assumes():
    torch.manual_seed(233)
    s = torch.get_rng_state().numpy()
    assert s.size == 5056
    u32 = s.view(np.uint32)
    assert (u32[0], u32[2], u32[3], u32[4]) == (233, 1, 1, 0)                  # seed | left = 1 | seeded | next = 0
    torch.randn(3, dtype=torch.float64)                                         # two pairs: one sample stays cached
    s = torch.get_rng_state().numpy()
    assert s.view(np.uint32)[2] == 624 - 7 and s[5024:5032].view(np.float64)[0] != 0.0 and s[5040:5044].view(np.int32)[0] == 1


def test_a_variant_reproduces_this_torch_build():
    v = torch_rng.variant()
    print("torch CPU capability:", torch.backends.cpu.get_cpu_capability(), "-> native float transform variant", v)
    assert v >= 0, "no restated variant reproduces torch's contiguous float normals on this machine (callers fall back to torch)"


@pytest.mark.parametrize("n", [1, 2, 3, 15, 16, 17, 31, 32, 33, 100, 2048, 4099, 1 << 17, (1 << 21) + 16, 3000003])
def test_randn_equals_torch_for_every_size_class(n):
    v = torch_rng.variant()
    assert v >= 0
    torch.manual_seed(7 + n)
    want, want_after = torch.randn(n), torch.randn(5, dtype=torch.float64)
    torch.manual_seed(7 + n)
    got = _native_randn(n, v)
    assert torch.equal(got, want) and torch.equal(torch.randn(5, dtype=torch.float64), want_after)      # same generator state afterwards


@pytest.mark.parametrize("B,first", [(1, True), (3, True), (3, False), (4, False), (6, True)])
def test_step_draws_equal_torchs_in_order_value_and_final_state(B, first):
    v = torch_rng.variant()
    assert v >= 0
    shape = dict(n=5, B=B, D=512, J=9, F=3, T=34)
    torch.manual_seed(99)
    torch.randn(3)                                          # a cached double sample waiting in the generator
    s0 = torch.get_rng_state()
    want_e, want_n = torch_rng._torch_steps(first_contiguous=first, **shape)
    want_after = torch.randn(4, dtype=torch.float64)
    torch.set_rng_state(s0)
    eps, nz = torch.empty_like(want_e), torch.empty_like(want_n)
    torch_rng.fill_steps(eps, nz, first, v)
    assert torch.equal(eps, want_e) and torch.equal(nz, want_n)
    assert torch.equal(torch.randn(4, dtype=torch.float64), want_after)
    # and in two pieces: the second piece starts where the first one left the generator (segmented tapes)
    torch.set_rng_state(s0)
    torch_rng.fill_steps(eps[:2], nz[:2], first, v)
    torch_rng.fill_steps(eps[2:], nz[2:], False, v)
    assert torch.equal(eps, want_e) and torch.equal(nz, want_n)


@pytest.mark.parametrize("threads", [1, 5])
def test_step_draws_at_a_size_that_is_dealt_to_the_worker_pool(threads, monkeypatch):
    """B = 201: every draw is split into several pool tasks (contiguous draws by 16-blocks, the memory-order draw by destination range,
    pairs straddling the ranges), with and without worker threads; odd element counts hand the cached sample from step to step."""
    v = torch_rng.variant()
    assert v >= 0
    monkeypatch.setattr(torch_rng, "n_threads", lambda: threads)
    shape = dict(n=3, B=201, D=512, J=9, F=3, T=33)        # B*J*F*T odd per step
    torch.manual_seed(1234)
    torch.randn(1)
    s0 = torch.get_rng_state()
    want_e, want_n = torch_rng._torch_steps(first_contiguous=False, **shape)
    want_after = torch.randn(4, dtype=torch.float64)
    torch.set_rng_state(s0)
    eps, nz = torch.empty_like(want_e), torch.empty_like(want_n)
    torch_rng.fill_steps(eps, nz, False, v)
    assert torch.equal(eps, want_e) and torch.equal(nz, want_n)
    assert torch.equal(torch.randn(4, dtype=torch.float64), want_after)


@pytest.mark.parametrize("threads", [2, 16])
def test_long_fills_dealt_to_generator_threads_stay_the_same_stream(threads, monkeypatch):
    """Fills of more than ~160 K words are produced by generator threads from state snapshots the calling thread takes while it runs the
    recurrence alone ahead of them (csrc/ls_torch_rng.cpp, Mt::fill_words): BEAT-like steps (1.2 M words of memory-order draw each, odd
    counts, a cached sample on entry), values and the generator state afterwards as torch's."""
    v = torch_rng.variant()
    assert v >= 0
    monkeypatch.setattr(torch_rng, "n_threads", lambda: threads)
    shape = dict(n=3, B=65, D=512, J=47, F=6, T=33)         # 604 890 elements per memory-order draw (odd pairs), contiguous draws of 33 280
    torch.manual_seed(4321)
    torch.randn(3)
    s0 = torch.get_rng_state()
    want_e, want_n = torch_rng._torch_steps(first_contiguous=True, **shape)
    want_after = torch.randn(6, dtype=torch.float64)
    torch.set_rng_state(s0)
    eps, nz = torch.empty_like(want_e), torch.empty_like(want_n)
    torch_rng.fill_steps(eps, nz, True, v)
    assert torch.equal(eps, want_e) and torch.equal(nz, want_n)
    assert torch.equal(torch.randn(6, dtype=torch.float64), want_after)


def test_bad_arguments_are_rejected():
    lib = _lib.load_library()
    st = torch.get_rng_state()
    out = torch.empty(32)
    assert lib.ls_trng_randn(st.data_ptr(), 100, out.data_ptr(), 32, 0, 1) < 0             # not a torch state blob
    assert lib.ls_trng_randn(st.data_ptr(), st.numel(), out.data_ptr(), 32, 9, 1) < 0      # unknown variant
    bad = st.clone()
    bad[12] = 0                                                                            # "not seeded"
    assert lib.ls_trng_randn(bad.data_ptr(), bad.numel(), out.data_ptr(), 32, 0, 1) < 0


@pytest.mark.parametrize("isa", [0, 1, 2])
def test_vectorised_double_pairs_stay_inside_their_margin_and_defer_to_libm_at_the_edge(isa):
    """The per-element double path stores floats.  Its vectorised evaluation (ls_torch_rng.cpp, LS_FAST_PAIRS_BODY) is used for a sample
    only when every double within 2^-44 r of it rounds to the same float; the contract behind that guard is |fast - libm| <= 2^-48 r.
    Measured here against libm's own doubles over 2^20 random pairs + the corner words, per ISA clone: the error bound, that every
    sample it keeps is libm's float, and that the degenerate pairs (r = 0: u2 = 0) are handed back."""
    lib = _lib.load_library()
    rng = np.random.default_rng(7 + isa)
    npairs = 1 << 20
    w = rng.integers(0, 1 << 32, size=4 * npairs, dtype=np.uint32)
    w[0:4] = 0                                                # u1 = u2 = 0: r = 0
    w[4:8] = 0xFFFFFFFF                                       # u1, u2 = 1 - 2^-53: theta next to 2 pi, the largest r
    w[8:12] = [0, 1, 0, 1]                                    # the smallest non-zero uniforms
    w[12:16] = [0x80000, 0, 0x1FFFFF, 0xFFFFFFFF]             # theta = pi / 2 ... (a zero of the cosine)
    fc, fs, lc, ls = (np.empty(npairs) for _ in range(4))
    zc, zs = np.empty(npairs, np.float32), np.empty(npairs, np.float32)
    redo = np.empty(npairs, np.uint8)
    rc = lib.ls_trng_pairs_debug(w.ctypes.data, npairs, isa, fc.ctypes.data, fs.ctypes.data, zc.ctypes.data, zs.ctypes.data, redo.ctypes.data,
                                 lc.ctypes.data, ls.ctypes.data)
    if rc == -5:                                              # LS_EUNSUPPORTED
        pytest.skip("this machine lacks the ISA of that clone")
    assert rc == 0
    r = np.hypot(lc, ls)
    live = r > 0
    err = np.maximum(np.abs(fc - lc), np.abs(fs - ls))[live] / r[live]
    print(f"isa {isa}: max |fast - libm| / r = 2^{np.log2(err.max()):.2f}, sent back to libm: {redo.mean():.2e}")
    assert err.max() < 2.0 ** -48
    assert redo[0] == 1 and redo[~live].all()
    keep = redo == 0
    assert np.array_equal(zc[keep], lc[keep].astype(np.float32)) and np.array_equal(zs[keep], ls[keep].astype(np.float32))
    assert 0 < redo.mean() < 1e-3                             # the guard is exercised, and rare


def test_pair_statistics_count_what_went_back_to_libm():
    lib = _lib.load_library()
    a, b, a2, b2 = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert lib.ls_trng_stats(C.byref(a), C.byref(b)) == 0
    torch.manual_seed(5)
    eps, nz = torch.empty(1, 2, 16, 16), torch.empty(1, 16, 250, 1, 34)       # the noise of a later step: the per-element double path
    torch_rng.fill_steps(eps, nz, False, max(torch_rng.variant(), 0))
    assert lib.ls_trng_stats(C.byref(a2), C.byref(b2)) == 0
    assert a2.value - a.value >= 16 * 250 * 34 // 2 and 0 <= b2.value - b.value < 100
    torch.manual_seed(5)
    torch.randn(16, 1, 16), torch.randn(16, 1, 16)
    assert torch.equal(nz[0], torch.randn_like(torch.empty(34, 16, 250, 1).permute(1, 2, 3, 0)))


def test_jump_ahead_reaches_the_same_state_as_the_recurrence():
    """ls_mt_jump.h: t^J modulo the characteristic polynomial (Berlekamp-Massey, degree 19937) applied as an XOR of windows of a 33-block
    expansion of the state == J steps of the recurrence, for piece lengths from one block to 10^5 blocks, every ISA clone."""
    import ctypes as C
    lib = _lib.load_library()
    lib.ls_trng_jump_check.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(C.c_int)]
    for seed, blocks in ((5489, 1), (5489, 33), (1, 64), (233, 525), (7, 8400), (99, 100003)):
        n = C.c_int()
        assert lib.ls_trng_jump_check(seed, 624 * blocks, C.byref(n)) == 0, (seed, blocks)
        assert (n.value == 1) if blocks < 32 else (n.value > 100 if blocks < 500 else 8000 < n.value < 12000), (blocks, n.value)      # t^J itself below the degree, ~half of the 19937 coefficients far above it
    assert lib.ls_trng_jump_check(1, 100, None) < 0                                                 # not a whole number of blocks


@pytest.mark.parametrize("B,J,F", [(256, 47, 6), (512, 9, 3)])
def test_long_fills_are_identical_with_and_without_jump_ahead(B, J, F):
    """The per-step draws of a sampling loop (BEAT B = 256: 4.9 M words per noise tensor; TED B = 512) with the generator threads started
    from jumped states vs behind the sequential scout: every float and the generator state handed back, bit for bit -- and torch's own."""
    import ctypes as C
    lib = _lib.load_library()
    v = torch_rng.variant()
    if v < 0:
        pytest.skip("the native stream does not reproduce torch on this machine")
    D, T, n = 512, 34, 3
    outs = []
    was = lib.ls_trng_set_jump(1)
    try:
        for on in (1, 0):
            lib.ls_trng_set_jump(on)
            torch.manual_seed(77)
            st = torch.get_rng_state().numpy().copy()
            eps, nz = np.empty((n, 2, B, D), np.float32), np.empty((n, B, J, F, T), np.float32)
            rc = lib.ls_trng_fill_steps(st.ctypes.data_as(C.POINTER(C.c_uint8)), st.size, B, D, J, F, T, n, 0, eps.ctypes.data_as(_lib.c_f32p),
                                        nz.ctypes.data_as(_lib.c_f32p), v, 8)
            assert rc == 0
            outs.append((eps, nz, st))
    finally:
        lib.ls_trng_set_jump(was)
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    # torch itself, first step: randn(B,1,D) x 2, then randn_like of the [T][B][J][F]-ordered view
    torch.manual_seed(77)
    e0, e1 = torch.randn(B, 1, D), torch.randn(B, 1, D)
    x = torch.empty(T, B, J, F).permute(1, 2, 3, 0)
    z = torch.randn_like(x)
    assert np.array_equal(outs[0][0][0, 0], e0.numpy().reshape(B, D)) and np.array_equal(outs[0][0][0, 1], e1.numpy().reshape(B, D))
    assert np.array_equal(outs[0][1][0], z.numpy())
