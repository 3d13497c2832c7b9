"""Caller plumbing row: layout change, mean add, bone normalisation, joint-angle change
curve, motion beats, dir-vec -> pose.  Oracle vs fixture G9 (CPU); HIP kernel vs fixture and oracle (GPU)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, max_abs
from livelyspeaker_amd import postprocess as pp


@pytest.fixture(scope="module")
def g9():
    return np.load(os.path.join(GOLDEN, "post_golden.npz"))


def _sample():
    return np.load(os.path.join(GOLDEN, "ted_golden.npz"))["G5_ddpm1000_final"]


def test_oracle_matches_reference_fixture(g9):
    from oracle import rag_oracle as orc
    o = orc.ted_post(_sample(), pp.TED_MEAN_DIR_VEC, pp.TED_ANGLE_PAIRS, pp.TED_CHANGE_ANGLE, pp.TED_BEAT_THRES, pp.TED_DIR_VEC_PAIRS)
    assert np.array_equal(o["aligned"], g9["G9_aligned"])
    assert max_abs(o["angle_diff"], g9["G9_angle_diff"]) < 1e-3       # acos near +-1 amplifies fp32 rounding
    assert max_abs(o["pose"], g9["G9_pose"]) < 1e-6
    assert np.array_equal(o["beat_mask"], g9["G9_beat_mask"]) and g9["G9_beat_mask"].sum() > 0


@pytest.mark.gpu
def test_hip_post_vs_reference_fixture(g9):
    import torch
    s = _sample()
    r = pp.ted_postprocess(s)
    assert np.array_equal(r["aligned_motions"], g9["G9_aligned"])
    assert max_abs(r["pose"], g9["G9_pose"]) < 1e-5
    d = max_abs(r["angle_diff"], g9["G9_angle_diff"])
    print("angle_diff max|d| =", d)
    assert d < 2e-3
    # beats are a thresholded local-minimum test: equal wherever the curve is not within rounding of the threshold
    assert (r["beat_mask"] != g9["G9_beat_mask"]).sum() <= 1
    assert r["motion_beat_times"][0] == [float(t) / 15.0 for t in np.nonzero(r["beat_mask"][0])[0]]
    # device-resident path (what the sampler returns) and a caller-sized batch
    big = torch.from_numpy(np.tile(s, (128, 1, 1, 1))).cuda()
    rb = pp.ted_postprocess(big)
    assert rb["aligned_motions"].is_cuda and tuple(rb["pose"].shape) == (512, 34, 10, 3)
    assert np.array_equal(rb["angle_diff"][:4].cpu().numpy(), r["angle_diff"])
    assert np.array_equal(rb["beat_mask"][508:].cpu().numpy(), r["beat_mask"])


def test_beat_consistency_accumulation():
    """BC score accumulation of the evaluation loop (test_RAG_ted.py:113-127), restated here in its literal loop form."""
    import math
    from livelyspeaker_amd.postprocess import BeatConsistency
    rng = np.random.Generator(np.random.PCG64(5))
    motion = [sorted((rng.integers(2, 33, size=n) / 15.0).tolist()) for n in (3, 0, 1, 6)]
    audio = [np.sort(rng.uniform(0, 2.2, size=n)) for n in (4, 5, 0, 7)]
    bc = BeatConsistency()
    bc.push(motion[:2], audio[:2])
    bc.push(motion[2:], audio[2:])
    sigma, total, nb, nm = 0.1, 0.0, 0, 0
    for mb, ab in zip(motion, audio):
        nm += len(mb)
        if len(mb) == 0:
            continue
        s = 0
        for a in ab:
            s += np.power(math.e, -np.min(np.power((a - np.asarray(mb)), 2)) / (2 * sigma * sigma))
        total += s
        nb += len(ab)
    assert bc.motion_beats_sum == nm and bc.num_beats == nb
    assert abs(bc.score() - total / nb) < 1e-12
    with pytest.raises(ValueError):
        bc.push(motion[:1], audio[:2])


# ---- BEAT twin: layout change + rot6d -> matrix -> Euler XYZ in degrees (scripts_beat/test_RAG_beat.py:86, 101) --------------------
@pytest.fixture(scope="module")
def g10():
    return np.load(os.path.join(GOLDEN, "post_beat_golden.npz"))


def _beat_sample():
    return np.load(os.path.join(GOLDEN, "beat_golden.npz"))["G3_ddpm50_final"]


def _circ_deg(a, b):
    """largest angular distance in degrees (atan2 wraps at +-180)"""
    return float(np.abs((np.asarray(a, np.float64) - np.asarray(b, np.float64) + 180.0) % 360.0 - 180.0).max())


def test_beat_oracle_matches_reference_fixture(g10):
    from oracle import rag_oracle as orc
    o = orc.beat_post(_beat_sample())
    assert np.array_equal(o["decoded_motions"], g10["G10_decoded"])
    assert _circ_deg(o["pred_euler"], g10["G10_euler"]) < 5e-3         # asin / atan2 near their singular points amplify fp32 rounding


@pytest.mark.gpu
def test_hip_beat_post_vs_reference_fixture(g10):
    import torch
    s = _beat_sample()
    r = pp.beat_postprocess(s)
    assert np.array_equal(r["decoded_motions"], g10["G10_decoded"])
    d = _circ_deg(r["pred_euler"], g10["G10_euler"])
    print("BEAT euler max circular |d| (deg) =", d)
    assert d < 5e-3
    big = torch.from_numpy(np.tile(s, (64, 1, 1, 1))).cuda()            # the callers' batch (256), device-resident
    rb = pp.beat_postprocess(big)
    assert rb["pred_euler"].is_cuda and tuple(rb["pred_euler"].shape) == (256, 34, 141)
    assert np.array_equal(rb["pred_euler"][252:].cpu().numpy(), r["pred_euler"])
    assert np.array_equal(rb["decoded_motions"][:4].cpu().numpy(), r["decoded_motions"])
    with pytest.raises(ValueError):
        pp.beat_postprocess(np.zeros((2, 47, 3, 34), np.float32))
