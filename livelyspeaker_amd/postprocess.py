"""Caller-side plumbing around the sampler: what ``scripts/test_RAG_ted.py:84-111`` and
``scripts/utils/data_utils.py:77-97`` do with a sampled batch, as one small GPU kernel (``ls_ted_post``).

The numbers below are DATASET CONSTANTS of the TED gesture data as published in the reference's scripts
(mean direction vector ``test_RAG_ted.py:22``, angle pairs ``:24-29``, per-pair normalisers ``:30``, beat
threshold ``:32``, bone tree and lengths ``utils/data_utils.py:13-14``); they are data, passed to the kernel."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

TED_MEAN_DIR_VEC = np.array([0.0154009, -0.9690125, -0.0884354, -0.0022264, -0.8655276, 0.4342174, -0.0035145, -0.8755367,
                             -0.4121039, -0.9236511, 0.3061306, -0.0012415, -0.5155854, 0.8129665, 0.0871897, 0.2348464,
                             0.1846561, 0.8091402, 0.9271948, 0.2960011, -0.013189, 0.5233978, 0.8092403, 0.0725451,
                             -0.2037076, 0.1924306, 0.8196916], dtype=np.float32)
TED_ANGLE_PAIRS = [(3, 4), (4, 5), (6, 7), (7, 8)]
TED_CHANGE_ANGLE = [0.0034540758933871984, 0.007043459918349981, 0.003493624273687601, 0.007205077446997166]
TED_BEAT_THRES = 0.03
TED_DIR_VEC_PAIRS = [(0, 1, 0.26), (1, 2, 0.18), (2, 3, 0.14), (1, 4, 0.22), (4, 5, 0.36), (5, 6, 0.33), (1, 7, 0.22),
                     (7, 8, 0.36), (8, 9, 0.33)]
TED_FPS = 15.0
TED_BEAT_SIGMA = 0.1          # test_RAG_ted.py:33


def ted_post_config() -> "_lib.LsPostConfig":
    c = _lib.LsPostConfig()
    c.njoints, c.n_pairs, c.n_pose_joints, c.thres = 9, len(TED_ANGLE_PAIRS), 10, TED_BEAT_THRES
    for k, (a, b) in enumerate(TED_ANGLE_PAIRS):
        c.pair_a[k], c.pair_b[k], c.change_angle[k] = a, b, TED_CHANGE_ANGLE[k]
    for j, (pa, ch, ln) in enumerate(TED_DIR_VEC_PAIRS):
        c.bone_parent[j], c.bone_child[j], c.bone_len[j] = pa, ch, ln
    for j, v in enumerate(TED_MEAN_DIR_VEC):
        c.mean_dir_vec[j] = float(v)
    return c


def ted_postprocess(sample, device: int = 0, want_pose: bool = True) -> dict:
    """sample: [B, 9, 3, 34] (numpy, CPU tensor or CUDA tensor as returned by the sampler).
    Returns aligned_motions [B,34,27], pose [B,34,10,3], angle_diff [B,34], beat_mask [B,34] (bool) and
    motion_beat_times (list of lists of seconds, t/15 as in test_RAG_ted.py:111)."""
    lib = _lib.load_library()
    m = _lib._Marshal(device, sample)
    B = int(sample.shape[0])
    cfg = ted_post_config()
    aligned, p_al = m.out((B, 34, 27))
    pose, p_pose = m.out((B, 34, 10, 3)) if want_pose else (None, None)
    diff, p_diff = m.out((B, 34))
    if m.on_device:
        mask = m.torch.empty((B, 34), dtype=m.torch.uint8, device=m.dev)
        p_mask = C.c_void_p(mask.data_ptr())
    else:
        mask = np.empty((B, 34), np.uint8)
        p_mask = mask.ctypes.data_as(C.c_void_p)
    rc = lib.ls_ted_post(device, int(m.on_device), B, C.byref(cfg), m.f32(sample, (B, 9, 3, 34)), p_al, p_pose, p_diff, p_mask)
    if rc != 0:
        raise _lib.EngineError(f"ls_ted_post failed ({rc})")
    mask_np = mask.cpu().numpy() if m.on_device else mask
    beats = [[float(t) / TED_FPS for t in np.nonzero(mask_np[b])[0]] for b in range(B)]
    return {"aligned_motions": aligned, "pose": pose, "angle_diff": diff,
            "beat_mask": mask.bool() if m.on_device else mask.astype(bool), "motion_beat_times": beats}


def beat_postprocess(sample, device: int = 0, want_euler: bool = True) -> dict:
    """BEAT caller plumbing (scripts_beat/test_RAG_beat.py:86, 101).  sample: [B, 47, 6, 34] (numpy, CPU tensor or CUDA tensor as
    returned by the sampler).  Returns decoded_motions [B, 34, 282] (the rot6d pose sequence the evaluator and the metrics
    consume) and pred_euler [B, 34, 141]: Euler XYZ angles in degrees of every joint (rot_utils.matrix_to_euler_angles(
    rot_utils.rotation_6d_to_matrix(.), "XYZ") / pi * 180).  Audio onsets / the alignment score stay with the caller."""
    lib = _lib.load_library()
    m = _lib._Marshal(device, sample)
    B, J = int(sample.shape[0]), int(sample.shape[1])
    if tuple(sample.shape[2:]) != (6, 34):
        raise ValueError(f"expected [B, J, 6, 34], got {tuple(sample.shape)}")
    dec, p_dec = m.out((B, 34, J * 6))
    eul, p_eul = m.out((B, 34, J * 3)) if want_euler else (None, None)
    rc = lib.ls_beat_post(device, int(m.on_device), B, J, m.f32(sample, (B, J, 6, 34)), p_dec, p_eul)
    if rc != 0:
        raise _lib.EngineError(f"ls_beat_post failed ({rc})")
    return {"decoded_motions": dec, "pred_euler": eul}


class BeatConsistency:
    """Running beat-alignment (BC) score over clips, as the evaluation loop accumulates it (test_RAG_ted.py:113-127): for every
    audio onset, exp(-min_m (onset - m)^2 / (2 sigma^2)) over the clip's motion beats; clips without a motion beat contribute
    nothing (not even their onsets).  Audio onset times are an INPUT here: the reference gets them from
    librosa.onset.onset_detect, a third-party package that is not part of this path."""

    def __init__(self, sigma: float = TED_BEAT_SIGMA):
        self.sigma = float(sigma)
        self.align_sum = 0.0
        self.num_beats = 0
        self.motion_beats_sum = 0

    def push(self, motion_beat_times, audio_beat_times):
        """Both arguments: one sequence of times (seconds) per clip."""
        if len(motion_beat_times) != len(audio_beat_times):
            raise ValueError("one list of motion beats and one list of audio onsets per clip")
        for mb, ab in zip(motion_beat_times, audio_beat_times):
            self.motion_beats_sum += len(mb)
            if len(mb) == 0:
                continue
            mb = np.asarray(mb, np.float64)
            for a in np.asarray(ab, np.float64).reshape(-1):
                self.align_sum += float(np.exp(-np.min((a - mb) ** 2) / (2.0 * self.sigma * self.sigma)))
            self.num_beats += len(ab)

    def score(self) -> float:
        return self.align_sum / self.num_beats
