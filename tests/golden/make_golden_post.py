#!/usr/bin/env python3
"""Golden vectors for the caller-plumbing row (fixture G9), produced by EXECUTING the reference's own lines at
generation time (nothing of them is stored): scripts/test_RAG_ted.py:22-32 (constants), :88-106 (angle-change curve),
:108-111 (beat test) and utils.data_utils.convert_dir_vec_to_pose (imported with a stub `librosa`)."""
import math
import os
import sys
import textwrap
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/scripts")
sys.modules["librosa"] = types.ModuleType("librosa")
import numpy as np                      # noqa: E402
import torch                            # noqa: E402
import torch.nn.functional as F         # noqa: E402
from utils.data_utils import convert_dir_vec_to_pose   # noqa: E402
from livelyspeaker_amd import postprocess as pp         # noqa: E402
from oracle import rag_oracle as orc                    # noqa: E402

src = open("/root/reference/scripts/test_RAG_ted.py").read().splitlines()
env = {"np": np, "torch": torch, "F": F, "math": math}
exec("\n".join(src[21:33]), env)                        # mean_dir_vec, angle_pair, change_angle, thres, sigma
golden = np.load(os.path.join(HERE, "ted_golden.npz"))
sample = golden["G5_ddpm1000_final"]                    # a real sampler output [4,9,3,34]
B = sample.shape[0]
env.update(aligned_motions=torch.from_numpy(sample).permute(0, 3, 1, 2).reshape(B, 34, -1), batch_size=B)
exec(textwrap.dedent("\n".join(src[87:104])).replace(".cuda()", ""), env)     # beat_vec ... angle_diff (with the leading zero)
angle_diff = env["angle_diff"]
mask = np.zeros((B, 34), dtype=bool)
beat_lines = textwrap.dedent("\n".join(src[108:111]))   # the `if ... if ... append` test for one (b, t)
for b in range(B):
    motion_beat_time = []
    for t in range(2, 33):
        exec(beat_lines, dict(env, b=b, t=t, motion_beat_time=motion_beat_time, angle_diff=angle_diff))
    for bt in motion_beat_time:
        mask[b, int(round(bt * 15.0))] = True
vec = env["aligned_motions"].numpy() + np.asarray(env["mean_dir_vec"], dtype=np.float32)
pose = convert_dir_vec_to_pose(vec.reshape(B, 34, 9, 3))
assert np.allclose(np.asarray(env["mean_dir_vec"], np.float32), pp.TED_MEAN_DIR_VEC) and env["angle_pair"] == pp.TED_ANGLE_PAIRS
assert env["change_angle"] == pp.TED_CHANGE_ANGLE and env["thres"] == pp.TED_BEAT_THRES
o = orc.ted_post(sample, pp.TED_MEAN_DIR_VEC, pp.TED_ANGLE_PAIRS, pp.TED_CHANGE_ANGLE, pp.TED_BEAT_THRES, pp.TED_DIR_VEC_PAIRS)
print("oracle vs reference: angle_diff", float(np.abs(o["angle_diff"] - angle_diff.numpy()).max()), "pose",
      float(np.abs(o["pose"] - pose).max()), "beats equal", bool((o["beat_mask"] == mask).all()), "n beats", int(mask.sum()))
assert np.abs(o["angle_diff"] - angle_diff.numpy()).max() < 1e-3 and np.abs(o["pose"] - pose).max() < 1e-6
np.savez_compressed(os.path.join(HERE, "post_golden.npz"), G9_angle_diff=angle_diff.numpy(), G9_beat_mask=mask,
                    G9_pose=pose.astype(np.float32), G9_aligned=env["aligned_motions"].numpy())
print("wrote post_golden.npz")
