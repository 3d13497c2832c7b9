#!/usr/bin/env python3
"""Golden vectors for the BEAT caller plumbing (fixture G10), produced by EXECUTING the reference's own functions at generation
time: scripts_beat/dataloaders/rot_utils.py (rotation_6d_to_matrix, matrix_to_euler_angles) applied as
scripts_beat/test_RAG_beat.py:86 and :101 apply them, to a real BEAT sampler output (beat_golden.npz, G3)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/scripts_beat")
import numpy as np                      # noqa: E402
import torch                            # noqa: E402
from dataloaders import rot_utils       # noqa: E402
from oracle import rag_oracle as orc    # noqa: E402

sample = np.load(os.path.join(HERE, "beat_golden.npz"))["G3_ddpm50_final"]         # [4, 47, 6, 34]
B = sample.shape[0]
decoded_motions = torch.from_numpy(sample).permute(0, 3, 1, 2).reshape(B, 34, 47 * 6)                    # test_RAG_beat.py:86
pred_euler = rot_utils.matrix_to_euler_angles(rot_utils.rotation_6d_to_matrix(decoded_motions.reshape(-1, 34, 47, 6)), "XYZ").flatten(2) / (np.pi) * 180   # :101
o = orc.beat_post(sample)
d = float(np.abs(o["pred_euler"] - pred_euler.numpy()).max())
print("oracle vs reference: decoded equal", bool(np.array_equal(o["decoded_motions"], decoded_motions.numpy())), "euler max|d| (deg)", d,
      "|euler|max", float(pred_euler.abs().max()))
assert np.array_equal(o["decoded_motions"], decoded_motions.numpy()) and d < 5e-3
np.savez_compressed(os.path.join(HERE, "post_beat_golden.npz"), G10_decoded=decoded_motions.numpy(), G10_euler=pred_euler.numpy().astype(np.float32))
print("wrote post_beat_golden.npz")
