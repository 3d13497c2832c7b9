"""FGD evaluator: oracle vs the fixtures generated from the reference (CPU) and the HIP pose encoder +
EmbeddingSpaceEvaluator drop-in vs both (GPU).  Tolerances: features 1e-4 relative to max|feature| (fp32, BatchNorm folded
into the weights on our side); FGD / distances 1e-3 relative (they amplify feature noise through a covariance square root)."""
import os

import numpy as np
import pytest

from livelyspeaker_amd import synth
from oracle import eval_oracle as evo

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_eval_oracle_matches_reference_fixture():
    g = np.load(os.path.join(GOLD, "eval_ted_golden.npz"))
    gen, real = synth.make_pose_sets(640)
    sd = synth.make_embedding_net_state_dict(27, 32)
    scale = float(np.abs(g["ted_gen_feat"]).max())
    assert np.abs(evo.pose_encoder(sd, gen) - g["ted_gen_feat"]).max() < 1e-5 * scale
    assert np.abs(evo.pose_encoder(sd, real) - g["ted_real_feat"]).max() < 1e-5 * scale
    fd, fe = evo.scores(g["ted_gen_feat"], g["ted_real_feat"])
    assert abs(fd - float(g["ted_frechet"])) < 1e-9 * abs(fd) and abs(fe - float(g["ted_feat_dist"])) < 1e-9
    gb = np.load(os.path.join(GOLD, "eval_beat_golden.npz"))
    sdb = synth.make_embedding_net_state_dict(141, 48, seed=synth.SEED_WEIGHTS + 201, hidden=(4, 2))
    genb, _ = synth.make_pose_sets(96, 141, seed=synth.SEED_COND + 3001)
    assert np.abs(evo.pose_encoder(sdb, genb) - gb["beat_feat"]).max() < 1e-5 * float(np.abs(gb["beat_feat"]).max())


@pytest.mark.gpu
def test_hip_pose_encoder_matches_fixture_ted_and_beat():
    import torch
    from livelyspeaker_amd import _lib
    g = np.load(os.path.join(GOLD, "eval_ted_golden.npz"))
    gen, real = synth.make_pose_sets(640)
    eng = _lib.EvalEngine(27, 34, 32, (256, 128))
    eng.load_state_dict(synth.make_embedding_net_state_dict(27, 32))
    scale = float(np.abs(g["ted_gen_feat"]).max())
    f = eng.features(gen)
    assert np.abs(f - g["ted_gen_feat"]).max() < 1e-4 * scale
    fd = eng.features(torch.from_numpy(real).cuda())                    # device-resident input -> device output
    assert fd.is_cuda and np.abs(fd.cpu().numpy() - g["ted_real_feat"]).max() < 1e-4 * scale
    gb = np.load(os.path.join(GOLD, "eval_beat_golden.npz"))
    engb = _lib.EvalEngine(141, 34, 48, (192, 96))
    engb.load_state_dict(synth.make_embedding_net_state_dict(141, 48, seed=synth.SEED_WEIGHTS + 201, hidden=(4, 2)))
    genb, _ = synth.make_pose_sets(96, 141, seed=synth.SEED_COND + 3001)
    assert np.abs(engb.features(genb) - gb["beat_feat"]).max() < 1e-4 * float(np.abs(gb["beat_feat"]).max())
    with pytest.raises(_lib.EngineError):
        _lib.EvalEngine(27, 34, 32, (256, 128)).features(gen)            # weights not committed


@pytest.mark.gpu
def test_embedding_space_evaluator_dropin_reproduces_reference_scores():
    import torch
    from livelyspeaker_amd.ted_evaluator import EmbeddingSpaceEvaluator
    g = np.load(os.path.join(GOLD, "eval_ted_golden.npz"))
    sd = {k: torch.from_numpy(v) for k, v in synth.make_embedding_net_state_dict(27, 32).items()}
    sd["decoder.net.0.weight"] = torch.zeros(4, 4)                       # the checkpoint also carries the decoder: ignored
    ev = EmbeddingSpaceEvaluator(ckpt={"pose_dim": 27, "gen_dict": sd})
    gen, real = synth.make_pose_sets(640)
    for i in range(0, 640, 64):
        ev.push_samples(torch.from_numpy(gen[i:i + 64]).cuda(), torch.from_numpy(real[i:i + 64]).cuda())
    assert ev.get_no_of_samples() == 10
    fd, feat_dist = ev.get_scores()
    assert abs(fd - float(g["ted_frechet"])) < 1e-3 * float(g["ted_frechet"])
    assert abs(feat_dist - float(g["ted_feat_dist"])) < 1e-4 * float(g["ted_feat_dist"])
    torch.manual_seed(4)
    div = ev.get_diversity_scores()
    assert abs(div - float(g["ted_diversity"])) < 1e-4 * float(g["ted_diversity"])
