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


def test_streaming_statistics_and_eigh_frechet_match_the_reference_scores():
    """The drop-in's own statistics (batch-wise parallel-variance accumulation; Tr sqrt(S1 S2) from two symmetric
    eigendecompositions) against the scores the reference's evaluator produced for the same features (fixture), against the
    scipy-sqrtm restatement, and on degenerate inputs."""
    from livelyspeaker_amd import ted_evaluator as te
    g = np.load(os.path.join(GOLD, "eval_ted_golden.npz"))
    gen, real = g["ted_gen_feat"], g["ted_real_feat"]
    mg, mr = te._Moments(), te._Moments()
    for i in range(0, len(gen), 64):                     # ragged last batch on purpose
        mg.add(gen[i:i + 37]); mg.add(gen[i + 37:i + 64])
        mr.add(real[i:i + 64])
    assert mg.n == len(gen) and np.allclose(mg.mean, gen.mean(0, dtype=np.float64), atol=1e-12)
    assert np.allclose(mg.covariance(), np.cov(gen.astype(np.float64), rowvar=False), rtol=1e-10, atol=1e-12)
    fd = te.EmbeddingSpaceEvaluator.calculate_frechet_distance(mg.mean, mg.covariance(), mr.mean, mr.covariance())
    # the reference takes the means with np.mean on float32 features (float32 result, ~1e-7 relative) and feeds them to
    # ||mu1 - mu2||^2; the accumulation here is float64 throughout, so agreement is at that level, not at float64 round-off
    assert abs(fd - float(g["ted_frechet"])) < 2e-6 * float(g["ted_frechet"])
    want64 = evo.calculate_frechet_distance(gen.mean(0, dtype=np.float64), np.cov(gen, rowvar=False), real.mean(0, dtype=np.float64),
                                            np.cov(real, rowvar=False))
    assert abs(fd - want64) < 1e-9 * want64              # same inputs in float64: eigh route == scipy sqrtm route
    rng = np.random.Generator(np.random.PCG64(2))
    for n, d in ((40, 6), (5, 9)):                       # second case: rank-deficient covariances (fewer samples than dims)
        a, b = rng.standard_normal((n, d)), rng.standard_normal((n, d)) * 0.5 + 0.3
        want = evo.calculate_frechet_distance(a.mean(0), np.cov(a, rowvar=False), b.mean(0), np.cov(b, rowvar=False))
        got = te.EmbeddingSpaceEvaluator.calculate_frechet_distance(a.mean(0), np.cov(a, rowvar=False), b.mean(0), np.cov(b, rowvar=False))
        assert abs(got - want) < 1e-6 * max(1.0, abs(want)), (n, d, got, want)
    same = te.EmbeddingSpaceEvaluator.calculate_frechet_distance(mg.mean, mg.covariance(), mg.mean, mg.covariance())
    assert abs(same) < 1e-9
    assert te.EmbeddingSpaceEvaluator.calculate_frechet_distance([np.nan, 0.0], np.eye(2), [0.0, 0.0], np.eye(2)) == float("inf")
