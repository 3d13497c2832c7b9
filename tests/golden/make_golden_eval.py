#!/usr/bin/env python3
"""Golden vectors for the FGD evaluator from the IMPORTED reference (build container only).

Runs the reference's own EmbeddingNet (scripts/model/embedding_net.py), BEAT's HalfEmbeddingNet
(scripts_beat/model/motion_autoencoder.py) and EmbeddingSpaceEvaluator.push_samples / get_scores /
get_diversity_scores (scripts/model/ted_evaluator.py:38-152) on synth weights and pose sets.  The evaluator's __init__
hard-codes a checkpoint path and cuda:0 (:13-19), so the instance is created with object.__new__ and given the net
directly; `umap` (imported at module level, used only for visualisation) is stubbed.
"""
import os
import subprocess
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def main(which):
    sys.dont_write_bytecode = True
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from livelyspeaker_amd import synth
    from oracle import eval_oracle as evo
    torch.set_num_threads(4)
    out = {}
    if which == "ted":
        sys.modules["umap"] = types.ModuleType("umap")
        sys.path.insert(0, os.path.join(REF, "scripts"))
        from model.embedding_net import EmbeddingNet
        from model import ted_evaluator as te
        sd = synth.make_embedding_net_state_dict(27, 32)
        net = EmbeddingNet(27, 34)
        missing, unexpected = net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=False)
        assert not unexpected and all(k.startswith("decoder.") or "num_batches_tracked" in k for k in missing), missing
        net.train(False)
        ev = object.__new__(te.EmbeddingSpaceEvaluator)
        ev.net, ev.pose_dim = net, 27
        ev.reset()
        gen, real = synth.make_pose_sets(640)
        with torch.no_grad():
            for i in range(0, 640, 64):
                ev.push_samples(torch.from_numpy(gen[i:i + 64]), torch.from_numpy(real[i:i + 64]))
        out["ted_gen_feat"] = np.vstack(ev.generated_feat_list)
        out["ted_real_feat"] = np.vstack(ev.real_feat_list)
        fd, feat_dist = ev.get_scores()
        out["ted_frechet"], out["ted_feat_dist"] = np.float64(fd), np.float64(feat_dist)
        torch.manual_seed(4)
        out["ted_diversity"] = np.float64(ev.get_diversity_scores())
        torch.manual_seed(4)
        out["ted_diversity_perm"] = torch.randperm(len(ev.generated_feat_list))[:500].numpy()
        # oracle
        of = evo.pose_encoder(sd, gen)
        fscale = float(np.abs(out["ted_gen_feat"]).max())
        print("  oracle vs reference features: max|d| =", float(np.abs(of - out["ted_gen_feat"]).max()), "max|feat| =", fscale)
        assert np.abs(of - out["ted_gen_feat"]).max() < 1e-5 * max(1.0, fscale)
        ofd, ofe = evo.scores(out["ted_gen_feat"], out["ted_real_feat"])
        assert abs(ofd - fd) < 1e-9 * max(1.0, abs(fd)) and abs(ofe - feat_dist) < 1e-9
        print(f"  frechet {fd:.6f} feat_dist {feat_dist:.6f} diversity {float(out['ted_diversity']):.6f}")
    else:
        sys.path.insert(0, os.path.join(REF, "scripts_beat"))
        from types import SimpleNamespace
        from model.motion_autoencoder import HalfEmbeddingNet
        sd = synth.make_embedding_net_state_dict(141, 48, seed=synth.SEED_WEIGHTS + 201, hidden=(4, 2))
        net = HalfEmbeddingNet(SimpleNamespace(pose_length=34, pose_dims=141, vae_length=48))
        missing, unexpected = net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=False)
        assert not unexpected and all(k.startswith("decoder.") or "num_batches_tracked" in k for k in missing), missing
        net.eval()
        gen, _ = synth.make_pose_sets(96, 141, seed=synth.SEED_COND + 3001)
        with torch.no_grad():
            out["beat_feat"] = net(torch.from_numpy(gen)).numpy()
        of = evo.pose_encoder(sd, gen)
        fscale = float(np.abs(out["beat_feat"]).max())
        print("  oracle vs reference (BEAT HalfEmbeddingNet) features: max|d| =", float(np.abs(of - out["beat_feat"]).max()), "max|feat| =", fscale)
        assert np.abs(of - out["beat_feat"]).max() < 1e-5 * max(1.0, fscale)
    np.savez_compressed(os.path.join(HERE, f"eval_{which}_golden.npz"), **out)
    print("  wrote", f"eval_{which}_golden.npz")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        main(sys.argv[1])
    else:
        for w in ("ted", "beat"):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), w])
