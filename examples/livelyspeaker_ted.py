#!/usr/bin/env python3
"""End-to-end LivelySpeaker inference on one MI355X with the drop-in modules, mirroring
scripts/test_LivelySpeaker_ted.py:57-113 + :176-224 of the reference on SYNTHETIC inputs (no dataset / checkpoints here):

    CLIP text feature z (stand-in)  ->  SAG decoder (script-guided motion)  ->  init_image
    RAG + classifier-free guidance, ddim100, skip_timesteps=80 (20 refinement steps conditioned on audio / speaker / prefix poses)
    ->  post-processing (aligned motions, poses, motion beats)  ->  FGD / diversity against the "real" clips

    python examples/livelyspeaker_ted.py [batch]
With real data: load RAG.pt / SAG.pth / the auto-encoder checkpoint with load_model_wo_clip / load_state_dict and build `cond`
exactly as the reference script does; everything below the weight loading is unchanged.
"""
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from livelyspeaker_amd import synth                                                   # noqa: E402
from livelyspeaker_amd.cfg_sampler import ClassifierFreeSampleModel                   # noqa: E402
from livelyspeaker_amd.model_util import create_model_and_diffusion, load_model_wo_clip   # noqa: E402
from livelyspeaker_amd.motionclip_module import Decoder_TRANSFORMER                   # noqa: E402
from livelyspeaker_amd.postprocess import ted_postprocess                             # noqa: E402
from livelyspeaker_amd.ted_evaluator import EmbeddingSpaceEvaluator                   # noqa: E402


def build(device="cuda:0"):
    cfg = synth.TED
    args = SimpleNamespace(mdm_condm="text", latent_dim=512, ff_size=1024, layers=8, cond_mask_prob=0.1, arch="trans_enc",
                           emb_trans_dec=False, dataset="humanml", lang_model=None, mlpact="silu", diffusion_steps=1000,
                           noise_schedule="cosine", sigma_small=True, lambda_vel=1.0, lambda_rcxyz=0.0, lambda_fc=0.0, njoints=9)
    model, diffusion = create_model_and_diffusion(args, 'ddim100')                          # test_LivelySpeaker_ted.py:190
    load_model_wo_clip(model, {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()})
    model = ClassifierFreeSampleModel(model).to(device)
    model.eval()
    sag_decoder = Decoder_TRANSFORMER(latent_dim=512, n_pre_poses=4, use_style=False)       # motionclip.py:90-91
    sag_decoder.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_sag_state_dict(cfg).items()}, strict=False)
    sag_decoder.to(device)
    sag_decoder.eval()
    evaluator = EmbeddingSpaceEvaluator(ckpt={"pose_dim": 27, "gen_dict": {k: torch.from_numpy(v) for k, v in
                                                                           synth.make_embedding_net_state_dict(27, 32).items()}},
                                        device=device)
    return cfg, model, diffusion, sag_decoder, evaluator


def make_inputs(cfg, B, device="cuda:0", guidance_param=2.5, seed=0):
    """What the reference's data loader + CLIP text encoder hand to the loop body (:63-100), on the device (`seed`: which loader batch)."""
    y = synth.make_cond(cfg, B, scale=guidance_param, seed=synth.SEED_COND + seed)
    vec_seq = torch.from_numpy(y["origin_x"]).to(device)                                    # [B,9,3,34] "ground truth" clip
    batch = {"x": vec_seq.clone(), "mask": torch.ones(B, 34, device=device).bool(),
             "z": torch.from_numpy(synth.make_text_features(B, seed=synth.SEED_COND + 2000 + seed)).to(device)}                 # clip_model.encode_text(...) stand-in
    cond = {"y": {"mask": torch.ones(B, 34, device=device).bool(), "audio_input": torch.from_numpy(y["audio_input"]).to(device),
                  "vid_indices": torch.from_numpy(y["vid_indices"]).to(device), "origin_x": vec_seq.clone(),
                  "scale": torch.ones(B, device=device) * guidance_param}}
    return vec_seq, batch, cond


def infer(model, diffusion, sag_decoder, batch, cond, skip_steps=80, seed=233, noise_source="torch_cpu"):
    """SAG decode -> guided refinement (:88-113).  noise_source 'torch_cpu' replays the reference's CPU random stream draw by
    draw (slow: ~1.3 s of host RNG at B=512); 'philox' generates the noise inside the step kernel."""
    diffusion.noise_source = noise_source
    B = batch["x"].shape[0]
    if hasattr(model, "prefetch_condition"):          # optional: the refinement's once-per-call stage overlaps the SAG decode
        model.prefetch_condition(cond["y"] if "y" in cond else cond)
    decoded_motions = sag_decoder(batch)["output"]
    torch.manual_seed(seed)
    sample = diffusion.ddim_sample_loop(model, (B, 9, 3, 34), clip_denoised=False, model_kwargs=cond, skip_timesteps=skip_steps,
                                        init_image=decoded_motions, progress=False, dump_steps=None, noise=None, const_noise=False)
    return decoded_motions, sample


def infer_pipelined(models, diffusion, sag_decoder, batches, conds, skip_steps=80, seed=233):
    """The reference's loader loop (test_LivelySpeaker_ted.py:57-113) over SEVERAL batches, pipelined across calls: the SAG decode and
    the refinement's once-per-call stage of batch n + 1 are ENQUEUED -- on the decoder's stream and on the stream of the other of two
    model replicas -- before batch n's refinement loop is waited for, so they run inside that loop's launch tails and the host never
    stands between two batches.  Results are bitwise those of `infer` called batch by batch with the same seed
    (tests/test_gpu_pipeline.py).  `models`: two CFG-wrapped RAG replicas with the same weights (a handle holds ONE prepared batch)."""
    from livelyspeaker_amd import _lib
    diffusion.noise_source = "philox"
    dev = next(models[0].parameters()).device
    torch_stream = torch.cuda.current_stream(dev).cuda_stream
    sag_stream = sag_decoder.engine()._stream

    def stage(n):
        models[n % 2].prefetch_condition(conds[n]["y"])
        return sag_decoder(batches[n], wait=False)["output"]

    torch.manual_seed(seed)
    decoded = stage(0)
    outs = []
    for n in range(len(batches)):
        _lib.stream_order(dev.index or 0, sag_stream, torch_stream)     # consumers of decode n (this stream, then the engine's) wait for it
        nxt = stage(n + 1) if n + 1 < len(batches) else None            # ... but not for decode n + 1, enqueued behind that point
        B = batches[n]["x"].shape[0]
        outs.append((decoded, diffusion.ddim_sample_loop(models[n % 2], (B, 9, 3, 34), clip_denoised=False, model_kwargs=conds[n],
                                                          skip_timesteps=skip_steps, init_image=decoded, progress=False, dump_steps=None,
                                                          noise=None, const_noise=False)))
        decoded = nxt
    return outs


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    if len(sys.argv) > 2:                                                                   # python examples/livelyspeaker_ted.py B N: N batches, pipelined
        return main_pipelined(B, int(sys.argv[2]))
    cfg, model, diffusion, sag_decoder, evaluator = build()
    vec_seq, batch, cond = make_inputs(cfg, B)
    infer(model, diffusion, sag_decoder, batch, cond, noise_source="philox")                # warm-up (graph capture, allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    decoded, sample = infer(model, diffusion, sag_decoder, batch, cond, noise_source="philox")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    post = ted_postprocess(sample)                                                          # test_RAG_ted.py:84-111
    real = vec_seq.permute(0, 3, 1, 2).reshape(B, 34, -1)
    for i in range(0, B, 64):
        evaluator.push_samples(post["aligned_motions"][i:i + 64], real[i:i + 64])
    fgd, feat_dist = evaluator.get_scores() if B > 32 else (float("nan"), float("nan"))
    n_beats = sum(len(b) for b in post["motion_beat_times"])
    print(f"B={B}: SAG decode + 20-step guided refinement {dt * 1e3:.1f} ms ({B * 34 / dt:.0f} pose-frames/s); "
          f"motion beats {n_beats}; FGD {fgd:.4f}, feature distance {feat_dist:.4f} (synthetic weights: numbers are not quality)")
    assert bool(torch.isfinite(sample).all())


def main_pipelined(B, N):
    cfg, model, diffusion, sag_decoder, _ = build()
    _, model2, _, _, _ = build()
    ins = [make_inputs(cfg, B, seed=n) for n in range(N)]
    batches, conds = [i[1] for i in ins], [i[2] for i in ins]
    infer_pipelined([model, model2], diffusion, sag_decoder, batches[:2], conds[:2])        # warm-up: both replicas capture their graphs
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = infer_pipelined([model, model2], diffusion, sag_decoder, batches, conds)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    print(f"B={B} x {N} batches, pipelined across calls: {dt * 1e3:.2f} ms per batch ({B * 34 / dt:.0f} pose-frames/s)")
    assert all(bool(torch.isfinite(o).all()) for _, o in outs)


if __name__ == "__main__":
    main()
