"""The components composed as scripts/test_LivelySpeaker_ted.py composes them (SAG decoder -> init_image -> CFG RAG, ddim100
with skip_timesteps=80 -> post-processing -> FGD evaluator), drop-in modules on the GPU vs the CPU oracles chained the same
way on the same synthetic inputs and the same torch-CPU random draws.  Catches layout / hand-over mistakes between parts that
the per-component parity tests cannot see."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pipelining_across_calls_is_bitwise_the_call_by_call_order():
    """The loader loop of scripts/test_LivelySpeaker_ted.py:57-113 over five batches (a ragged last one), two ways: call by call (decode,
    prepare, refine, wait -- per batch), and pipelined (decode + prepare of batch n + 1 enqueued on their own streams before batch n's
    refinement is waited for, two model replicas alternating).  Same seed -> the same Philox keys in the same order -> identical bits."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import livelyspeaker_ted as ex
    cfg, model, diffusion, sag_decoder, _ = ex.build()
    _, model2, _, _, _ = ex.build()
    sizes = [40, 40, 40, 40, 13]
    ins = [ex.make_inputs(cfg, b, seed=n) for n, b in enumerate(sizes)]
    batches, conds = [i[1] for i in ins], [i[2] for i in ins]
    diffusion.noise_source = "philox"
    torch.manual_seed(5)
    serial = []
    for n in range(len(sizes)):                                      # ex.infer without its per-call manual_seed
        dec = sag_decoder(batches[n])["output"]
        serial.append((dec, diffusion.ddim_sample_loop(model, (sizes[n], 9, 3, 34), clip_denoised=False, model_kwargs=conds[n],
                                                       skip_timesteps=80, init_image=dec, progress=False, dump_steps=None, noise=None,
                                                       const_noise=False)))
    for rep in range(3):                                             # first run captures the replicas' graphs, later ones replay them
        piped = ex.infer_pipelined([model, model2], diffusion, sag_decoder, batches, conds, seed=5)
        torch.cuda.synchronize()
        for n, ((d0, s0), (d1, s1)) in enumerate(zip(serial, piped)):
            assert torch.equal(d0, d1), (rep, n, "decode")
            assert torch.equal(s0, s1) and bool(torch.isfinite(s1).all()), (rep, n, float((s0 - s1).abs().max()))
    assert not torch.equal(serial[0][1], serial[1][1])               # the batches differ (the check has teeth)


def test_livelyspeaker_pipeline_matches_chained_oracles():
    import torch
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import livelyspeaker_ted as ex
    from livelyspeaker_amd import postprocess as pp
    from livelyspeaker_amd import synth
    from oracle import eval_oracle as evo
    from oracle import rag_oracle as orc

    B, scale, skip = 6, 2.5, 80
    cfg, model, diffusion, sag_decoder, evaluator = ex.build()
    vec_seq, batch, cond = ex.make_inputs(cfg, B, guidance_param=scale)
    decoded, sample = ex.infer(model, diffusion, sag_decoder, batch, cond, skip_steps=skip, seed=11)
    post = pp.ted_postprocess(sample)
    real = vec_seq.permute(0, 3, 1, 2).reshape(B, 34, -1)
    feat = evaluator.net(post["aligned_motions"], variational_encoding=False)[0].cpu().numpy()

    # ---- the same chain on the CPU oracles, replaying the loop's torch-CPU draws (reference order) ----
    y = synth.make_cond(cfg, B, scale=scale)
    sag = orc.SagDecoderOracle(synth.make_sag_state_dict(cfg))
    init = sag.decode(y["origin_x"], synth.make_text_features(B), None)
    assert np.abs(decoded.cpu().numpy() - init).max() < 1e-4
    sch = orc.Schedule(1000, "ddim100")
    n_exec = sch.num_timesteps - skip
    torch.manual_seed(11)
    shape = (B, 9, 3, 34)
    x_init = torch.randn(*shape).numpy()
    eps_tape, noise_tape = [], []
    proto = torch.empty(shape)
    for k in range(n_exec):
        ec, eu = torch.randn(B, 1, 512).numpy().reshape(B, 512), torch.randn(B, 1, 512).numpy().reshape(B, 512)
        eps_tape.append(np.stack([ec, eu]))
        if k >= 1:                                     # the reference's x is a permuted view after the first step (G7)
            proto = torch.empty(34, B, 9, 3).permute(1, 2, 3, 0)
        noise_tape.append(torch.randn_like(proto).numpy())
    oracle = orc.RagOracle(synth.make_state_dict(cfg), cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens)
    want = orc.sample_loop(oracle, sch, y, x_init, eps_tape, noise_tape, ddim=True, eta=0.0, skip_timesteps=skip, init_image=init)
    d = float(np.abs(sample.cpu().numpy() - want).max())
    print(f"pipeline sample max|d| = {d:.3e}")
    assert d < 1e-3
    opost = orc.ted_post(want, pp.TED_MEAN_DIR_VEC, pp.TED_ANGLE_PAIRS, pp.TED_CHANGE_ANGLE, pp.TED_BEAT_THRES, pp.TED_DIR_VEC_PAIRS)
    assert np.abs(post["aligned_motions"].cpu().numpy() - opost["aligned"]).max() < 1e-3
    assert np.abs(post["pose"].cpu().numpy() - opost["pose"]).max() < 2e-3
    ofeat = evo.pose_encoder(synth.make_embedding_net_state_dict(27, 32), opost["aligned"])
    assert np.abs(feat - ofeat).max() < 2e-3 * max(1.0, float(np.abs(ofeat).max()))
    assert real.shape == (B, 34, 27)


def test_prefetched_condition_equals_the_serial_order():
    """RAG.prefetch_condition (ls_prepare_async under the SAG decode) must give bit-identical samples to the reference's order
    (decode, then the once-per-call stage inside the sampling call), with the conditioning cache on or off; prepare_ms reads -1 while
    the asynchronous stage is in flight and a real time after the call that synchronises."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import livelyspeaker_ted as ex
    B = 40
    cfg, model, diffusion, sag_decoder, _ = ex.build()
    _, batch, cond = ex.make_inputs(cfg, B, guidance_param=2.5)
    diffusion.noise_source = "philox"

    def run(prefetch):
        if prefetch:
            model.prefetch_condition(cond["y"])
        dec = sag_decoder(batch)["output"]
        torch.manual_seed(5)
        out = diffusion.ddim_sample_loop(model, (B, 9, 3, 34), clip_denoised=False, model_kwargs=cond, skip_timesteps=80, init_image=dec,
                                         progress=False, dump_steps=None, noise=None, const_noise=False)
        return dec.cpu().numpy(), out.cpu().numpy()

    inner = model.model if hasattr(model, "model") else model
    for cache in (True, False):
        inner.cache_conditioning = cache
        inner._cond_key = None
        d0, s0 = run(False)
        inner._cond_key = None
        d1, s1 = run(True)
        assert np.array_equal(d0, d1) and np.array_equal(s0, s1), f"cache_conditioning={cache}"
    # a prefetch that is superseded by another conditioning must not be honoured later
    _, batch2, cond2 = ex.make_inputs(cfg, B, guidance_param=1.5)
    cond2["y"]["audio_input"] = cond2["y"]["audio_input"] * 0.5
    inner.cache_conditioning = True
    inner._cond_key = None
    _, ref2 = run(False)                                   # cond (resident afterwards)
    model.prefetch_condition(cond["y"])
    dec = sag_decoder(batch)["output"]
    torch.manual_seed(5)
    other = diffusion.ddim_sample_loop(model, (B, 9, 3, 34), clip_denoised=False, model_kwargs=cond2, skip_timesteps=80, init_image=dec,
                                       progress=False, dump_steps=None, noise=None, const_noise=False).cpu().numpy()
    assert not np.array_equal(other, ref2)                 # really the other conditioning
    _, again = run(False)                                  # back to cond: must re-run the stage, not trust the stale prefetch
    assert np.array_equal(again, ref2)
    eng = inner.engine()
    inner._cond_key = None
    model.prefetch_condition(cond["y"])
    assert eng.timing()["prepare_ms"] == -1.0 or eng.timing()["prepare_ms"] > 0      # in flight, or already done when queried
    eng.synchronize() if hasattr(eng, "synchronize") else None
    run(False)
    assert eng.timing()["prepare_ms"] > 0


def test_bench_strong_scaling_path_under_torch_distributed_run(tmp_path):
    """BASELINE configs[3]'s code path end to end on ONE rank: `torch.distributed.run --nproc-per-node 1 bench.py --gpus 1
    --global-batch G` (RCCL process group, shard + all_gather inside the timed region, shard cross-check, in-run parity), next to
    the plain weak-scaling invocation of the same workload: one JSON line each, values within a few percent."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--diffusion-steps", "100", "--no-cpu-baseline", "--no-extra-legs"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r1 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                         "--master-port", "29611", os.path.join(root, "bench.py"), "--global-batch", "512"] + common,
                        capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r1.returncode == 0, r1.stderr[-3000:]
    lines = [l for l in r1.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r1.stdout
    strong = json.loads(lines[0])
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=900, cwd=root)
    assert r2.returncode == 0, r2.stderr[-3000:]
    weak = json.loads([l for l in r2.stdout.splitlines() if l.strip()][0])
    print("strong:", strong["value"], strong["shard_check"], strong["parity_in_run"])
    print("weak:  ", weak["value"], weak["parity_in_run"])
    assert strong["scaling"] == "strong" and weak["scaling"] == "weak"
    assert strong["rccl_ranks"] == 1 and strong["shard_check"]["bitwise_equal"]
    assert strong["shard_check"]["checksum_recomputed"] == strong["shard_check"]["checksum_sharded"]
    assert strong["parity_in_run"]["ok"] and weak["parity_in_run"]["ok"]
    assert strong["config"]["global_batch"] == 512 and weak["config"]["global_batch"] == 512
    assert abs(strong["value"] / weak["value"] - 1) < 0.05           # 100-step calls (150 ms): the gather and host jitter are visible


def _bench(root, args, timeout=1200):
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=root)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                                   # ONE JSON line on stdout, whatever the ranks print elsewhere
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks_and_runs_the_n_rank_path_with_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` (no launcher) re-executes itself under torch.distributed.run.  With --ranks-share-device both
    ranks use cuda:0 -- RCCL refuses that ("Duplicate GPU detected"), so the collectives run over gloo on host copies -- and
    everything else of the N > 1 path executes for real: rank-0 weight broadcast (rank 1 starts from DIFFERENT weights), Philox
    streams keyed by the global sample index, shard cross-check of the neighbour's shard, max-over-ranks timing, rank 0's single
    JSON line; and configs[2]'s leg with the CLIP-feature broadcast + all_gather in its timed region."""
    root = ROOT
    common = ["--steps", "1", "--warmup", "1", "--diffusion-steps", "60", "--no-cpu-baseline", "--batch", "128"]
    two = _bench(root, ["--gpus", "2", "--ranks-share-device", "--legs", "lively"] + common)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["config"]["global_batch"] == 256
    assert two["collective_backend"] == "gloo" and two["rccl_ranks"] == 0 and two["shard_check"]["ranks"] == 2
    assert two["rccl_error"]            # RCCL was TRIED (shard.init_groups), failed its probe, and the run carried on over gloo and said so
    assert two["shard_check"]["bitwise_equal"] and two["shard_check"]["checksum_recomputed"] == two["shard_check"]["checksum_sharded"]
    assert two["parity_in_run"]["ok"]
    lv = two["livelyspeaker"]
    assert "error" not in lv, lv
    assert lv["n_gpus"] == 2 and lv["text_feature_broadcast"]["bytes"] == 256 * 512 * 4 and lv["text_feature_broadcast"]["ms"] > 0
    # strong scaling through the same self-launch: 256 clips over the two ranks, gather in the timed region
    strong = _bench(root, ["--gpus", "2", "--ranks-share-device", "--legs", "none", "--global-batch", "256"] + common)
    assert strong["scaling"] == "strong" and strong["shard_check"]["bitwise_equal"] and strong["parity_in_run"]["ok"]
    # one rank, same workload: the leg appears there too, and the values per clip are the same computation
    one = _bench(root, ["--gpus", "1", "--legs", "lively"] + common)
    assert one["n_gpus"] == 1 and "error" not in one["livelyspeaker"] and one["livelyspeaker"]["text_feature_broadcast"] is None
    print("2 ranks on one GPU:", two["value"], lv["value"], lv["text_feature_broadcast"], "| strong:", strong["value"], "| 1 rank:", one["value"])


def test_bench_threads_launcher_two_handles_on_one_gpu():
    """`--launcher threads`: one process, one engine handle per device driven by one Python thread each, no process group and no
    collective -- the independent N-device number / cross-check of the torchrun path.  On this one-GPU box both handles sit on
    cuda:0 (--ranks-share-device): same shards, same Philox streams as two torchrun ranks would produce."""
    common = ["--steps", "1", "--warmup", "1", "--diffusion-steps", "60", "--batch", "128"]
    thr = _bench(ROOT, ["--gpus", "2", "--launcher", "threads", "--ranks-share-device"] + common)
    assert thr["n_gpus"] == 2 and thr["config"]["global_batch"] == 256 and thr["collective_backend"] == "none" and thr["rccl_ranks"] == 0
    assert thr["shard_check"]["bitwise_equal"] and thr["launcher"].startswith("threads")
    strong = _bench(ROOT, ["--gpus", "2", "--launcher", "threads", "--ranks-share-device", "--global-batch", "300"] + common)
    assert strong["scaling"] == "strong" and strong["config"]["global_batch"] == 300 and strong["shard_check"]["bitwise_equal"]
    print("threads launcher, 2 handles on one GPU:", thr["value"], "| strong 300:", strong["value"])


def test_eight_launchers_on_one_gpu_both_ways():
    """What one GPU allows of the driver's N = 8 run: eight torchrun ranks on cuda:0 (gloo control plane, RCCL probed and refused,
    every rank re-generating its neighbour's shard bitwise, ONE JSON line), and eight engine handles driven by eight threads of one
    process.  Throughput here is eight step loops time-sharing one GPU (eight streams over the runtime's hardware queues), not a
    scaling number; what is checked is that nothing deadlocks or serialises completely: the eight handles together reach at least
    half of what one handle does with the same 512 clips."""
    common = ["--steps", "1", "--warmup", "1", "--diffusion-steps", "40", "--no-cpu-baseline", "--legs", "none"]
    eight = _bench(ROOT, ["--gpus", "8", "--ranks-share-device", "--batch", "16"] + common, timeout=1800)
    assert eight["n_gpus"] == 8 and eight["config"]["global_batch"] == 128 and eight["collective_backend"] == "gloo" and eight["rccl_error"]
    assert eight["shard_check"]["ranks"] == 8 and eight["shard_check"]["bitwise_equal"] and eight["parity_in_run"]["ok"]
    thr_common = ["--steps", "2", "--warmup", "1", "--diffusion-steps", "100"]
    thr = _bench(ROOT, ["--gpus", "8", "--launcher", "threads", "--ranks-share-device", "--batch", "64", "--path", "pass"] + thr_common)
    one = _bench(ROOT, ["--gpus", "1", "--launcher", "threads", "--batch", "512"] + thr_common)
    assert thr["n_gpus"] == 8 and thr["config"]["global_batch"] == 512 and thr["shard_check"]["bitwise_equal"] and thr["shard_check"]["ranks"] == 8
    print("8 ranks on one GPU:", eight["value"], "| 8 handles / 8 threads:", thr["value"], "| 1 handle, 512 clips:", one["value"])
    assert thr["value"] > 0.5 * one["value"]


def test_two_handles_driven_from_two_threads_equal_the_sequential_results():
    """include/ls_hip.h: "a handle is not thread-safe, distinct handles are independent".  Two handles on cuda:0 (different batches,
    different kernels: the fused one and the sample-split one), each driven by its own thread at the same time, give bitwise the
    results they give one after the other; a failing ls_create on one thread does not disturb the other thread's error text."""
    import threading
    import numpy as np
    from livelyspeaker_amd import _lib, synth
    from oracle import rag_oracle as orc
    cfg = synth.TED
    sd = synth.make_state_dict(cfg)
    engs = []
    for path, B in (("fused", 130), ("coop", 24)):
        e = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, path=path)
        e.load_state_dict(sd)
        e.set_schedule(orc.Schedule(40, ""))
        e.prepare(synth.make_cond(cfg, B, seed=B))
        engs.append(e)
    try:
        seq = [e.sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=7 + i) for i, e in enumerate(engs)]
        for rnd in range(3):
            got, errs = [None, None], []
            bar = threading.Barrier(2)

            def work(i):
                try:
                    bar.wait()
                    for _ in range(3):
                        got[i] = engs[i].sample(sampler=_lib.LS_SAMPLER_DDPM, philox_seed=7 + i)
                    if i == 1:          # a failing create on this thread: its message must not leak into the other thread's state
                        with pytest.raises(_lib.EngineError, match="unsupported shape"):
                            _lib.Engine(5, 5, 1, cfg.audio_len)
                except Exception as ex:     # noqa: BLE001
                    errs.append(ex)
            ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            assert not errs, errs
            assert np.array_equal(got[0], seq[0]) and np.array_equal(got[1], seq[1]), rnd
    finally:
        for e in engs:
            e.close()
