#!/usr/bin/env python3
"""Headline benchmark: pose-frames/sec denoised (BASELINE.json metric).

One "step" = one full sampling call (`diffusion.p_sample_loop`) of the drop-in path over one synthetic
batch: TED RAG, batch 512 x 34 frames, 1000 DDPM steps, CFG scale 1.5 (BASELINE.json configs[1]),
including the once-per-call stage (audio encoder, static input projection, speaker style) and on-device
Philox noise; conditioning tensors are already resident in HBM when the timed region starts.

  python bench.py [--gpus N --steps K --warmup W]             # N=1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W               # N>1: one rank per GPU, batch sharded, weak scaling

Rank 0 prints ONE JSON line (see the driver contract in the task statement) with two extra objects:
"roofline" (fused step kernel vs the gfx950 FP32-matrix MFMA peak; duration from HIP events on the engine's
own stream) and "cpu_baseline" (the CPU oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
_RESULT_OUT = sys.stdout        # replaced by a private duplicate of fd 1 when run as a script (_reserve_stdout)

FLOP_PER_SAMPLE_STEP = {"ted": 317_431_808, "beat": 362_496_000}     # BASELINE.md section 3 (CFG: 2 forwards, hoisted form)
MFMA_F32_PEAK_TFLOPS = 157.3                                          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
# HBM bytes per k_step launch from rocprofv3 PMC passes (profiles/r01b_kernel_trace_and_pmc.md): 2*FETCH_SIZE + WRITE_SIZE
# (KiB -> B, with the guide's gfx950 FETCH_SIZE correction); measured for the default workload only.
PMC_TRAFFIC_BYTES = {("ted", 512): 218_940_170}      # profiles/r01e_final_kernel_trace_and_pmc.md (PMC section)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dataset", default="ted", choices=["ted", "beat"])
    ap.add_argument("--batch", type=int, default=512, help="clips per GPU")
    ap.add_argument("--diffusion-steps", type=int, default=1000)
    ap.add_argument("--respacing", default="", help="'' = DDPM over all steps, 'ddim100' = DDIM")
    ap.add_argument("--skip", type=int, default=0)
    ap.add_argument("--scale", type=float, default=1.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-split-leg", action="store_true", help="skip the secondary bf16x3 measurement")
    ap.add_argument("--train-leg", action="store_true", help="also time the training step (default on at 1 GPU; at N>1 it "
                    "adds the RCCL gradient all-reduce, the build's only collective)")
    ap.add_argument("--no-train-leg", action="store_true")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16x3_perpass"],
                    help="fp32 = exact (headline); bf16x3 = opt-in split-precision channel mixing")
    return ap.parse_args()


def mk_args(cfg, steps):
    return SimpleNamespace(mdm_condm="text", latent_dim=512, ff_size=1024, layers=8, cond_mask_prob=0.1,
                           arch="trans_enc", emb_trans_dec=False, dataset="humanml", lang_model=None, mlpact="silu",
                           diffusion_steps=steps, noise_schedule="cosine", sigma_small=True, lambda_vel=1.0,
                           lambda_rcxyz=0.0, lambda_fc=0.0, njoints=cfg.njoints)


def cpu_baseline(cfg, args):
    """CPU oracle (numpy port of the reference algorithm) on a bounded sample, extrapolated linearly in the
    number of (homogeneous) diffusion steps.  Reported next to the GPU number; it is not the target."""
    from livelyspeaker_amd import synth
    from oracle import rag_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        cores = os.cpu_count() or 1
    Bc, n_h, n_f = 64, 12, 2
    sd = synth.make_state_dict(cfg)
    oracle = orc.RagOracle(sd, cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens)
    sch = orc.Schedule(args.diffusion_steps, args.respacing)
    y = synth.make_cond(cfg, Bc, scale=args.scale)
    tape = synth.NoiseTape(cfg, Bc, max(n_h, n_f))
    ddim = args.respacing.startswith("ddim")
    t0 = time.perf_counter()
    oracle.prepare(y)
    t_prep = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.sample_loop(oracle, sch, y, tape.x_init, tape.eps, tape.noise, ddim=ddim, hoisted=True, max_steps=n_h)
    t_h = (time.perf_counter() - t0 - 0.0) / n_h            # includes one more prepare; subtract below
    t_h = max(t_h - t_prep / n_h, 1e-9)
    t0 = time.perf_counter()
    orc.sample_loop(oracle, sch, y, tape.x_init, tape.eps, tape.noise, ddim=ddim, hoisted=False, max_steps=n_f)
    t_f = (time.perf_counter() - t0) / n_f
    n_exec = sch.num_timesteps - args.skip
    frames = Bc * cfg.nframes
    return {"value": round(frames / (t_prep + n_exec * t_h), 3), "unit": "pose-frames/s", "cores": int(cores),
            "kind": "port",
            "sample": f"numpy oracle, B={Bc}: 1 prepare + {n_h} hoisted steps ({t_h * 1e3:.1f} ms/step) extrapolated to "
                      f"{n_exec} steps; reference-faithful mode (audio encoder re-run 2x/step) {n_f} steps "
                      f"({t_f * 1e3:.1f} ms/step)",
            "reference_faithful_value": round(frames / (n_exec * t_f), 3)}


def train_leg(a, cfg, model, diffusion, dev, world, rank, use_dist, fence):
    """Time TrainLoop.run_step-equivalent work: forward + losses + backward (ls_train_forward_backward), gradient
    all-reduce over the ranks (RCCL, only when world > 1), AdamW.  Batch per GPU = --batch (reference default 512)."""
    import numpy as np
    import torch
    from livelyspeaker_amd import _lib, synth
    from livelyspeaker_amd.train_loop import allreduce_mean_
    B = a.batch
    tr = _lib.Trainer(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions, device=dev.index,
                      diffusion_steps=1000)
    tr.load_state_dict({k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if not k.endswith(".pe")})
    tr.set_schedule(_train_schedule(a))
    x_start, y, noise, drop, eps = synth.make_train_batch(cfg, B, 0, first_sample=rank * B, total=world * B)
    tod = lambda v: torch.from_numpy(np.asarray(v)).to(dev)
    x_start, noise, drop, eps = tod(x_start), tod(noise), tod(drop), tod(eps)
    y = {k: tod(v) for k, v in y.items()}
    t = np.random.Generator(np.random.PCG64(rank)).integers(0, 1000, size=(B,))
    n = 4
    fwd = bwd = 0.0
    terms = None
    for i in range(n + 1):
        if i == 1:
            fence()
            t0 = time.perf_counter()
        terms = tr.forward_backward(x_start, t, noise, y, drop, eps)
        allreduce_mean_(tr.grad)
        tr.adamw()
        if i >= 1:
            fwd += terms["fwd_ms"]
            bwd += terms["bwd_ms"]
    fence()
    el = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        el = float(tt.item())
    tr.close()
    cpu = None
    if world == 1 and not a.no_cpu_baseline:
        # the torch-CPU oracle (autograd) on this box's host cores, bounded sample: B=32, 2 steps after one warm-up
        from oracle import train_oracle as tro
        nthr = torch.get_num_threads()
        orc_t = tro.TrainOracle({k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if not k.endswith(".pe")},
                                cfg.n_prefix_tokens)
        xs, yy, nz, dr, ep = synth.make_train_batch(cfg, 32, 0)
        tt0 = np.random.Generator(np.random.PCG64(0)).integers(0, 1000, size=(32,))
        orc_t.optimizer_step(orc_t.forward_backward(xs, tt0, nz, yy, dr, ep)[2])
        c0 = time.perf_counter()
        for _ in range(2):
            orc_t.optimizer_step(orc_t.forward_backward(xs, tt0, nz, yy, dr, ep)[2])
        cpu = {"value": round(2 * 32 / (time.perf_counter() - c0), 2), "unit": "samples/s", "cores": nthr, "kind": "port",
               "sample": "torch-CPU oracle (oracle/train_oracle.py), B=32, 2 steps after 1 warm-up"}
    return {"cpu_baseline": cpu, "what": "forward + Huber/velocity/KLD losses + backward + AdamW (ls_train_*), fp32, inputs resident in HBM",
            "value": round(world * B * n / el, 1), "unit": "samples/s", "ms_per_step": round(el / n * 1e3, 3),
            "fwd_ms": round(fwd / n, 3), "bwd_ms": round(bwd / n, 3), "batch_per_gpu": B,
            "gradient_allreduce": "RCCL, 1 bucket of 16 MB, averaged" if world > 1 else "none (1 GPU)",
            "loss_after": round(terms["total"], 5),
            "parity": "gradients within 8e-6 (rel. to max|g|) of the reference-pinned oracle, tests/test_gpu_train.py"}


def _train_schedule(a):
    from livelyspeaker_amd.model_util import create_gaussian_diffusion
    from types import SimpleNamespace
    return create_gaussian_diffusion(SimpleNamespace(diffusion_steps=1000, noise_schedule="cosine", sigma_small=True, lambda_vel=1.0,
                                                     lambda_rcxyz=0.0, lambda_fc=0.0), "")


def main():
    global _RESULT_OUT
    a = parse()
    _RESULT_OUT = _reserve_stdout()
    import torch
    import torch.distributed as dist
    from livelyspeaker_amd import synth
    from livelyspeaker_amd.cfg_sampler import ClassifierFreeSampleModel
    from livelyspeaker_amd.model_util import create_model_and_diffusion

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = "RANK" in os.environ          # launched by torch.distributed.run (also with a single rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = synth.CONFIGS[a.dataset]
    B = a.batch
    model, diffusion = create_model_and_diffusion(mk_args(cfg, a.diffusion_steps), a.respacing, dataset=a.dataset)
    # random-init weights of the architecture, re-drawn with non-degenerate scale (the reference's own init
    # zeroes the channel-mix weights); rank 0's copy is broadcast over RCCL so all ranks agree.
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=synth.SEED_WEIGHTS + (0 if rank == 0 else 1)).items()}
    if use_dist:
        from livelyspeaker_amd import shard
        sd = shard.broadcast_state_dict(sd, dev)
    model.load_state_dict(sd, strict=False)
    model.to(dev)
    model.eval()
    model.precision = a.precision
    model.cache_conditioning = False        # every timed call re-runs the once-per-call stage (a new batch)
    cfgm = ClassifierFreeSampleModel(model)
    diffusion.noise_source = "philox"
    diffusion.use_graph = not a.no_graph
    diffusion.sample_offset = rank * B      # Philox streams keyed by the global sample index (shard-invariant)

    y_np = synth.make_cond(cfg, B, scale=a.scale, seed=synth.SEED_COND + rank)
    y = {k: torch.from_numpy(v).to(dev) for k, v in y_np.items()}
    shape = (B, cfg.njoints, cfg.nfeats, cfg.nframes)
    ddim = a.respacing.startswith("ddim")
    fn = diffusion.ddim_sample_loop if ddim else diffusion.p_sample_loop

    def one_call():
        return fn(cfgm, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=a.skip, init_image=None,
                  progress=False, dump_steps=None, noise=None, const_noise=False)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    torch.manual_seed(233)
    out = None
    for _ in range(a.warmup):
        out = one_call()
    eng = model.engine()
    loop_ms, launches, prep_ms = 0.0, 0, 0.0
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = one_call()
        tm = eng.timing()
        loop_ms += tm["loop_ms"]
        launches += tm["n_step_launches"]
        prep_ms += tm["prepare_ms"]
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert out is not None and bool(torch.isfinite(out).all()), "non-finite samples"

    # Secondary leg (never the headline `value`): the opt-in bf16x3 split-precision mode, same workload, same run.
    split = None
    if a.precision == "fp32" and not a.no_split_leg:
        model.precision = "bf16x3"
        one_call()
        fence()
        t1 = time.perf_counter()
        n2 = max(1, min(a.steps, 2))
        l2, k2 = 0.0, 0
        for _ in range(n2):
            one_call()
            tm = eng.timing()
            l2 += tm["loop_ms"]
            k2 += tm["n_step_launches"]
        fence()
        e2 = time.perf_counter() - t1
        if use_dist:
            t = torch.tensor([e2], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e2 = float(t.item())
        split = {"mode": "bf16x3: channel/token mixing as 3 bf16 MFMAs per fp32 product (hi.hi+hi.lo+lo.hi), fp32 accumulate",
                 "value": round(world * B * cfg.nframes * n2 / e2, 2), "unit": "pose-frames/s",
                 "kernel_ms": round(l2 / max(k2, 1), 4),
                 "parity": "max-abs vs reference golden after 1000 DDPM steps 3.5e-5 (contract 1e-3), tests/test_gpu_edge.py",
                 "note": "opt-in (RAG.precision / ls_set_precision); the headline value above is the exact-fp32 path"}
        model.precision = "fp32"

    # Secondary leg: one optimisation step of the denoiser, data-parallel over the ranks.
    train = None
    if (a.train_leg or world == 1) and not a.no_train_leg and a.precision == "fp32":
        try:
            train = train_leg(a, cfg, model, diffusion, dev, world, rank, use_dist, fence)
        except Exception as e:                      # never let the secondary leg take the headline line down
            train = {"error": repr(e)[:300]}

    if rank == 0:
        frames = world * B * cfg.nframes * a.steps
        n_exec = diffusion.num_timesteps - a.skip
        kernel_ms = loop_ms / max(launches, 1)
        achieved = FLOP_PER_SAMPLE_STEP[a.dataset] * B / (kernel_ms * 1e-3) / 1e12
        # fp32: FP32-matrix MFMA peak.  bf16x3: three bf16 MFMAs per algorithmic product -> dense bf16 peak / 3.
        peak = MFMA_F32_PEAK_TFLOPS if a.precision == "fp32" else round(2500.0 / 3.0, 1)
        rec = {
            "metric": "pose-frames/sec denoised", "value": round(frames / elapsed, 2), "unit": "pose-frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.precision == "fp32" else "bf16x3 split (fp32 accumulate) for channel mixing, f32 elsewhere",
            "data": "synthetic",
            "config": {"workload": f"{a.dataset.upper()} RAG, batch {B} x {cfg.nframes} frames per GPU, "
                                   f"{n_exec}-step {'DDIM' if ddim else 'DDPM'} ({a.diffusion_steps} diffusion steps"
                                   f"{', respacing ' + a.respacing if a.respacing else ''}), CFG scale {a.scale}, "
                                   f"random-init weights + synthetic audio/speaker/prefix-pose conditioning, Philox noise on device",
                       "global_batch": world * B, "frames": cfg.nframes, "denoise_steps": n_exec,
                       "guidance_scale": a.scale, "parallelism": f"batch-sharded x{world}, no per-step collective",
                       "hipgraph": bool(diffusion.use_graph)},
            "roofline": {"bound": "mfma", "kernel": "ls::k_step (fused CFG denoiser + sampler update, 1 launch/step)",
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4),
                         "traffic": PMC_TRAFFIC_BYTES.get((a.dataset, B)), "traffic_unit": "B/launch (rocprofv3 PMC, profiles/)",
                         "kernel_ms": round(kernel_ms, 4), "flop_per_launch": FLOP_PER_SAMPLE_STEP[a.dataset] * B,
                         "prepare_ms_per_call": round(prep_ms / a.steps, 3)},
        }
        if split is not None:
            rec["split_precision"] = split
        if train is not None:
            rec["train_step"] = train
        if world == 1 and not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(cfg, a)
        _RESULT_OUT.write(json.dumps(rec) + "\n")
        _RESULT_OUT.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def _reserve_stdout():
    """Keep stdout for the ONE JSON line: libraries below us write there too (RCCL prints a version banner at communicator
    creation), so fd 1 is pointed at stderr for the run and the result goes to a private duplicate of the original stdout."""
    sys.stdout.flush()
    out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return out


if __name__ == "__main__":
    main()
