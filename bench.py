#!/usr/bin/env python3
"""Headline benchmark: pose-frames/sec denoised (BASELINE.json metric).

One "step" = one full sampling call (`diffusion.p_sample_loop`) of the drop-in path over one synthetic
batch: TED RAG, batch 512 x 34 frames, 1000 DDPM steps, CFG scale 1.5 (BASELINE.json configs[1]),
including the once-per-call stage (audio encoder, static input projection, speaker style) and on-device
Philox noise; conditioning tensors are already resident in HBM when the timed region starts.

  python bench.py [--gpus N --steps K --warmup W]             # N=1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W               # N>1: one rank per GPU over RCCL
  ... bench.py --gpus N --global-batch 4096                   # STRONG scaling (BASELINE configs[3]): one global batch sharded
                                                              # over the ranks, result all-gathered inside the timed region

Default is weak scaling (512 clips per GPU, no data-path collective).  Rank 0 prints ONE JSON line (driver contract) with,
besides the contract's keys:
  "roofline"       fused step kernel vs the gfx950 FP32-matrix MFMA peak; duration from HIP events on the engine's own stream
  "parity_in_run"  two samples of the FIRST TIMED call replayed through the CPU oracle on the restated Philox noise
  "single_pass"    the guidance-scale-1 workload (uncond pass legitimately skipped), own FLOP count -- never mixed into `value`
  "livelyspeaker"  BASELINE configs[2] as the reference runs it: SAG decode + ddim100 / skip 80 refine, own roofline
  "configs4_beat"  BASELINE configs[4]: BEAT at 34 frames (B=256) and the SYNTHETIC 150-frame variant (B=32), own rooflines
  "mid_batches"    batches that do not fill the chip a whole number of times (64 / 80 / 128 / 160 / 384 clips): the engine's plan, own rooflines;
                   BEAT B=256 in the reference's own RNG mode (host-RNG bound stated)
  "shard_check"    the config-4 premise on hardware: every rank re-generates ANOTHER rank's shard via sample_offset
  "split_precision", "train_step"   secondary legs
  "cpu_baseline"   torch-CPU port of the reference algorithm timed on this box's host cores (bounded sample)
"""
import argparse
import hashlib
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
_RESULT_OUT = sys.stdout        # replaced by a private duplicate of fd 1 when run as a script (_reserve_stdout)

# BASELINE.md section 3 / SURVEY.md 8(d): hoisted form, 2 FLOPs per MAC.  CFG = two forwards per sample per step.
FLOP_PER_FORWARD = {"ted": 158_715_904, "beat": 181_248_000,
                    # synthetic 150-frame BEAT variant (S = 152): 2 x 43 315 200 (x_t in / poseFinal) + 8 x (23 658 496 token + 79 691 776 channel)
                    "beat150": 913_432_576}
MFMA_F32_PEAK_TFLOPS = 157.3                                          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "k_step_traffic.json")   # rocprofv3 PMC passes, keyed to the kernel source


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dataset", default="ted", choices=["ted", "beat", "beat150"],
                    help="beat150 = SYNTHETIC long-sequence variant (configs[4] as worded; perf-only, no parity claim vs the reference)")
    ap.add_argument("--batch", type=int, default=512, help="clips per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: total clips, sharded over the ranks")
    ap.add_argument("--diffusion-steps", type=int, default=1000)
    ap.add_argument("--respacing", default="", help="'' = DDPM over all steps, 'ddim100' = DDIM")
    ap.add_argument("--skip", type=int, default=0)
    ap.add_argument("--scale", type=float, default=1.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run oracle check of the first timed call")
    ap.add_argument("--parity-samples", type=int, default=2)
    ap.add_argument("--no-extra-legs", action="store_true", help="headline only: no single-pass / LivelySpeaker / bf16x3 / train legs")
    ap.add_argument("--legs", default="all", help="secondary legs to run: 'all', 'none' or a comma list of "
                    "single,split,lively,beat,train,seeds,small (e.g. --legs lively)")
    ap.add_argument("--no-split-leg", action="store_true", help="skip the secondary bf16x3 measurement")
    ap.add_argument("--train-leg", action="store_true", help="also time the training step (default on at 1 GPU; at N>1 it "
                    "adds the RCCL gradient all-reduce, the build's only per-step collective)")
    ap.add_argument("--no-train-leg", action="store_true")
    ap.add_argument("--ranks-share-device", action="store_true",
                    help="TEST ONLY (1-GPU boxes): every rank uses cuda:0 and the collectives run over gloo on host copies -- RCCL "
                         "refuses two ranks on one device ('Duplicate GPU detected'), so this exercises the whole N>1 control flow "
                         "(self-launch, sharding, broadcast, gather, cross-check, max-over-ranks timing) but not RCCL itself")
    ap.add_argument("--launcher", default="torchrun", choices=["torchrun", "threads"],
                    help="N > 1: 'torchrun' = one process per GPU with torch.distributed (RCCL, gloo fallback); 'threads' = ONE process, "
                         "N engine handles on N devices driven by N Python threads, no collective at all (an independent N-GPU number)")
    ap.add_argument("--path", default="auto", choices=["auto", "fused", "batch", "coop", "pass"],
                    help="kernel family of the step loop (ls_set_path); default: the engine's plan for the batch")
    ap.add_argument("--no-rccl", action="store_true", help="do not try RCCL: collectives over gloo on host copies")
    ap.add_argument("--rccl-probe-timeout", type=float, default=120.0, help="seconds the RCCL bring-up + first collectives may take")
    ap.add_argument("--no-traffic-pass", action="store_true", help="do not spawn the two rocprofv3 PMC passes that measure roofline.traffic")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3"],
                    help="fp32 = exact (headline); bf16x3 = opt-in split-precision channel mixing")
    return ap.parse_args()


def mk_args(cfg, steps):
    return SimpleNamespace(mdm_condm="text", latent_dim=512, ff_size=1024, layers=8, cond_mask_prob=0.1,
                           arch="trans_enc", emb_trans_dec=False, dataset="humanml", lang_model=None, mlpact="silu",
                           diffusion_steps=steps, noise_schedule="cosine", sigma_small=True, lambda_vel=1.0,
                           lambda_rcxyz=0.0, lambda_fc=0.0, njoints=cfg.njoints)


def step_kernel_label(tm):
    """What the step loop ran on, from ls_timing (the engine plans per prepared batch: fused rounds, one-pass-per-workgroup, sample-split,
    batch-level kernels, or up to three of them for a batch that does not fill the chip a whole number of times)."""
    names = {0: "ls::k_step (fused CFG denoiser + sampler update, one workgroup per clip, 1 launch/step)",
             1: "batch-level kernels (ls_long.hip, 21 launches/step)",
             2: "ls::k_coop (sample-split: {s} slice workgroups per (clip, CFG pass), 1 launch per step and resident set)",
             3: "ls::k_pass (one workgroup per (clip, CFG pass), two per CU, 1 launch/step)"}
    names[2] = names[2].format(s=tm.get("coop_slices") or "8 / 4 / 2")
    s = names.get(tm["step_path"], "?")
    for n, p in ((tm.get("tail_samples", 0), tm.get("tail_path", 0)), (tm.get("tail2_samples", 0), tm.get("tail2_path", 0))):
        if n:
            s += f" + {n} clips on {names.get(p, '?').split(' (')[0]}"
    return s


def pmc_traffic(dataset, B):
    """HBM bytes per k_step launch from the committed rocprofv3 PMC passes -- only if they were taken on THIS kernel source
    (the file stores the hash of ls_step_kernel.h it was measured on); otherwise null rather than a stale constant."""
    try:
        rec = json.load(open(PMC_TRAFFIC_FILE))
        src = open(os.path.join(ROOT, "livelyspeaker_amd", "csrc", "ls_step_kernel.h"), "rb").read()
        ent = rec["entries"].get(f"{dataset}:{B}")
        if ent and ent["kernel_source_sha256"] == hashlib.sha256(src).hexdigest():
            return ent["bytes_per_launch"], ent["source"]
        return None, f"{os.path.relpath(PMC_TRAFFIC_FILE, ROOT)} has no entry for this kernel source (re-profile)"
    except Exception as e:
        return None, f"unavailable: {e!r}"[:120]


def measure_traffic(a, B):
    """HBM-side bytes per k_step launch measured IN THIS RUN: two rocprofv3 PMC passes (one counter per run, --kernel-trace only, as
    MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass) of a 40-launch slice of this very workload in
    a child process; traffic = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes; the x2 is the guide's gfx950 correction for wide
    coalesced reads, which is how the weight images and static rows are fetched).  Returns (bytes | None, provenance)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--diffusion-steps", "40", "--no-cpu-baseline",
             "--no-parity", "--legs", "none", "--no-traffic-pass", "--dataset", a.dataset, "--batch", str(B), "--scale", str(a.scale)]
    kib, launches = {}, 0
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ls_pmc_", dir="/tmp")
        try:
            r = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "pmc", "--"] + child, capture_output=True, text=True,
                               timeout=420, env=env, cwd="/tmp")
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-160:]!r}"
            cur = sqlite3.connect(dbs[0]).cursor()
            # the step kernel(s) of the plan: fused (k_step), one-pass-per-workgroup (k_pass), sample-split (k_coop); summed per launch index
            ids = [did for name, did in cur.execute("select name, dispatch_id from kernels") if any(k in name for k in ("k_step", "k_pass", "k_coop"))]
            per = {}
            for did, cname, val in cur.execute("select dispatch_id, counter_name, value from counters_collection"):
                if cname == counter:
                    per[did] = per.get(did, 0.0) + val
            vals = [per[i] for i in ids if i in per]
            if not vals:
                return None, f"no {counter} rows for the step kernels (ls::k_step / k_pass / k_coop) in the PMC pass"
            kib[counter], launches = sum(vals) / len(vals), len(vals)
        except Exception as e:
            return None, f"traffic pass unavailable: {e!r}"[:200]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(round((2.0 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024)), (
        f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (one counter per pass, --kernel-trace only) over {launches} "
        f"step-kernel launches of a child process running this workload; 2 x FETCH_SIZE ({kib['FETCH_SIZE']:.0f} KiB) + WRITE_SIZE "
        f"({kib['WRITE_SIZE']:.0f} KiB)")


def cpu_baseline(cfg, args):
    """SURVEY.md 8(d) "CPU baseline timing": the torch-CPU port of the reference algorithm (oracle/rag_torch_cpu.py, pinned to
    reference-generated fixtures) on this box's host cores.  Steps are homogeneous, so >= 20 hoisted steps at the headline batch
    are timed (after one warm-up step) and scaled linearly; the reference-faithful mode (audio encoder re-run in both passes of
    every step, as the reference does) is timed separately; plus one full config-1 loop (B=4, 50-step DDPM).  Reported next to
    the GPU number; it is a baseline, not the target."""
    import torch
    from livelyspeaker_amd import synth
    from oracle import rag_oracle as orc
    from oracle.rag_torch_cpu import TorchCpuSampler
    port = TorchCpuSampler(synth.make_state_dict(cfg), cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens)
    # Thread count: aten's intra-op pool at the box's full core count (128 here) is far past its sweet spot for these shapes
    # (measured: 7.7 s/step at 128 threads vs ~1 s at 8-32), so calibrate on a small sample and use -- and report -- the best.
    default_threads = torch.get_num_threads()
    cal = {}
    yc = port._y(synth.make_cond(cfg, 128, scale=args.scale))
    tc = synth.NoiseTape(cfg, 128, 3)
    schc = orc.Schedule(args.diffusion_steps, args.respacing)
    for n in sorted({t for t in (8, 16, 32, 64, default_threads) if t <= default_threads}):
        torch.set_num_threads(n)
        port.sample_loop(schc, yc, tc.x_init, tc.eps, tc.noise, ddim=args.respacing.startswith("ddim"), hoisted=True, max_steps=1)
        t0 = time.perf_counter()
        port.sample_loop(schc, yc, tc.x_init, tc.eps, tc.noise, ddim=args.respacing.startswith("ddim"), hoisted=True, max_steps=2)
        cal[n] = time.perf_counter() - t0
    threads = min(cal, key=cal.get)
    torch.set_num_threads(threads)
    B, n_h, n_f = args.batch, 20, 2
    ddim = args.respacing.startswith("ddim")
    sch = orc.Schedule(args.diffusion_steps, args.respacing)
    y = port._y(synth.make_cond(cfg, B, scale=args.scale))
    tape = synth.NoiseTape(cfg, B, n_h + 1)
    port.sample_loop(sch, y, tape.x_init, tape.eps, tape.noise, ddim=ddim, hoisted=True, max_steps=1)       # warm-up (oneDNN primitives)
    t0 = time.perf_counter()
    port.prepare(y)
    t_prep = time.perf_counter() - t0
    t0 = time.perf_counter()
    port.sample_loop(sch, y, tape.x_init, tape.eps, tape.noise, ddim=ddim, hoisted=True, max_steps=n_h)
    t_h = max((time.perf_counter() - t0 - t_prep) / n_h, 1e-9)                                               # the loop re-runs prepare once
    port.sample_loop(sch, y, tape.x_init, tape.eps, tape.noise, ddim=ddim, hoisted=False, max_steps=1)       # warm-up
    t0 = time.perf_counter()
    port.sample_loop(sch, y, tape.x_init, tape.eps, tape.noise, ddim=ddim, hoisted=False, max_steps=n_f)
    t_f = (time.perf_counter() - t0) / n_f
    c1 = synth.TED if cfg.name == "ted" else cfg
    sch1 = orc.Schedule(50, "")
    tape1 = synth.NoiseTape(c1, 4, 50)
    y1 = port._y(synth.make_cond(c1, 4, scale=1.5))
    port.sample_loop(sch1, y1, tape1.x_init, tape1.eps, tape1.noise, hoisted=False, max_steps=2)
    t0 = time.perf_counter()
    port.sample_loop(sch1, y1, tape1.x_init, tape1.eps, tape1.noise, hoisted=False)
    t_c1 = time.perf_counter() - t0
    n_exec = sch.num_timesteps - args.skip
    frames = B * cfg.nframes
    return {"value": round(frames / (t_prep + n_exec * t_h), 3), "unit": "pose-frames/s", "cores": int(threads), "kind": "port",
            "sample": f"torch-CPU port (oracle/rag_torch_cpu.py), {threads} intra-op threads (best of a calibration over "
                      f"{ {k: round(v, 2) for k, v in cal.items()} } s per [prepare + 2 steps at B=128]; torch's default here is {default_threads}), "
                      f"B={B}: 1 prepare ({t_prep:.2f} s) + "
                      f"{n_h} hoisted steps ({t_h * 1e3:.0f} ms/step) scaled to {n_exec} steps; reference-faithful mode (audio encoder "
                      f"re-run 2x/step) {n_f} steps at {t_f * 1e3:.0f} ms/step; full config-1 loop (B=4, 50-step DDPM, faithful) {t_c1:.2f} s. "
                      f"The REAL reference measured in the build container (8 threads, SURVEY.md [probe]): 6.04 s/step at B=512 = "
                      f"2.9 pose-frames/s, config 1 in 1.64 s",
            "hoisted_ms_per_step": round(t_h * 1e3, 1), "reference_faithful_ms_per_step": round(t_f * 1e3, 1),
            "reference_faithful_value": round(frames / (n_exec * t_f), 3), "config1_loop_s": round(t_c1, 3),
            "threads_default": default_threads, "threads_calibration_s": {str(k): round(v, 3) for k, v in cal.items()},
            "reference_probe_8_threads": {"s_per_step_b512": 6.04, "pose_frames_per_s": 2.9, "config1_loop_s": 1.64}}


def parity_in_run(cfg, a, out_first, seed, sample_offset, y_np, n_pick):
    """Replay `n_pick` samples of the first TIMED call (same Philox seed / global sample indices, every diffusion step) through
    the CPU oracle fed with the numpy restatement of the device RNG, and return max|hip - oracle| (contract: 1e-3)."""
    from livelyspeaker_amd import synth
    from oracle import philox_oracle as po
    from oracle import rag_oracle as orc
    B = out_first.shape[0]
    pick = np.unique(np.linspace(0, B - 1, n_pick).round().astype(int))
    sch = orc.Schedule(a.diffusion_steps, a.respacing)
    n_exec = sch.num_timesteps - a.skip
    oracle = orc.RagOracle(synth.make_state_dict(cfg), cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, nframes=cfg.nframes)
    t0 = time.perf_counter()
    gidx = sample_offset + pick
    eps, noise = po.step_tapes(seed, gidx, n_exec, (cfg.njoints, cfg.nfeats, cfg.nframes))
    x_T = po.x_init(seed, gidx, cfg.jf, cfg.nframes, (cfg.njoints, cfg.nfeats))
    want = orc.sample_loop(oracle, sch, {k: v[pick] for k, v in y_np.items()}, x_T, eps, noise, ddim=a.respacing.startswith("ddim"),
                           skip_timesteps=a.skip)
    d = float(np.abs(out_first[pick].astype(np.float64) - want).max())
    return {"max_abs_diff": float(f"{d:.3e}"), "tolerance": 3e-4, "contract": 1e-3, "ok": bool(d <= 3e-4),
            "samples": [int(p) for p in pick], "steps_replayed": n_exec,
            "what": "first timed call (Philox noise, hipGraph) vs oracle/rag_oracle.py on oracle/philox_oracle.py's restated noise",
            "oracle_seconds": round(time.perf_counter() - t0, 1)}


def train_leg(a, cfg, model, diffusion, dev, world, rank, use_dist, fence):
    """Time TrainLoop.run_step-equivalent work: forward + losses + backward (ls_train_forward_backward), gradient
    all-reduce over the ranks (RCCL, only when world > 1), AdamW.  Batch per GPU = --batch (reference default 512)."""
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
    # a training run is thousands of steps: the steady state is what is timed.  8 untimed steps first -- the shader clock needs a few
    # hundred ms of load to reach its ceiling (docs/DESIGN_NOTES_r1-r3.md §3.2 (a)); 1 warm-up + 4 steps read 7.07 ms where the steady state is 6.85
    n, nwarm = 12, 8
    fwd = bwd = 0.0
    terms = None
    for i in range(n + nwarm):
        if i == nwarm:
            fence()
            t0 = time.perf_counter()
        terms = tr.forward_backward(x_start, t, noise, y, drop, eps)
        allreduce_mean_(tr.grad)
        tr.adamw()
        if i >= nwarm:
            fwd += terms["fwd_ms"]
            bwd += terms["bwd_ms"]
    fence()
    el = time.perf_counter() - t0
    if use_dist:
        from livelyspeaker_amd import shard
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        shard.all_reduce_(tt, torch.distributed.ReduceOp.MAX)
        el = float(tt.item())
    tr.close()
    cpu = None
    if world == 1 and not a.no_cpu_baseline:
        # the torch-CPU oracle (autograd) on this box's host cores, bounded sample: B=32, 2 steps after one warm-up
        from oracle import train_oracle as tro
        # aten's intra-op pool at the box's full core count (128) is past its sweet spot and very noisy for these shapes (13.9 vs
        # 32.2 samples/s between two runs of one build in round 2): 16 threads, two warm-up steps, six timed ones, median reported
        default_threads = torch.get_num_threads()
        nthr = min(16, default_threads)
        torch.set_num_threads(nthr)
        try:
            orc_t = tro.TrainOracle({k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if not k.endswith(".pe")},
                                    cfg.n_prefix_tokens)
            xs, yy, nz, dr, ep = synth.make_train_batch(cfg, 32, 0)
            tt0 = np.random.Generator(np.random.PCG64(0)).integers(0, 1000, size=(32,))
            per = []
            for i in range(8):
                c0 = time.perf_counter()
                orc_t.optimizer_step(orc_t.forward_backward(xs, tt0, nz, yy, dr, ep)[2])
                if i >= 2:
                    per.append(time.perf_counter() - c0)
        finally:
            torch.set_num_threads(default_threads)
        med = float(np.median(per))
        cpu = {"value": round(32 / med, 2), "unit": "samples/s", "cores": nthr, "kind": "port",
               "sample": f"torch-CPU oracle (oracle/train_oracle.py), B=32, {nthr} intra-op threads (torch's default here: {default_threads}), "
                         f"2 warm-up + 6 timed steps, median step {med * 1e3:.0f} ms (min {min(per) * 1e3:.0f}, max {max(per) * 1e3:.0f})"}
    return {"cpu_baseline": cpu, "what": "forward + Huber/velocity/KLD losses + backward + AdamW (ls_train_*), fp32, inputs resident in HBM",
            "value": round(world * B * n / el, 1), "unit": "samples/s", "ms_per_step": round(el / n * 1e3, 3),
            "fwd_ms": round(fwd / n, 3), "bwd_ms": round(bwd / n, 3), "batch_per_gpu": B, "timed_steps": n, "untimed_steps_before": nwarm,
            "gradient_allreduce": "RCCL, 1 bucket of 16 MB, averaged" if world > 1 else "none (1 GPU)",
            "loss_after": round(terms["total"], 5),
            "parity": "gradients within 8e-6 (rel. to max|g|) of the reference-pinned oracle, tests/test_gpu_train.py"}


def _train_schedule(a):
    from livelyspeaker_amd.model_util import create_gaussian_diffusion
    return create_gaussian_diffusion(SimpleNamespace(diffusion_steps=1000, noise_schedule="cosine", sigma_small=True, lambda_vel=1.0,
                                                     lambda_rcxyz=0.0, lambda_fc=0.0), "")


def livelyspeaker_leg(cfg, model_sd, dev, B, fence, reps=3, world=1, rank=0, max_over_ranks=lambda v: v):
    """BASELINE configs[2] as the reference runs it (scripts/test_LivelySpeaker_ted.py:85-113): SAG decoder on a synthetic CLIP text
    feature -> init_image -> CFG RAG refine with ddim100, skip_timesteps=80 (20 steps), guidance 2.5.  Own roofline object.
    On N ranks (weak scaling, B clips per rank): the frozen CLIP text features of the GLOBAL batch exist on rank 0 only (one text
    encoder) and are broadcast over RCCL inside the timed region -- the north star's "RCCL broadcast of the frozen CLIP text
    embeddings" -- then every rank decodes + refines its shard and the result is all-gathered."""
    import torch
    from livelyspeaker_amd import shard, synth
    from livelyspeaker_amd.cfg_sampler import ClassifierFreeSampleModel
    from livelyspeaker_amd.model_util import create_model_and_diffusion
    from livelyspeaker_amd.motionclip_module import Decoder_TRANSFORMER
    model, diffusion = create_model_and_diffusion(mk_args(cfg, 1000), "ddim100", dataset="ted")
    model.load_state_dict(model_sd, strict=False)
    model.to(dev)
    model.eval()
    model.cache_conditioning = False
    cfgm = ClassifierFreeSampleModel(model)
    diffusion.noise_source = "philox"
    diffusion.sample_offset = rank * B
    sag = Decoder_TRANSFORMER(latent_dim=512, n_pre_poses=4, use_style=False)
    sag.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_sag_state_dict(cfg).items()}, strict=False)
    sag.to(dev)
    sag.eval()
    total = world * B
    y_np = synth.make_cond(cfg, B, scale=2.5, seed=synth.SEED_COND + rank)
    y = {k: torch.from_numpy(v).to(dev) for k, v in y_np.items()}
    # rank 0 holds the text features of all `total` clips (a device tensor, as a CLIP text encoder would leave them); the other
    # ranks hold an empty buffer of that shape
    z_global = (torch.from_numpy(synth.make_text_features(total)) if rank == 0 else torch.zeros(total, 512)).to(dev)
    batch = {"x": y["origin_x"].clone(), "mask": torch.ones(B, 34, device=dev).bool(), "z": None}
    bcast = {"ms": 0.0, "n": 0}

    def call(overlap=True):
        # the once-per-call stage of the refinement needs nothing from the SAG decode: enqueued first (RAG.prefetch_condition ->
        # ls_prepare_async), it runs on the engine's stream while the decoder runs on its own
        if overlap:
            cfgm.prefetch_condition(y)
        if world > 1:
            t0 = time.perf_counter()
            zg = shard.broadcast_tensor(z_global, dev)              # RCCL broadcast (gloo on host copies in the shared-device test mode)
            torch.cuda.synchronize()
            bcast["ms"] += (time.perf_counter() - t0) * 1e3
            bcast["n"] += 1
            batch["z"] = zg[rank * B:(rank + 1) * B]
        else:
            batch["z"] = z_global
        decoded = sag(batch)["output"]
        out = diffusion.ddim_sample_loop(cfgm, (B, cfg.njoints, cfg.nfeats, cfg.nframes), clip_denoised=False, model_kwargs={"y": y},
                                         skip_timesteps=80, init_image=decoded, progress=False, dump_steps=None, noise=None,
                                         const_noise=False)
        return shard.gather_samples(out, total) if world > 1 else out
    call()
    eng, seng = model.engine(), sag.engine()
    bcast["ms"], bcast["n"] = 0.0, 0
    fence()
    t0 = time.perf_counter()
    loop_ms = prep_ms = 0.0
    launches = 0
    for _ in range(reps):
        out = call()
        tm = eng.timing()
        loop_ms += tm["loop_ms"]
        prep_ms += tm["prepare_ms"]
        launches += tm["n_step_launches"]
    fence()
    el = max_over_ranks((time.perf_counter() - t0) / reps)
    # (checked at the END of the leg: an exception on one rank in the middle of a sequence of collectives would leave the others waiting)
    ok_finite, ok_shape = bool(torch.isfinite(out).all()), out.shape[0] == total
    ok_bcast = not (world > 1 and rank == 0) or float(out[B:].abs().sum()) > 0     # rank r's gathered shard depends on the broadcast features
    kernel_ms = loop_ms / max(launches, 1)
    ach = 2 * FLOP_PER_FORWARD["ted"] * B / (kernel_ms * 1e-3) / 1e12
    sag_ms = seng.last_decode_ms()
    broadcast_ms = bcast["ms"] / max(bcast["n"], 1)
    # the same call in the reference's order (decode, THEN the once-per-call stage inside the sampling call), for the serial split
    call(overlap=False)
    fence()
    t1 = time.perf_counter()
    for _ in range(reps):
        call(overlap=False)
    fence()
    el_serial = max_over_ranks((time.perf_counter() - t1) / reps)
    sag_serial, prep_serial = seng.last_decode_ms(), eng.timing()["prepare_ms"]
    steady = None
    if world == 1:
        # The reference iterates loader batches (scripts/test_LivelySpeaker_ted.py:57-113): steady state over NB different batches back to
        # back, pipelined ACROSS calls -- decode + once-per-call stage of batch n + 1 are enqueued (decoder's stream / the other of two
        # model replicas) before batch n's refinement loop is waited for -- against the same batches call by call; outputs compared bitwise.
        from livelyspeaker_amd import _lib
        NB = 8
        model2, _ = create_model_and_diffusion(mk_args(cfg, 1000), "ddim100", dataset="ted")
        model2.load_state_dict(model_sd, strict=False)
        model2.to(dev)
        model2.eval()
        model2.cache_conditioning = False
        reps2 = [cfgm, ClassifierFreeSampleModel(model2)]
        ys = [{k: torch.from_numpy(v).to(dev) for k, v in synth.make_cond(cfg, B, scale=2.5, seed=synth.SEED_COND + 100 + n).items()} for n in range(NB)]
        bts = [{"x": ys[n]["origin_x"].clone(), "mask": torch.ones(B, 34, device=dev).bool(),
                "z": torch.from_numpy(synth.make_text_features(B, seed=synth.SEED_COND + 2100 + n)).to(dev)} for n in range(NB)]
        ts, ss = torch.cuda.current_stream(dev).cuda_stream, sag.engine()._stream
        shape = (B, cfg.njoints, cfg.nfeats, cfg.nframes)

        def refine(m, n, dec):
            return diffusion.ddim_sample_loop(m, shape, clip_denoised=False, model_kwargs={"y": ys[n]}, skip_timesteps=80, init_image=dec,
                                              progress=False, dump_steps=None, noise=None, const_noise=False)

        def run_serial():
            outs = []
            for n in range(NB):
                cfgm.prefetch_condition(ys[n])
                outs.append(refine(cfgm, n, sag(bts[n])["output"]))
            return outs

        def run_piped():
            def stage(n):
                reps2[n % 2].prefetch_condition(ys[n])
                return sag(bts[n], wait=False)["output"]
            outs, dec = [], stage(0)
            for n in range(NB):
                _lib.stream_order(dev.index or 0, ss, ts)
                nxt = stage(n + 1) if n + 1 < NB else None
                outs.append(refine(reps2[n % 2], n, dec))
                dec = nxt
            return outs

        def timed(fn):
            diffusion.philox_seed = 20260930          # the same key for every call of both orders: outputs comparable bit for bit
            fn()
            fence()
            t = time.perf_counter()
            o = fn()
            fence()
            return (time.perf_counter() - t) / NB, o
        try:
            el_ser, o_ser = timed(run_serial)
            el_pip, o_pip = timed(run_piped)
            steady = {"batches": NB, "ms_per_call": round(el_pip * 1e3, 3), "value": round(B * cfg.nframes / el_pip, 1), "unit": "pose-frames/s",
                      "call_by_call_ms_per_call": round(el_ser * 1e3, 3),
                      "bitwise_equal_to_call_by_call": bool(all(torch.equal(a_, b_) for a_, b_ in zip(o_ser, o_pip))),
                      "call_frac": round((2 * FLOP_PER_FORWARD["ted"] * B * 20 / 1e12 / MFMA_F32_PEAK_TFLOPS) / (el_pip * 1e3) * 1e3, 4),
                      "what": "SAG decode + ls_prepare_async of batch n + 1 enqueued on their own streams (decoder handle / the other of two "
                              "RAG replicas) before batch n's 20-step refinement loop is waited for; 8 different batches back to back"}
        except Exception as e:      # noqa: BLE001
            steady = {"error": repr(e)[:300]}
        finally:
            diffusion.philox_seed = None
            model2.engine().close()
    eng.close()
    assert ok_finite and ok_shape and ok_bcast, (ok_finite, ok_shape, ok_bcast)
    return {"workload": f"TED LivelySpeaker: SAG decode (synthetic CLIP text feature) + CFG RAG refine, ddim100 with skip_timesteps=80 "
                        f"(20 DDIM steps, what scripts/test_LivelySpeaker_ted.py runs), batch {B} per GPU x {world}, guidance 2.5, Philox noise",
            "value": round(total * cfg.nframes / el, 1), "unit": "pose-frames/s", "ms_per_call": round(el * 1e3, 3), "n_gpus": world,
            "text_feature_broadcast": (None if world == 1 else
                                       {"ms": round(broadcast_ms, 3), "bytes": total * 512 * 4, "src": 0,
                                        "what": "frozen CLIP text features [global batch, 512] fp32 from rank 0 to all ranks, inside the timed call; "
                                                "followed by an all_gather of the refined clips"}),
            "overlap": "ls_prepare_async on the engine's stream under the SAG decode on its own stream; sag_decode_ms / prepare_ms below are "
                       "each stream's own span while the two share the GPU",
            "sag_decode_ms": None if sag_ms is None else round(sag_ms, 3), "prepare_ms": round(prep_ms / reps, 3),
            "refine_loop_ms": round(loop_ms / reps, 3), "denoise_steps": 20,
            "sag_plus_prepare_wall_ms": round(el * 1e3 - loop_ms / reps, 3),
            "steady_state": steady,
            "serial_order": {"ms_per_call": round(el_serial * 1e3, 3), "sag_decode_ms": None if sag_serial is None else round(sag_serial, 3),
                             "prepare_ms": round(prep_serial, 3)},
            "full_100_steps_note": "BASELINE words it as '100 DDIM steps'; `--respacing ddim100` runs that variant as the headline workload",
            "roofline": {"bound": "mfma", "kernel": "ls::k_step", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "kernel_ms": round(kernel_ms, 4),
                         "call_frac": round((2 * FLOP_PER_FORWARD["ted"] * B * 20 / 1e12 / MFMA_F32_PEAK_TFLOPS) / (el * 1e3) * 1e3, 4),
                         "call_frac_note": "k_step FLOPs of the 20 steps at the MFMA peak / whole-call wall time (SAG + prepare + loop + host)",
                         # the call's OTHER algorithmic work priced too (per clip): SAG decoder 0.55 GFLOP, audio encoder 175.0
                         # MFLOP + static projection 9.9 MFLOP + style 0.52 MFLOP once per call; with these in the numerator the ceiling is 1.0
                         "call_frac_all_flops": round(((2 * FLOP_PER_FORWARD["ted"] * 20 + 0.55e9 + 174999360 + 9.9e6 + 524288) * B / 1e12 / MFMA_F32_PEAK_TFLOPS)
                                                      / (el * 1e3) * 1e3, 4)}}


def other_config_leg(dataset, B, dev, fence, steps=1000, noise="philox"):
    """One more BASELINE config measured in the same run, compactly: a fresh model of that dataset, one warm-up and one timed
    `p_sample_loop` call (1000-step DDPM, CFG 1.5, Philox noise), own FLOP count.  `beat150` is the SYNTHETIC 150-frame variant."""
    import torch
    from livelyspeaker_amd import synth
    from livelyspeaker_amd.cfg_sampler import ClassifierFreeSampleModel
    from livelyspeaker_amd.model_util import create_model_and_diffusion
    cfg = synth.CONFIGS[dataset]
    model, diffusion = create_model_and_diffusion(mk_args(cfg, steps), "", dataset=dataset)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()}, strict=False)
    model.to(dev)
    model.eval()
    model.cache_conditioning = False
    cfgm = ClassifierFreeSampleModel(model)
    diffusion.noise_source = noise
    y = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_cond(cfg, B, scale=1.5).items()}
    shape = (B, cfg.njoints, cfg.nfeats, cfg.nframes)
    if noise == "torch_cpu":
        torch.manual_seed(233)
    call = lambda: diffusion.p_sample_loop(cfgm, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=0, init_image=None,
                                           progress=False, dump_steps=None, noise=None, const_noise=False)
    call()
    fence()
    t0 = time.perf_counter()
    out = call()
    fence()
    el = time.perf_counter() - t0
    assert bool(torch.isfinite(out).all())
    tm = model.engine().timing()
    km = tm["loop_ms"] / max(tm["n_step_launches"], 1)
    ach = 2 * FLOP_PER_FORWARD[dataset] * B / (km * 1e-3) / 1e12
    names = {0: "fused step kernel (one workgroup per clip)",
             1: ("x_t projection GEMM + one-launch mixer (ls_mix_kernel.h: the eight blocks, four slice workgroups per (clip, CFG pass), one launch per 32 clips, poseFinal in its tail as per-slice partial products) + "
                 "update kernel: 3 launches per step" if (dataset == "beat150" and tm.get("coop_slices") == 4) else
                 "batch-level kernels (ls_long.hip, 21 launches per step)"),
             2: f"sample-split step kernel (ls_coop_kernel.h: {tm.get('coop_slices') or '8 / 4 / 2'} slice workgroups per (clip, CFG pass) in its first launch; one launch per step and resident set)",
             3: "one-pass-per-workgroup step kernel (ls_pass_kernel.h: a workgroup per (clip, CFG pass), two per CU)"}
    kernels = names[tm["step_path"]] + (f" + {tm['tail_samples']} clips on the {names[tm['tail_path']].split(' (')[0]}" if tm["tail_samples"] else "") \
        + (f" + {tm['tail2_samples']} clips on the {names[tm['tail2_path']].split(' (')[0]}" if tm["tail2_samples"] else "")
    seeds_extra = {}
    if noise == "torch_cpu":
        n_seg = int(diffusion.last_tape_segments)
        words = 2 * B * 512 + B * cfg.jf * cfg.nframes
        host = diffusion.last_host_rng_ms / steps
        seeds_extra = {"identical_seeds": {"host_rng_ms_per_step": round(host, 3), "upload_ms_per_step": round(tm["tape_upload_ms"] / steps, 3),
                                           "loop_ms_per_step": round(km, 4), "normals_per_step": words, "segments": n_seg,
                                           "native_stream": bool(getattr(diffusion, "last_host_rng_native", False)),
                                           "bound": ("host RNG" if host > 0.8 * km else "step kernel") + f": one sequential mt19937 stream of {words / 1e6:.2f} M "
                                                    f"normals per step takes {host:.2f} ms on the host (the [T][B][J][F]-ordered randn_like is torch's "
                                                    "per-element double Box-Muller: two 32-bit words per normal from ONE generator); the loop advanced at "
                                                    f"{km:.2f} ms per step -- a segment is drawn while the previous one runs, so a step costs the slower of "
                                                    "the host draws and the step kernel (the same workload on Philox noise: `configs4_beat`).  The words of "
                                                    "a long fill are already produced by generator threads behind a state-only scout (csrc/ls_torch_rng.cpp); "
                                                    "what is left is the scout's own sequential state recurrence and the double-precision transforms, "
                                                    "0.9 - 1.4 ms per step on the GPU hosts measured"}}
    model.engine().close()
    return {"workload": f"{dataset.upper()} RAG, batch {B} x {cfg.nframes} frames, {steps}-step DDPM, CFG 1.5, "
                        + ("noise_source='torch_cpu' (the reference's own draws in its order: 'identical seeds')" if noise == "torch_cpu" else "Philox noise")
                        + (" -- SYNTHETIC shape (150 frames: the reference cannot run it), perf-only, no parity claim vs the reference"
                           if dataset == "beat150" else ""),
            "value": round(B * cfg.nframes / el, 1), "unit": "pose-frames/s", "ms_per_call": round(el * 1e3, 2),
            "kernels": kernels + (" -- self-pinned: checked against this repository's oracle only" if dataset == "beat150" else ""),
            "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "step_ms": round(km, 4),
                         "flop_per_sample_step": 2 * FLOP_PER_FORWARD[dataset]}, **seeds_extra}


def _self_launch(a):
    """`python bench.py --gpus N` (N > 1) without a launcher: re-exec this script under torch.distributed.run, one rank per GPU,
    rendezvous on 127.0.0.1 at a free port.  The ranks inherit this process's stdout, so rank 0's ONE JSON line (written to its
    private duplicate of fd 1, see _reserve_stdout) IS this process's stdout; everything else the ranks print goes to stderr."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on these hosts (RCCL P2P setup)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


def threads_main(a):
    """`--launcher threads`: the N-device run without any process group.  One process, N engine handles (one per device; all on
    cuda:0 with --ranks-share-device), N Python threads -- ctypes releases the GIL inside ls_sample, so the devices run concurrently.
    The global batch is split with ls_shard_range exactly like the torchrun path (shard r = rank r's batch, Philox streams keyed by
    the global sample index through sample_offset), the results are concatenated on the host, and shard 1 is re-generated on device
    0 as the cross-check.  Same barrier -> K calls -> barrier timing, max over threads = the wall time of the slowest device."""
    global _RESULT_OUT
    _RESULT_OUT = _reserve_stdout()
    import ctypes
    import threading
    import torch
    from livelyspeaker_amd import _lib, synth
    world = a.gpus
    ndev = torch.cuda.device_count()
    if not a.ranks_share_device and world > ndev:
        raise SystemExit(f"--launcher threads --gpus {world}: only {ndev} GPU(s) visible (add --ranks-share-device to test on one)")
    cfg = synth.CONFIGS[a.dataset]
    strong = a.global_batch > 0
    total = a.global_batch if strong else world * a.batch
    lib = _lib.load_library()
    spans = []
    for r in range(world):
        f, c = ctypes.c_int64(), ctypes.c_int64()
        assert lib.ls_shard_range(total, world, r, ctypes.byref(f), ctypes.byref(c)) == 0
        spans.append((f.value, c.value))
    sd = synth.make_state_dict(cfg, seed=synth.SEED_WEIGHTS)
    sch = synth.schedule(a.diffusion_steps, a.respacing)
    ddim = a.respacing.startswith("ddim")
    sampler = _lib.LS_SAMPLER_DDIM if ddim else _lib.LS_SAMPLER_DDPM
    engines = []
    for r in range(world):
        e = _lib.Engine(cfg.njoints, cfg.nfeats, cfg.n_prefix_tokens, cfg.audio_len, n_emotions=cfg.n_emotions,
                        device=0 if a.ranks_share_device else r, path=a.path)
        e.load_state_dict(sd)
        e.set_schedule(sch)
        if a.precision != "fp32":
            e.set_precision(a.precision)
        engines.append(e)
    conds = [synth.make_cond(cfg, spans[r][1], scale=a.scale, seed=synth.SEED_COND + r) for r in range(world)]
    seed = 20260929
    outs, times, errs = [None] * world, [0.0] * world, []
    bar = threading.Barrier(world)

    def worker(r, n, keep):
        try:
            e = engines[r]
            bar.wait()
            t0 = time.perf_counter()
            for i in range(n):
                e.prepare(conds[r])
                o = e.sample(sampler=sampler, philox_seed=seed, sample_offset=spans[r][0], skip_timesteps=a.skip, use_graph=not a.no_graph)
                if keep and i == 0:
                    outs[r] = o
            bar.wait()
            times[r] = time.perf_counter() - t0
        except Exception as ex:         # noqa: BLE001
            errs.append(repr(ex)[:300])
            bar.abort()

    def run(n, keep):
        ths = [threading.Thread(target=worker, args=(r, n, keep)) for r in range(world)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise SystemExit(f"threads launcher: {errs}")
        return max(times)

    if a.warmup:
        run(a.warmup, False)
    elapsed = run(a.steps, True)
    whole = np.concatenate(outs, axis=0)
    assert whole.shape[0] == total and np.isfinite(whole).all()
    # cross-check: device 0 re-generates the NEXT shard by itself (same Philox key, that shard's conditioning, sample_offset)
    nb = 1 % world
    engines[0].prepare(conds[nb])
    redo = engines[0].sample(sampler=sampler, philox_seed=seed, sample_offset=spans[nb][0], skip_timesteps=a.skip, use_graph=not a.no_graph)
    diff = float(np.abs(redo - outs[nb]).max())
    tm = engines[0].timing()
    n_exec = sch.num_timesteps - a.skip
    rec = {"metric": "pose-frames/sec denoised", "value": round(total * cfg.nframes * a.steps / elapsed, 2), "unit": "pose-frames/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
           "scaling": "strong" if strong else "weak", "vs_baseline": None,
           "dtype": "f32" if a.precision == "fp32" else "bf16x3 split (fp32 accumulate) for channel mixing, f32 elsewhere", "data": "synthetic",
           "launcher": "threads: one process, one engine handle per device, one Python thread per handle (ctypes releases the GIL); no process "
                       "group, no collective -- shards by ls_shard_range + sample_offset, results concatenated on the host",
           "config": {"workload": f"{a.dataset.upper()} RAG, batch {spans[0][1]} x {cfg.nframes} frames per GPU, {n_exec}-step {'DDIM' if ddim else 'DDPM'}, "
                                  f"CFG scale {a.scale}, random-init weights + synthetic conditioning, Philox noise on device",
                      "global_batch": total, "frames": cfg.nframes, "denoise_steps": n_exec, "guidance_scale": a.scale,
                      "parallelism": f"batch-sharded x{world}, no collective", "hipgraph": not a.no_graph},
           "collective_backend": "none", "rccl_ranks": 0,
           "shard_check": {"what": "device 0 re-generated the next shard via sample_offset and compared it with that handle's result",
                           "max_abs_diff": diff, "bitwise_equal": diff == 0.0, "ranks": world},
           "roofline": {"bound": "mfma", "kernel_ms_device0": round(tm["loop_ms"] / max(tm["n_step_launches"], 1), 4), "peak": MFMA_F32_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "note": "per-device kernel time of the last call on device 0; the headline roofline object is the torchrun launcher's"}}
    if a.ranks_share_device:
        rec["ranks_share_device"] = "TEST MODE: every handle on cuda:0 (the handles time-share one GPU); `value` is not a scaling number"
    for e in engines:
        e.close()
    _RESULT_OUT.write(json.dumps(rec) + "\n")
    _RESULT_OUT.flush()
    return 0


def main():
    global _RESULT_OUT
    a = parse()
    if a.launcher == "threads":
        return threads_main(a)
    if a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(_self_launch(a))
    _RESULT_OUT = _reserve_stdout()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on these hosts: RCCL's P2P setup needs it, whoever launched us
    import torch
    import torch.distributed as dist
    from livelyspeaker_amd import shard, synth
    from livelyspeaker_amd.cfg_sampler import ClassifierFreeSampleModel
    from livelyspeaker_amd.model_util import create_model_and_diffusion

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but the launcher started {world} rank(s): pass --gpus {world} (or run plain "
                         f"`python bench.py --gpus {a.gpus}`, which launches its own ranks)")
    if a.ranks_share_device:
        local = 0
    elif local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = "RANK" in os.environ          # launched by torch.distributed.run (also with a single rank)
    backend, groups = "none", None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # gloo default group (control plane; a rank that never arrives ends the run with an error after 15 minutes instead of holding
        # the node) + an RCCL group for the data collectives that is PROBED first: if RCCL does not come up, or its first collectives
        # fail or hang, every rank stays on gloo with its own GPU and the line says so (collective_backend, rccl_ranks: 0, rccl_error)
        groups = shard.init_groups(dev, rank, world, want_rccl=not a.no_rccl, probe_timeout_s=a.rccl_probe_timeout)
        backend = groups["collective_backend"]

    cfg = synth.CONFIGS[a.dataset]
    strong = a.global_batch > 0
    if strong:
        first, B = shard.shard_range(a.global_batch, world, rank)
        if a.global_batch % world:
            raise SystemExit("--global-batch must be divisible by the number of ranks")
    else:
        first, B = rank * a.batch, a.batch
    total = a.global_batch if strong else world * B
    model, diffusion = create_model_and_diffusion(mk_args(cfg, a.diffusion_steps), a.respacing, dataset=a.dataset)
    # random-init weights of the architecture, re-drawn with non-degenerate scale (the reference's own init
    # zeroes the channel-mix weights); rank 0's copy is broadcast over RCCL so all ranks agree.
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=synth.SEED_WEIGHTS + (0 if rank == 0 else 1)).items()}
    if use_dist:
        sd = shard.broadcast_state_dict(sd, dev)
    model.load_state_dict(sd, strict=False)
    model.to(dev)
    model.eval()
    model.precision = a.precision
    if a.path != "auto":
        model.step_path = a.path
    model.cache_conditioning = False        # every timed call re-runs the once-per-call stage (a new batch)
    cfgm = ClassifierFreeSampleModel(model)
    diffusion.noise_source = "philox"
    diffusion.use_graph = not a.no_graph
    diffusion.sample_offset = first         # Philox streams keyed by the global sample index (shard-invariant)

    # conditioning of the global batch = concatenation of per-shard draws (seed = SEED_COND + shard index), so any rank can
    # rebuild any shard; shard r of the strong-scaling run IS the batch rank r of the weak-scaling run would hold
    def cond_of(shard_index, scale):
        return synth.make_cond(cfg, B, scale=scale, seed=synth.SEED_COND + shard_index)

    y_np = cond_of(rank, a.scale)
    y = {k: torch.from_numpy(v).to(dev) for k, v in y_np.items()}
    shape = (B, cfg.njoints, cfg.nfeats, cfg.nframes)
    ddim = a.respacing.startswith("ddim")
    fn = diffusion.ddim_sample_loop if ddim else diffusion.p_sample_loop

    def one_call(yy=None):
        local_out = fn(cfgm, shape, clip_denoised=False, model_kwargs={"y": yy or y}, skip_timesteps=a.skip, init_image=None,
                       progress=False, dump_steps=None, noise=None, const_noise=False)
        if strong:
            return local_out, shard.gather_samples(local_out, total)       # all_gather_into_tensor over RCCL, inside the timed region
        return local_out, None

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if not use_dist:
            return v
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        shard.all_reduce_(t, dist.ReduceOp.MAX)
        return float(t.item())

    def timed(n, yy=None):
        """n calls bracketed by barrier + synchronize; max over ranks.  Returns (elapsed, loop_ms, launches, prep_ms, first_out, seed)."""
        loop_ms, launches, prep_ms, first_out, seed, gathered = 0.0, 0, 0.0, None, None, None
        fence()
        t0 = time.perf_counter()
        for i in range(n):
            out, gathered = one_call(yy)
            tm = eng.timing()
            loop_ms += tm["loop_ms"]
            launches += tm["n_step_launches"]
            prep_ms += tm["prepare_ms"]
            if i == 0:
                first_out, seed = out, diffusion.last_philox_seed
        fence()
        el = time.perf_counter() - t0
        el = max_over_ranks(el)
        return el, loop_ms, launches, prep_ms, first_out, seed, gathered

    torch.manual_seed(233)                  # every rank draws the same Philox key per call
    for _ in range(a.warmup):
        one_call()
    eng = model.engine()
    elapsed, loop_ms, launches, prep_ms, first_out, first_seed, gathered = timed(a.steps)
    assert first_out is not None and bool(torch.isfinite(first_out).all()), "non-finite samples"
    main_tm = eng.timing()                  # what the headline's step loop ran on (later legs re-plan the same engine)
    single_pass_main = bool(main_tm["single_pass"])

    # the config-4 premise, on hardware: rank r re-generates the shard of rank (r+1) % world on ITS OWN GPU from the same Philox key
    # with sample_offset, and compares it with what that rank produced (strong mode: the gathered tensor; weak mode: a P2P-free
    # all_gather of the per-rank results).  Bitwise equality expected: the streams depend on the global index only.
    shard_check = None
    if use_dist or strong:
        try:
            def regenerate(first_idx, count, shard_index):
                y_sh = {k: torch.from_numpy(v).to(dev) for k, v in cond_of(shard_index, a.scale).items()}
                diffusion.sample_offset, diffusion.philox_seed = first_idx, first_seed
                try:
                    return fn(cfgm, shape, clip_denoised=False, model_kwargs={"y": y_sh}, skip_timesteps=a.skip, init_image=None,
                              progress=False, dump_steps=None, noise=None, const_noise=False)
                finally:
                    diffusion.sample_offset, diffusion.philox_seed = first, None
            shard_check = shard.cross_check(regenerate, first_out, total, equal_shards_of=B)

        except Exception as e:          # symmetric on every rank (same code path); never take the headline line down
            shard_check = {"error": repr(e)[:300], "rccl_ranks": world if backend in ("nccl", "none") else 0, "ranks": world,
                           "collective_backend": backend}
            diffusion.sample_offset, diffusion.philox_seed = first, None

    # diagnostics of a multi-rank run, so that the first run on N real GPUs can be read from its one line: every rank's device, CU count,
    # kernel plan and own timings (collected over the gloo control plane), and what the result gather costs on the data group
    rank_recs = gather_rec = None
    if use_dist:
        try:
            prop = torch.cuda.get_device_properties(dev)
            rank_recs = shard.rank_reports({"device": f"cuda:{local}", "name": prop.name, "n_cus": int(main_tm["n_cus"]), "batch": B,
                                            "step_kernel": step_kernel_label(main_tm), "loop_ms_per_call": round(loop_ms / a.steps, 3),
                                            "prepare_ms_per_call": round(prep_ms / a.steps, 3), "collective_backend": shard.backend(),
                                            "rccl_error": (groups or {}).get("rccl_error")})
            gather_rec = shard.timed_gather(first_out, total)
        except Exception as e:          # noqa: BLE001  (symmetric on every rank)
            rank_recs, gather_rec = [{"error": repr(e)[:300]}], None

    extra = not a.no_extra_legs and a.precision == "fp32" and a.legs != "none"
    legs = set(("single", "split", "lively", "beat", "train", "seeds", "small", "mid") if a.legs in ("all", "none") else a.legs.split(","))

    # Secondary object: guidance scale 1 (what the reference's callers run): the uncond pass is legitimately skipped
    single = None
    if extra and a.scale != 1.0 and "single" in legs:
        y1 = dict(y, scale=torch.ones(B, device=dev))
        one_call(y1)
        n1 = max(1, min(a.steps, 2))
        e1, l1, k1, _, o1, _, _ = timed(n1, y1)
        assert bool(eng.timing()["single_pass"]) and bool(torch.isfinite(o1).all())
        km = l1 / max(k1, 1)
        ach1 = FLOP_PER_FORWARD[a.dataset] * B / (km * 1e-3) / 1e12
        single = {"workload": "same as the headline but guidance scale 1.0 (scripts/test_RAG_ted.py:183): out_u + 1*(out_c - out_u) = out_c, "
                              "so ONE forward per sample per step (two samples per workgroup); never mixed into `value`",
                  "value": round(total * cfg.nframes * n1 / e1, 2), "unit": "pose-frames/s", "ms_per_call": round(e1 / n1 * 1e3, 3),
                  "flop_per_sample_step": FLOP_PER_FORWARD[a.dataset],
                  "roofline": {"bound": "mfma", "achieved": round(ach1, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach1 / MFMA_F32_PEAK_TFLOPS, 4), "kernel_ms": round(km, 4)},
                  "parity": "reference fixture G11 (scale 1, both passes evaluated by the reference) <= 3e-4, tests/test_gpu_singlepass.py"}

    # Secondary leg (never the headline `value`): the opt-in bf16x3 split-precision mode, same workload, same run.
    split = None
    if extra and not a.no_split_leg and "split" in legs:
        model.precision = "bf16x3"
        one_call()
        n2 = max(1, min(a.steps, 2))
        e2, l2, k2, _, _, _, _ = timed(n2)
        split = {"mode": "bf16x3: channel/token mixing as 3 bf16 MFMAs per fp32 product (hi.hi+hi.lo+lo.hi), fp32 accumulate",
                 "value": round(total * cfg.nframes * n2 / e2, 2), "unit": "pose-frames/s",
                 "kernel_ms": round(l2 / max(k2, 1), 4),
                 "parity": "max-abs vs reference golden after 1000 DDPM steps 3.5e-5 (contract 1e-3), tests/test_gpu_edge.py",
                 "note": "opt-in (RAG.precision / ls_set_precision); the headline value above is the exact-fp32 path"}
        model.precision = "fp32"

    # Secondary object: the reference's own RNG contract ("identical seeds"): every draw of the loop made from torch's CPU generator
    # in the reference's order, in K-step segments through two page-locked buffers, uploaded under the previous segment's steps
    seeds = None
    if extra and world == 1 and "seeds" in legs and a.dataset != "beat150":
        try:
            diffusion.noise_source = "torch_cpu"
            torch.manual_seed(233)
            # torch's own generator makes ~90 ms of draws per step at this batch: then a bounded sample -- the LAST n_s steps of the schedule
            # (skip_timesteps, the reference's own argument), scaled linearly; the native stream (~1 ms per step) runs the whole schedule
            n_total = diffusion.num_timesteps - a.skip
            from livelyspeaker_amd import torch_rng as _trng
            n_s = n_total if _trng.variant() >= 0 else min(96, n_total)      # the native stream is fast enough for the whole schedule
            fence()
            t0 = time.perf_counter()
            o_s = fn(cfgm, shape, clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=diffusion.num_timesteps - n_s, init_image=None,
                     progress=False, dump_steps=None, noise=None, const_noise=False)
            fence()
            e_s = time.perf_counter() - t0
            tm = eng.timing()
            assert bool(torch.isfinite(o_s).all()) and tm["n_step_launches"] == n_s
            per_step = (2 * B * 512 + B * cfg.jf * cfg.nframes) * 4
            n_seg = int(diffusion.last_tape_segments)
            k_seg = -(-n_s // max(n_seg, 1))
            full_s = e_s * n_total / n_s
            seeds = {"workload": "same as the headline but noise_source='torch_cpu': x_T, two style eps and one randn_like(x) per step drawn from "
                                 "torch's CPU generator in the reference's order (gaussian_diffusion.py:700-743, RAG.py:120), so "
                                 "torch.manual_seed(s) reproduces the reference's CPU samples (fixture G7)",
                     "sample": (f"the whole schedule ({n_total} steps)" if n_s == n_total else
                                f"the last {n_s} of the {n_total} steps (skip_timesteps = {diffusion.num_timesteps - n_s}), scaled linearly to {n_total}"),
                     "value": round(total * cfg.nframes / full_s, 2), "unit": "pose-frames/s", "ms_per_call_scaled": round(full_s * 1e3, 1),
                     "measured_ms": round(e_s * 1e3, 1), "measured_steps": n_s,
                     "host_rng_ms_per_step": round(diffusion.last_host_rng_ms / n_s, 2), "upload_ms_per_step": round(tm["tape_upload_ms"] / n_s, 3),
                     "gpu_kernel_ms_per_step": round(loop_ms / max(launches, 1), 3), "segments": n_seg, "steps_per_segment": k_seg,
                     "tape_bytes_if_one_piece": per_step * n_total,
                     "pinned_host_bytes": 2 * k_seg * per_step if n_seg > 1 else per_step * n_s,
                     "host_rng": ("native restatement of torch's CPU normal stream (csrc/ls_torch_rng.cpp: mt19937 words in bulk by one producer thread -- "
                                  "vectorised block update and tempering -- the float and double Box-Muller transforms on a worker pool behind "
                                  "it), continued from and handed back to torch's generator state; checked "
                                  "bitwise against torch once per process" if getattr(diffusion, "last_host_rng_native", False) else
                                  "torch's own generator (the native restatement does not reproduce this torch build)"),
                     "bound": ("the slower of host RNG (one sequential mt19937 stream: 2 x B x 512 + B x J x F x T normals = 1.46 M words per step at "
                               "512 clips) and the step kernel: a segment is drawn while the previous segment's steps run, its upload rides the copy "
                               "stream"),
                     "hipgraph": False if n_seg > 1 else bool(diffusion.use_graph)}
        except Exception as e:
            seeds = {"error": repr(e)[:300]}
        finally:
            diffusion.noise_source = "philox"

    # BASELINE configs[0]'s shape on the GPU (4 clips, 50-step DDPM): a latency case -- the engine runs such batches on its batch-level
    # kernels (ls_set_path "auto"); the fused one-workgroup-per-sample kernel is timed beside it
    small = None
    if extra and world == 1 and "small" in legs and a.dataset == "ted":
        try:
            small = {}
            for path in ("auto", "fused"):
                m4, d4 = create_model_and_diffusion(mk_args(cfg, 50), "", dataset=a.dataset)
                m4.load_state_dict(sd, strict=False)
                m4.to(dev)
                m4.eval()
                m4.cache_conditioning = False
                m4.step_path = path
                c4 = ClassifierFreeSampleModel(m4)
                d4.noise_source = "philox"
                y4 = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_cond(cfg, 4, scale=1.5).items()}
                call4 = lambda: d4.p_sample_loop(c4, (4, cfg.njoints, cfg.nfeats, cfg.nframes), clip_denoised=False, model_kwargs={"y": y4},
                                                 skip_timesteps=0, init_image=None, progress=False, dump_steps=None, noise=None, const_noise=False)
                call4(); call4()
                fence()
                t0 = time.perf_counter()
                for _ in range(10):
                    o4 = call4()
                fence()
                e4 = (time.perf_counter() - t0) / 10
                tm4 = m4.engine().timing()
                small[path] = {"step_path": tm4["step_path"], "ms_per_call": round(e4 * 1e3, 3), "value": round(4 * cfg.nframes / e4, 1), "unit": "pose-frames/s",
                               "step_ms": round(tm4["loop_ms"] / max(tm4["n_step_launches"], 1), 4), "finite": bool(torch.isfinite(o4).all())}
                m4.engine().close()
            small["workload"] = "TED RAG, 4 clips x 34 frames, 50-step DDPM, CFG 1.5 (BASELINE configs[0]'s shape; Philox noise): `auto` = the engine's choice "
            small["workload"] += "(the sample-split kernel: 16 workgroups per clip, one launch per step), `fused` = one workgroup per clip"
        except Exception as e:
            small = {"error": repr(e)[:300]}

    lively = None
    if extra and a.dataset == "ted" and not strong and "lively" in legs:
        try:
            lively = livelyspeaker_leg(cfg, sd, dev, B, fence, world=world, rank=rank, max_over_ranks=max_over_ranks)
        except Exception as e:                      # never let a secondary leg take the headline line down
            lively = {"error": repr(e)[:300]}

    # BASELINE configs[4] in the same run (1 GPU only): BEAT at the reference's 34 frames with the whole 256-clip job on this GPU, and the
    # synthetic 150-frame variant at one GPU's share (256 / 8 = 32 clips)
    others = None
    if extra and a.dataset == "ted" and world == 1 and "beat" in legs:
        others = {}
        # (B = 32 = one GPU's share of the 256-clip job on 8 GPUs: at 34 frames the engine runs it on its batch-level kernels, section 3.8)
        for name, ds, bb in (("beat_34_frames_b256", "beat", 256), ("beat_34_frames_b32_one_gpu_share_of_256", "beat", 32),
                             ("beat150_synthetic_b32", "beat150", 32)):
            try:
                others[name] = other_config_leg(ds, bb, dev, fence)
            except Exception as e:
                others[name] = {"error": repr(e)[:300]}

    # Batches that are not a multiple of the chip (the reference's loaders end on ragged batches, scripts/test_RAG_ted.py:43-82): what the
    # engine's plan makes of them -- the one-pass-per-workgroup kernel alone or behind full fused rounds -- and BEAT B = 256 in the
    # reference's own RNG mode (2.7 M normals per step from one sequential host stream)
    mid = None
    if extra and a.dataset == "ted" and world == 1 and "mid" in legs:
        mid = {}
        for name, ds, bb, nz in (("ted_b64", "ted", 64, "philox"), ("ted_b80", "ted", 80, "philox"),
                                 ("ted_b128", "ted", 128, "philox"), ("ted_b384", "ted", 384, "philox"), ("ted_b160", "ted", 160, "philox"),
                                 ("beat_b256_identical_seeds", "beat", 256, "torch_cpu")):
            try:
                mid[name] = other_config_leg(ds, bb, dev, fence, noise=nz)
            except Exception as e:
                mid[name] = {"error": repr(e)[:300]}

    # Secondary leg: one optimisation step of the denoiser, data-parallel over the ranks.
    train = None
    if extra and (a.train_leg or world == 1) and not a.no_train_leg and not strong and "train" in legs:
        try:
            train = train_leg(a, cfg, model, diffusion, dev, world, rank, use_dist, fence)
        except Exception as e:
            train = {"error": repr(e)[:300]}

    if rank == 0:
        frames = total * cfg.nframes * a.steps
        n_exec = diffusion.num_timesteps - a.skip
        kernel_ms = loop_ms / max(launches, 1)
        passes = 1 if single_pass_main else 2
        flop_launch = passes * FLOP_PER_FORWARD[a.dataset] * B
        achieved = flop_launch / (kernel_ms * 1e-3) / 1e12
        # fp32: FP32-matrix MFMA peak.  bf16x3: three bf16 MFMAs per algorithmic product -> dense bf16 peak / 3.
        peak = MFMA_F32_PEAK_TFLOPS if a.precision == "fp32" else round(2500.0 / 3.0, 1)
        committed, committed_src = pmc_traffic(a.dataset, B)
        traffic, traffic_src = None, "not measured (--no-traffic-pass)"
        if world == 1 and not a.no_traffic_pass and a.dataset != "beat150" and a.precision == "fp32":
            traffic, traffic_src = measure_traffic(a, B)
        rec = {
            "metric": "pose-frames/sec denoised", "value": round(frames / elapsed, 2), "unit": "pose-frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32" if a.precision == "fp32" else "bf16x3 split (fp32 accumulate) for channel mixing, f32 elsewhere",
            "data": "synthetic" if a.dataset != "beat150" else "synthetic (shape too: 150 frames, which the reference cannot run -- no parity claim)",
            "config": {"workload": f"{a.dataset.upper()} RAG, batch {B} x {cfg.nframes} frames per GPU, "
                                   f"{n_exec}-step {'DDIM' if ddim else 'DDPM'} ({a.diffusion_steps} diffusion steps"
                                   f"{', respacing ' + a.respacing if a.respacing else ''}), CFG scale {a.scale}"
                                   f"{' (single pass: every scale is 1)' if single_pass_main else ''}, "
                                   f"random-init weights + synthetic audio/speaker/prefix-pose conditioning, Philox noise on device",
                       "global_batch": total, "frames": cfg.nframes, "denoise_steps": n_exec,
                       "guidance_scale": a.scale,
                       "parallelism": (f"global batch {total} sharded x{world}, all_gather of the result in the timed region" if strong
                                       else f"batch-sharded x{world}, no per-step collective"),
                       "hipgraph": bool(diffusion.use_graph)},
            "roofline": {"bound": "mfma", "kernel": (step_kernel_label(main_tm) if a.dataset != "beat150" else
                                                     ("long-sequence step, 3 launches: x_t projection GEMM, ls::k_mix (the eight blocks + poseFinal in one launch per 32 clips, four "
                                                      "slice workgroups per (clip, CFG pass)), update; achieved = algorithmic FLOPs / mean step time"
                                                      if main_tm.get("coop_slices") == 4 else
                                                      "long-sequence step: 21 launches/step (batch-level kernels: GEMMs on ls::k_gemm_tr + fused token-mixing / LayerNorm-partial "
                                                      "kernels); achieved = algorithmic FLOPs / mean step time")),
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4),
                         "traffic": traffic, "traffic_unit": "B/launch (rocprofv3 PMC)", "traffic_source": traffic_src,
                         "traffic_from_committed_profile": committed, "traffic_from_committed_profile_source": committed_src,
                         "kernel_ms": round(kernel_ms, 4), "flop_per_launch": flop_launch,
                         "prepare_ms_per_call": round(prep_ms / a.steps, 3)},
        }
        if rank_recs is not None:
            rec["ranks"] = rank_recs
            rec["gather"] = gather_rec
        if shard_check is not None:
            rec["shard_check"] = shard_check
            rec["rccl_ranks"] = shard_check["rccl_ranks"]
            rec["collective_backend"] = backend
            if groups is not None and groups.get("rccl_error"):
                rec["rccl_error"] = groups["rccl_error"]
            if a.ranks_share_device:
                rec["ranks_share_device"] = ("TEST MODE: all ranks on cuda:0 (RCCL rejects duplicate devices, so its probe fails and the "
                                             "collectives fall back to gloo on host copies -- the fallback path itself); `value` is not a scaling number")
        if not a.no_parity:
            try:
                rec["parity_in_run"] = parity_in_run(cfg, a, first_out.detach().cpu().numpy(), first_seed, first, y_np, a.parity_samples)
            except Exception as e:
                rec["parity_in_run"] = {"error": repr(e)[:300]}
        if single is not None:
            rec["single_pass"] = single
        if seeds is not None:
            rec["identical_seeds_mode"] = seeds
        if small is not None:
            rec["config1_shape"] = small
        if lively is not None:
            rec["livelyspeaker"] = lively
        if others is not None:
            rec["configs4_beat"] = others
        if mid is not None:
            rec["mid_batches"] = mid
        if split is not None:
            rec["split_precision"] = split
        if train is not None:
            rec["train_step"] = train
        if world == 1 and not a.no_cpu_baseline:
            try:
                rec["cpu_baseline"] = cpu_baseline(cfg, a)
            except Exception as e:
                rec["cpu_baseline"] = {"error": repr(e)[:300]}
        _RESULT_OUT.write(json.dumps(rec) + "\n")
        _RESULT_OUT.flush()
    if use_dist:
        dist.barrier()
        shard.finish(0)        # destroy_process_group -- or os._exit when an RCCL probe thread had to be abandoned (said on stderr)


def _reserve_stdout():
    """Keep stdout for the ONE JSON line: libraries below us write there too (RCCL prints a version banner at communicator
    creation), so fd 1 is pointed at stderr for the run and the result goes to a private duplicate of the original stdout."""
    sys.stdout.flush()
    out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return out


if __name__ == "__main__":
    main()
