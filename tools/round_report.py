#!/usr/bin/env python3
"""Assemble profiles/<name>.md from what tools/round_measure.sh left under gpurun_out/<dir>/ (bench lines, kernel trace, PMC passes).

    python tools/round_report.py gpurun_out/r02d profiles/r02d_final_measurements.md "Round 2, run D (final build of the round)"
"""
import json, os, sys

src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
RUNS = [
    ("bench_default", "(defaults: TED, B=512, 1000-step DDPM, CFG 1.5)"),
    ("bench_strong_n1", "`torch.distributed.run --nproc-per-node 1 ... --gpus 1 --global-batch 512` (strong-scaling path, RCCL group of 1)"),
    ("bench_scale1", "`--scale 1.0` (single pass)"),
    ("bench_ddim100_full", "`--respacing ddim100` (configs[2] as worded: 100 DDIM steps)"),
    ("bench_config1_shape", "`--batch 4 --diffusion-steps 50` (configs[0] shape)"),
    ("bench_beat256", "`--dataset beat --batch 256` (configs[4] at 34 frames, the whole job on one GPU)"),
    ("bench_beat150_b32", "`--dataset beat150 --batch 32` (configs[4] as worded, per-GPU share of 256/8; synthetic)"),
    ("bench_beat150_b256", "`--dataset beat150 --batch 256 --diffusion-steps 200` (synthetic)"),
    ("bench_threads_two_handles", "`--gpus 2 --launcher threads --ranks-share-device --batch 256` (ONE process, two engine handles on one GPU driven by two "
                                  "threads, no process group: the collective-free launcher, not a scaling number)"),
    ("bench_two_ranks_one_gpu", "`--gpus 2 --ranks-share-device --batch 256 --legs lively` (self-launched; TWO RANKS ON ONE GPU, collectives over gloo: "
                                "the N > 1 control flow, not a scaling number)"),
    ("bench_eight_ranks_one_gpu", "`--gpus 8 --ranks-share-device --batch 64 --legs none` (self-launched; EIGHT RANKS ON ONE GPU over gloo, RCCL probed and refused: "
                                  "the driver's N = 8 control flow on the one GPU there is, not a scaling number)"),
    ("bench_threads_eight_handles", "`--gpus 8 --launcher threads --ranks-share-device --batch 64 --path pass` (one process, eight handles on one GPU, eight threads)"),
]


def load(name):
    try:
        return json.loads(open(os.path.join(src, name + ".json")).read().strip().splitlines()[-1])
    except Exception:
        return None


def cat(path):
    p = os.path.join(src, path)
    return open(p).read().rstrip() if os.path.exists(p) else f"(missing: {path})"


out = [f"# {title}: bench lines, kernel traces and PMC passes on MI355X", ""]
tail = cat("pytest_gpu.log").splitlines()[-1] if os.path.exists(os.path.join(src, "pytest_gpu.log")) else "?"
out += [f"All from ONE `gpurun` call (`tools/round_measure.sh {os.path.basename(src)}`, same box, same build): `pytest -m gpu` ({tail}), the bench lines below, a",
        "rocprofv3 kernel trace of the headline command, and kernel-trace + PMC passes (one counter per run, `--pmc X --kernel-trace` only) of the",
        "LivelySpeaker example and the synthetic 150-frame variant.  Assembled by `tools/round_report.py`.", "",
        "## bench.py lines (value = pose-frames/s; kernel_ms = mean step time from HIP events on the engine's stream)", "",
        "| run | command (after `python bench.py`) | value | ms per call | kernel_ms | roofline frac | traffic B/launch | parity_in_run max\\|d\\| |", "|---|---|---|---|---|---|---|---|"]
for name, cmd in RUNS:
    r = load(name)
    if not r:
        out.append(f"| {name} | {cmd} | (missing) | | | | | |")
        continue
    rf = r["roofline"]
    out.append(f"| {name} | {cmd} | {r['value']:.1f} | {r['ms_per_step']:.2f} | {rf.get('kernel_ms', rf.get('kernel_ms_device0'))} | {rf.get('frac')} | {rf.get('traffic')} | "
               f"{(r.get('parity_in_run') or {}).get('max_abs_diff')} |")
d = load("bench_default")
if d:
    out += ["", "Secondary objects of the default line:", ""]
    for k in ("single_pass", "identical_seeds_mode", "config1_shape", "livelyspeaker", "configs4_beat", "mid_batches", "split_precision", "train_step", "cpu_baseline",
              "shard_check", "parity_in_run"):
        if k in d:
            out.append(f"* `{k}`: `{json.dumps(d[k])}`")
    out += ["", "The default line in full:", "", "```", json.dumps(d), "```"]
out += ["", "## Kernel trace of the headline command (`rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 ...`)", "", cat("kt_bench.md"), "",
        "### per (kernel, grid)", "", cat("kt_dispatch.md"), "",
        "## LivelySpeaker example (`examples/livelyspeaker_ted.py 512`): per (kernel, grid) dispatch summary", "", cat("lively/kt.md")]
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"):
    out += ["", f"### PMC {c}", "", cat(f"lively/pmc_{c}.md")]
out += ["", "## Synthetic 150-frame variant, B=32 (`bench.py --dataset beat150 --batch 32 ... --diffusion-steps 20`): dispatch summary", "", cat("long/kt.md"), ""]
two = load("bench_two_ranks_one_gpu")
if two:
    out += ["", "## Two ranks on one GPU (`python bench.py --gpus 2 --ranks-share-device ...`, launched by bench.py itself): shard check and the LivelySpeaker leg", "",
            f"* `shard_check`: `{json.dumps(two.get('shard_check'))}`", f"* `livelyspeaker`: `{json.dumps(two.get('livelyspeaker'))}`"]
out += ["", "## Synthetic 150-frame variant, B=256: dispatch summary", "", cat("long256/kt.md"),
        "", "## Training step (`tools/train_perf.py ted 512 4`): dispatch summary", "", cat("train/kt.md"),
        "", "## Once-per-call stage (`tools/prepare_only.py`, B=512): dispatch summary and PMC passes", "", cat("prepare/kt.md")]
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
    out += ["", f"### PMC {c}", "", cat(f"prepare/pmc_{c}.md")]
out += ["", "## ms per diffusion step over the batch size, by kernel family and for `auto` (`tools/coop_time.py`)", "", "```", cat("tvb_ted.txt"), "",
        cat("tvb_beat.txt"), "```",
        "", "## Sample-split step kernel, BEAT B = 32 (`tools/coop_time.py beat 20 32 coop`): dispatch summary and PMC passes", "", cat("coop/kt.md")]
for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
    out += ["", f"### PMC {c}", "", "\n".join(l for l in cat(f"coop/pmc_{c}.md").splitlines() if "k_coop" in l or l.startswith("| kernel") or l.startswith("|---"))]
# round 5: the one-pass-per-workgroup kernel, one workgroup per CU (B = 128) and two (B = 256), fp32 and bf16x3, the fused kernel beside it
def pmc_rows(d, pick):
    rows = []
    if not os.path.isdir(os.path.join(src, d)):
        return rows
    for f in sorted(os.listdir(os.path.join(src, d))):
        if f.startswith("pmc_") and f.endswith(".md"):
            for l in cat(f"{d}/{f}").splitlines():
                if pick in l:
                    rows.append(l)
    return rows
HDR = ["| kernel | workgroups (x,y,z) x threads | launches | avg us | min us | vgpr | lds B | counters (mean per launch) |", "|---|---|---|---|---|---|---|---|"]
sec = []
for d, what, pick in (("pass128", "`tools/coop_time.py ted 20 128 pass`: ONE workgroup per CU", "k_pass<35"), ("pass256", "`tools/coop_time.py ted 20 256 pass`: TWO workgroups per CU", "k_pass<35"),
                      ("fused256", "`tools/coop_time.py ted 20 256 fused`: the fused kernel, same 256 clips", "k_step<35"),
                      ("pass256_bf16x3", "`tools/bf16x3_time.py pass 256 256`: bf16x3, two workgroups per CU", "k_pass<"),
                      ("fused256_bf16x3", "`tools/bf16x3_time.py fused 256 256`: bf16x3, the fused kernel", "k_step<")):
    rows = pmc_rows(d, pick)
    if rows:
        sec += ["", f"### {what}", ""] + HDR + rows
if sec:
    out += ["", "## One-pass-per-workgroup step kernel (ls_pass_kernel.h): one PMC pass per counter, mean per launch summed over the chip"] + sec + [
            "", "Reading: GRBM_GUI_ACTIVE / 8 / avg us = the shader clock the launch ran at (profiled passes run ~4 % slower than un-profiled ones);",
            "SQ_VALU_MFMA_BUSY_CYCLES / 1024 / (GRBM_GUI_ACTIVE / 8) = matrix pipe busy per SIMD; (SQ_INSTS_VALU - SQ_INSTS_MFMA) / 1024 x 4 = issue cycles of the other vector instructions.",
            "", "Workgroup placement and start / end offsets of CU neighbours (`tools/wg_place.py`, -DLS_DEBUG build), fp32 then bf16x3:", "", "```", cat("wg_place_pass_ted256.txt"), "",
            cat("wg_place_pass_ted256_bf16x3.txt"), "```",
            "", "## bf16x3 (opt-in) step time by kernel family and batch (`tools/bf16x3_time.py auto,fused,pass ...`)", "", "```", cat("bf16x3_time.txt"), "```"]
KSTEP = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
         "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE")
rows = []
for c in KSTEP:
    for l in cat(f"kstep/pmc_{c}.md").splitlines():
        if "k_step" in l:
            rows.append(l)
            break
if rows:
    out += ["", "## Fused step kernel, TED B = 512 (`tools/coop_time.py ted 20 512 fused`): one PMC pass per counter, mean per launch summed over the chip",
            "", "| kernel | workgroups (x,y,z) x threads | launches | avg us | min us | vgpr | lds B | counters (mean per launch) |", "|---|---|---|---|---|---|---|---|"] + rows + [
            "", "Reading (1024 SIMDs, 8 XCDs): GRBM_GUI_ACTIVE / 8 = cycles of the launch; SQ_VALU_MFMA_BUSY_CYCLES / 1024 = matrix-pipe cycles per SIMD",
            "(= SQ_INSTS_MFMA / 1024 x 32: every `v_mfma_f32_16x16x4_f32` holds the pipe 32 cycles); (SQ_INSTS_VALU - SQ_INSTS_MFMA) / 1024 x 4 = issue",
            "cycles of the other vector instructions per SIMD; SQ_VALU_MFMA_COEXEC_CYCLES = 0: the two never overlap on this kernel."]
out += [
        "", "## Stride-6 conv layers stand-alone (`tools/conv_bench.cpp`) and the per-stage barrier timeline of one workgroup (`-DLS_CONV_PROF`)", "", "```",
        cat("conv_bench.txt"), "", cat("conv_bench_prof.txt"), "```",
        "", "## Backward conv kernels of the training step stand-alone (`tools/conv_bwd_bench.cpp`: timing after a clock warm-up, sampled entries against a host evaluation)",
        "", "```", cat("conv_bwd_bench.txt"), "```", "", "SAG decode: " + cat("sag_time.txt").splitlines()[-1],
        "", "Training step (`tools/train_perf.py ted 512 8`, last steps): ", "", "```", cat("train_perf.txt"), "```", ""]
open(dst, "w").write("\n".join(out))
print("wrote", dst, len(out), "lines")
