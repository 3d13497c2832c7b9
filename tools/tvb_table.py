#!/usr/bin/env python3
"""profiles/rNN_throughput_vs_batch.md from the tvb_*.txt files of tools/round_measure.sh (tools/coop_time.py output):
    python tools/tvb_table.py gpurun_out/r05p profiles/r05_throughput_vs_batch.md "Round 5" """
import os
import re
import sys

src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
FAM = {"coop8": "sample-split x8", "coop4": "sample-split x4", "coop2": "sample-split x2", "coop": "sample-split", "batch": "batch-level", "pass": "one-pass-per-workgroup", "fused": "fused"}
NAMES = {0: "fused", 1: "batch-level", 2: "sample-split", 3: "one-pass-per-workgroup"}
out = [f"# {title}: step time vs batch, by kernel family (MI355X, 30-step DDPM, CFG 1.5, Philox noise, hipGraph replay)", "",
       "`python tools/coop_time.py <ds> 30 <batches> coop8,coop4,coop2,batch,pass,fused,auto` (sample-split with 8 / 4 / 2 slice workgroups per (clip, CFG pass)) in the round's final measurement set (one box, one run; ms per step | pose-frames/s",
       "extrapolated to a 1000-step call = B * 34 / ms).  `auto` = the plan `plan_steps` (ls_api.cpp) makes from its step-time model; the last column",
       "says which pieces it ran: family, then `+ n clips on family` for the second / third piece.  one-pass-per-workgroup = `k_pass`: 8-wave",
       "workgroups, one per CU, while the grid fits the chip once (B <= 128), 4-wave workgroups, two per CU, beyond.", ""]
for ds, fname in (("TED (S = 35, J*F = 27)", "tvb_ted.txt"), ("BEAT (S = 36, J*F = 282)", "tvb_beat.txt")):
    path = os.path.join(src, fname)
    if not os.path.exists(path):
        continue
    rows = {}
    for line in open(path):
        m = re.match(r"(\w+)\s+B=\s*(\d+)\s+([\d.]+) ms/step\s+(\d+) frames/s\s+path=(\d)(.*)", line)
        if m:
            rows.setdefault(int(m.group(2)), {})[m.group(1)] = (float(m.group(3)), int(m.group(4)), int(m.group(5)), m.group(6).strip())
    fams = [f for f in FAM if any(f in r for r in rows.values())]
    out += [f"## {ds}", "", "| B | " + " | ".join(f"{FAM[f]} ms | frames/s" for f in fams) + " | best family | `auto` ms | frames/s | what `auto` ran |",
            "|---|" + "---|" * (2 * len(fams) + 4)]
    for B in sorted(rows):
        r = rows[B]
        cells = []
        for f in fams:
            cells += [f"{r[f][0]:.4f}", str(r[f][1])] if f in r else ["", ""]
        best = min((f for f in fams if f in r), key=lambda f: r[f][0], default=None)
        a = r.get("auto")
        what = ""
        if a:
            what = NAMES[a[2]]
            for n, p in re.findall(r"\+(\d+)@(\d)", a[3]):
                what += f" + {n} on {NAMES[int(p)]}"
        out.append(f"| {B} | " + " | ".join(cells) + f" | {FAM.get(best, '')} | " + (f"{a[0]:.4f} | {a[1]} | {what} |" if a else " | | |"))
    out.append("")
open(dst, "w").write("\n".join(out))
print("wrote", dst)
