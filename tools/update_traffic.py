#!/usr/bin/env python
"""Refresh profiles/k_step_traffic.json from the per-(kernel, grid) PMC summaries tools/prof_call.sh writes.

    python tools/update_traffic.py KEY KERNEL_SUBSTR GRID_X DIR [SOURCE_NOTE]

KEY is bench.py's lookup key ("ted:512", "ted:512:single_pass", "beat:256"); KERNEL_SUBSTR selects the k_step instantiation
("k_step<35, 1, 27, 0, 0, 0>"); GRID_X is the launch's workgroup count; DIR holds pmc_FETCH_SIZE.md and pmc_WRITE_SIZE.md.
The entry is stamped with the SHA-256 of ls_step_kernel.h as it is NOW, which is what bench.py checks before quoting it.
"""
import hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "profiles", "k_step_traffic.json")
KSRC = os.path.join(ROOT, "livelyspeaker_amd", "csrc", "ls_step_kernel.h")


def pick(md, kernel, gx, counter):
    for line in open(md):
        cells = [c.strip() for c in line.split("|")]
        if len(cells) < 9 or kernel not in cells[1]:
            continue
        m = re.match(r"\((\d+),", cells[2])
        if not m or int(m.group(1)) != gx:
            continue
        v = re.search(counter + r"=([0-9.e+]+)", cells[8])
        if v:
            return cells[1].strip("`"), int(cells[3]), float(v.group(1))
    raise SystemExit(f"{md}: no row for {kernel} grid {gx}")


def main():
    key, kernel, gx, d = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    note = sys.argv[5] if len(sys.argv) > 5 else d
    name, n, fetch = pick(os.path.join(d, "pmc_FETCH_SIZE.md"), kernel, gx, "FETCH_SIZE")
    _, _, write = pick(os.path.join(d, "pmc_WRITE_SIZE.md"), kernel, gx, "WRITE_SIZE")
    doc = json.load(open(PATH))
    doc["entries"][key] = {
        "bytes_per_launch": int(round((2.0 * fetch + write) * 1024)),
        "fetch_size_kib": fetch, "write_size_kib": write, "kernel": name, "launches_averaged": n,
        "kernel_source_sha256": hashlib.sha256(open(KSRC, "rb").read()).hexdigest(), "source": note}
    json.dump(doc, open(PATH, "w"), indent=1)
    print(key, doc["entries"][key])


if __name__ == "__main__":
    main()
