#!/usr/bin/env python3
"""Per-(kernel, grid) dispatch summary of a rocprofv3 rocpd database, in launch order of first appearance:
    python profiles/dispatch_summary.py results.db [name-filter]
Separates launches that share a kernel name but not a shape (e.g. the GEMMs of the once-per-call stage), which the
per-name `top_kernels` view (profiles/summarize_rocprof.py) merges.  With PMC runs it also prints the mean counter values."""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    m = re.match(r"([^(]+)", name)
    return m.group(1).strip()


def main(db, flt=""):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, duration, start, vgpr_count, lds_size, dispatch_id from kernels order by start").fetchall()
    groups, order = {}, []
    for name, gx, gy, gz, wx, dur, start, vg, lds, did in rows:
        key = (short(name), gx // max(wx, 1), gy, gz, wx)
        if flt and flt not in key[0]:
            continue
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append((dur, vg, lds, did))
    pmc = {}
    try:
        for did, cname, val in cur.execute("select dispatch_id, counter_name, value from counters_collection"):
            pmc.setdefault(did, {}).setdefault(cname, 0.0)
            pmc[did][cname] += val
    except sqlite3.Error:
        pass
    print("| kernel | workgroups (x,y,z) x threads | launches | avg us | min us | vgpr | lds B | counters (mean per launch) |\n|---|---|---|---|---|---|---|---|")
    for key in order:
        g = groups[key]
        d = [x[0] for x in g]
        cn = {}
        for _, _, _, did in g:
            for c, v in pmc.get(did, {}).items():
                cn.setdefault(c, []).append(v)
        cs = ", ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(cn.items()))
        print(f"| `{key[0][:80]}` | ({key[1]},{key[2]},{key[3]}) x {key[4]} | {len(g)} | {sum(d) / len(d) / 1e3:.1f} | {min(d) / 1e3:.1f} | {g[0][1]} | {g[0][2]} | {cs} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
