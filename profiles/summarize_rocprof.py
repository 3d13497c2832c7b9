#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (``rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd``) into the
markdown per-kernel summary that is committed under profiles/ (gpurun_out/ is scratch)."""
import sqlite3
import sys


def main(db_path, title, bench_line=""):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    print(f"# {title}\n")
    if bench_line:
        print("bench.py line of the profiled run:\n\n```json\n" + bench_line.strip() + "\n```\n")
    print("| kernel | calls | total (us) | avg (us) | % |\n|---|---|---|---|---|")
    for name, calls, total, avg, pct in rows[:14]:
        short = name.split("(")[0].replace("void ", "")
        if len(short) > 70:
            short = short[:67] + "..."
        print(f"| `{short}` | {calls} | {total:.1f} | {avg:.2f} | {pct:.3f} |")
    r = cur.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x, "
                    "min(duration), max(duration) from kernels where name like '%k_step%' group by name").fetchall()
    for x in r:
        print(f"\n`{x[0].split('(')[0]}`: vgpr={x[1]} agpr={x[2]} sgpr={x[3]} lds={x[4]} B scratch={x[5]} B/lane "
              f"grid={x[6]} wg={x[7]} duration min/max = {x[8] / 1e3:.1f}/{x[9] / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], open(sys.argv[3]).read().strip().splitlines()[-1] if len(sys.argv) > 3 else "")
