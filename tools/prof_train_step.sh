#!/bin/bash
# Per-kernel breakdown of the training step on the GPU box (rocprofv3 kernel trace of tools/train_perf.py, 4 steps).
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/tp -o tp -- python tools/train_perf.py ${1:-ted} ${2:-512} 3 > /dev/null 2>&1
python - <<'PY'
import sqlite3
cur = sqlite3.connect("gpurun_out/tp/tp_results.db").cursor()
rows = list(cur.execute("select name, grid_x/workgroup_x, grid_y, grid_z, count(*), avg(duration), sum(duration) from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc"))
tot = sum(r[6] for r in rows) / 4
for r in rows[:34]:
    print(f"{r[0].split('(')[0].replace('void ','')[:46]:46s} ({r[1]},{r[2]},{r[3]}) n/step={r[4]/4:5.2f} avg={r[5]/1e3:8.1f}us per-step={r[6]/1e3/4:8.1f}us {r[6]/4/tot*100:5.1f}%")
print(f"kernel time per step: {tot/1e3:.1f} us in {sum(r[4] for r in rows)/4:g} launches")
PY
