cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/tp -o tp -- python tools/train_perf.py ted 512 3 > /dev/null 2>&1
python - <<'PY'
import sqlite3
cur = sqlite3.connect("gpurun_out/tp/tp_results.db").cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
rows = list(cur.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), avg(duration), sum(duration) from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc"))
tot = sum(r[7] for r in rows)
for r in rows[:32]:
    print(f"{r[0].split('(')[0][:50]:50s} grid=({r[1]//r[4]},{r[2]},{r[3]}) n={r[5]:3d} avg={r[6]/1e3:8.1f}us  per-step={r[7]/1e3/4:8.1f}us")
print("total per step us", tot/1e3/4)
PY
