cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE; do
  timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o p -- python tools/train_perf.py ted 512 2 > /dev/null 2>&1
  python - $c <<'PY'
import sqlite3, sys, glob
c = sys.argv[1]
dbs = glob.glob(f"gpurun_out/pmc_{c}/*.db")
if not dbs: print(c, "no db"); sys.exit()
cur = sqlite3.connect(dbs[0]).cursor()
try:
    rows = list(cur.execute("""select k.name, k.grid_z, count(*), avg(pc.value), avg(k.duration) from counters_collection pc join kernels k on pc.dispatch_id = k.dispatch_id
       where k.name like '%k_gemm_tr%' and k.grid_z = 48 group by k.name"""))
    for r in rows: print(c, r[0][:40], "n", r[2], "avg", r[3], "dur_us", r[4]/1e3)
except Exception as e:
    print(c, "ERR", e, [r[0] for r in cur.execute("select name from sqlite_master")][:40])
PY
done
