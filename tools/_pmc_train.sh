cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_WAVES FETCH_SIZE WRITE_SIZE SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE; do
  timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o p -- python tools/train_perf.py ted 512 2 > /dev/null 2>&1
  python - $c <<'PY'
import sqlite3, sys, glob
c = sys.argv[1]
dbs = glob.glob(f"gpurun_out/pmc_{c}/*.db")
if not dbs: print(c, "no db"); sys.exit()
cur = sqlite3.connect(dbs[0]).cursor()
try:
    rows = list(cur.execute("""select k.name, k.grid_x/k.workgroup_x, k.grid_y, k.grid_z, count(*), avg(pc.value), avg(k.duration) from counters_collection pc join kernels k on pc.dispatch_id = k.dispatch_id
       where (k.name like '%k_conv%' or k.name like '%mixer_bwd%' or k.name like '%k_step%' or k.name like '%tokmix%') group by k.name, k.grid_x, k.grid_y, k.grid_z order by avg(k.duration) desc"""))
    for r in rows[:12]: print(f"{c:26s} {r[0].split('(')[0][-28:]:28s} ({r[1]},{r[2]},{r[3]}) n={r[4]} avg={r[5]:.4g} dur_us={r[6]/1e3:.1f}")
except Exception as e:
    print(c, "ERR", e)
PY
done
