cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/calib2; mkdir -p $O
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o cf -- tools/pmc_calib/pmc_calib > $O/cf.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o cw -- tools/pmc_calib/pmc_calib > $O/cw.log 2>&1
timeout 120 rocprofv3 --kernel-trace --stats -d $O -o cs -- tools/pmc_calib/pmc_calib > $O/cs.log 2>&1
cat $O/cs.log | tail -2
python - <<'PY'
import sqlite3, glob
for nm, ctr in (("cf","FETCH_SIZE"),("cw","WRITE_SIZE")):
    db = glob.glob("gpurun_out/calib2/**/%s_results.db" % nm, recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    for name, avg, n in cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? group by kernel_name", (ctr,)):
        print(ctr, name.split("(")[0], "KB avg", avg, "n", n)
db = glob.glob("gpurun_out/calib2/**/cs_results.db", recursive=True)[0]
for row in sqlite3.connect(db).cursor().execute("select name,total_calls,average from top_kernels"):
    print(row[0].split("(")[0], row[1], row[2])
PY
