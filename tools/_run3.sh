cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -q -m gpu --timeout 120 -x -k "dc_ver or dc_scan or passthrough or unsupported or cjpeg" > gpurun_out/t3.log 2>&1; tail -12 gpurun_out/t3.log
O=gpurun_out/calib; mkdir -p $O
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o cf -- tools/pmc_calib/pmc_calib > $O/cf.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o cw -- tools/pmc_calib/pmc_calib > $O/cw.log 2>&1
timeout 120 rocprofv3 --kernel-trace --stats -d $O -o cs -- tools/pmc_calib/pmc_calib > $O/cs.log 2>&1
python - <<'PY'
import sqlite3, glob
for nm, ctr in (("cf","FETCH_SIZE"),("cw","WRITE_SIZE")):
    db = glob.glob("gpurun_out/calib/**/%s_results.db" % nm, recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    for name, avg, n in cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? group by kernel_name", (ctr,)):
        print(ctr, name.split("(")[0], "KB avg", avg, "n", n)
db = glob.glob("gpurun_out/calib/**/cs_results.db", recursive=True)[0]
for row in sqlite3.connect(db).cursor().execute("select name,total_calls,average from top_kernels"):
    print(row)
PY
