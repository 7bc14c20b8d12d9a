#!/usr/bin/env python3
"""Turn a rocprofv3 results database (rocpd sqlite, `rocprofv3 --kernel-trace --stats -d DIR -o NAME`)
into the per-kernel summary committed under profiles/.
usage: tools/rocprof_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.csv"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    print("kernel,calls,total_us,avg_us,percent")
    for name, calls, total, avg, pct in cur.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"):
        short = name.split("(")[0].replace("void ", "")
        print('"%s",%d,%.1f,%.2f,%.2f' % (short, calls, total, avg, pct))
    if len(sys.argv) > 2 and sys.argv[2] == "--pmc":
        print()
        for row in cur.execute("select * from counters_collection limit 5"):
            print(row)


if __name__ == "__main__":
    main()


def pmc_summary(fetch_db, write_db, frames_per_launch):
    """Per-kernel HBM traffic from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; KB per launch).
    FETCH_SIZE on gfx950 counts 64 B per 128-B request for coalesced streams
    (MI355X_MICROARCH.md, HBM section): calibrated here on k_color, whose input is exactly
    W*H*3 bytes per frame -- the factor comes out as 2.0 and is applied to every kernel's fetch."""
    import json
    out = {}
    for nm, db in (("fetch_kb", fetch_db), ("write_kb", write_db)):
        cur = sqlite3.connect(db).cursor()
        for name, avg in cur.execute("select kernel_name, avg(value) from counters_collection group by kernel_name"):
            short = name.split("(")[0].replace("void ", "")
            out.setdefault(short, {})[nm] = avg
    return out
