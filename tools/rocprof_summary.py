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
