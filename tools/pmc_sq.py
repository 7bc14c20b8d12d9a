#!/usr/bin/env python3
"""Per-kernel averages of every counter found in rocprofv3 --pmc result databases (separate passes).
usage: tools/pmc_sq.py pass1_results.db [pass2_results.db ...] [--kernels substr,substr] > json"""
import json
import sqlite3
import sys


def main():
    dbs = [a for a in sys.argv[1:] if not a.startswith("--")]
    filt = None
    for a in sys.argv[1:]:
        if a.startswith("--kernels="):
            filt = a.split("=", 1)[1].split(",")
    out = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"
        for name, cname, avg, n in cur.execute(q):
            short = name.split("(")[0].replace("void ", "")
            if filt and not any(f in short for f in filt):
                continue
            out.setdefault(short, {})[cname] = round(avg, 1)
            out[short]["launches"] = n
    # kernels launched more than once per encode call (round 6: the AC trellis over two image ranges): the averages are per launch,
    # launches_per_step says how many of them make one call (counted against the colour kernel's one launch per call)
    calls = [v["launches"] for k, v in out.items() if k.startswith("k_color")]
    if calls:
        for k, v in out.items():
            r = v["launches"] / float(calls[0])
            v["launches_per_step"] = int(round(r)) if r > 1.4 else 1
    # which machine code the passes were taken on (bench.py quotes the counters only for kernels that still compile to it)
    try:
        import os
        import subprocess
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        isa = json.load(open(os.path.join(root, "mozjpeg_amd", "kernel_isa.json")))
        frames = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--frames=")]
        out["_meta"] = {"kernel_isa": {k: v["sha"] for k, v in isa["kernels"].items() if k in out},
                        "profile_head": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=root, stdout=subprocess.PIPE).stdout.decode().strip(),
                        "frames_per_launch": int(frames[0]) if frames else None}
    except Exception:
        pass
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
