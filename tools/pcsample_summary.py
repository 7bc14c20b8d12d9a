#!/usr/bin/env python3
"""Summary of a rocprofv3 PC-sampling CSV (tools/gpu_pcsample.sh): samples per kernel, the hottest instructions of the three
heaviest kernels, mean active lanes per sample.  Column names differ between rocprofv3 versions: matched loosely."""
import collections
import csv
import sys


def col(header, *names):
    low = {h.lower(): h for h in header}
    for n in names:
        for k, h in low.items():
            if n in k:
                return h
    return None


def main():
    rows = csv.DictReader(open(sys.argv[1], newline=""))
    hdr = rows.fieldnames
    c_ins = col(hdr, "instruction")
    c_com = col(hdr, "comment")
    c_exec = col(hdr, "exec_mask", "exec")
    c_disp = col(hdr, "dispatch_id", "dispatch", "kernel")
    c_why = col(hdr, "stall", "reason", "wave_issued", "issued")
    print("columns:", hdr)
    per = collections.defaultdict(collections.Counter)
    lanes = collections.defaultdict(lambda: [0, 0])
    why = collections.defaultdict(collections.Counter)
    n = 0
    for r in rows:
        n += 1
        k = r.get(c_disp, "?")
        ins = (r.get(c_ins) or "?").strip() + ("   ; " + r[c_com].strip() if c_com and r.get(c_com) else "")
        per[k][ins] += 1
        if c_exec and r.get(c_exec):
            try:
                m = int(r[c_exec], 0)
                lanes[k][0] += bin(m).count("1")
                lanes[k][1] += 1
            except ValueError:
                pass
        if c_why and r.get(c_why) is not None:
            why[k][r[c_why]] += 1
    print("samples:", n)
    for k, cnt in sorted(per.items(), key=lambda kv: -sum(kv[1].values()))[:6]:
        tot = sum(cnt.values())
        la = lanes[k]
        print("\n== %s: %d samples (%.1f %%), mean active lanes %.1f" % (k, tot, 100.0 * tot / max(n, 1), la[0] / la[1] if la[1] else -1))
        if why[k]:
            print("   ", dict(why[k].most_common(8)))
        for ins, c in cnt.most_common(40):
            print("  %6d  %5.2f %%  %s" % (c, 100.0 * c / tot, ins[:150]))


if __name__ == "__main__":
    main()
