#!/usr/bin/env python3
"""Per-kernel HBM traffic from two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), as
MI355X_MICROARCH.md's HBM section prescribes (own runs, --kernel-trace only).

usage: tools/pmc_traffic.py FETCH_results.db WRITE_results.db FRAMES_PER_LAUNCH IN_BYTES_PER_FRAME > profiles/rNN_pmc_hbm_traffic.json

Both counters are reported in KB.  On gfx950 FETCH_SIZE under-reports coalesced streams by 2x
(64 B counted per 128-B request): the factor is not assumed but calibrated on k_color, whose only
input is exactly IN_BYTES_PER_FRAME bytes per frame, and applied to every kernel's fetch."""
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(db_path, counter):
    cur = sqlite3.connect(db_path).cursor()
    out = {}
    q = ("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? "
         "group by kernel_name")
    for name, avg, n in cur.execute(q, (counter,)):
        short = name.split("(")[0].replace("void ", "")
        out[short] = (avg, n)
    return out


def main():
    fetch_db, write_db, frames, in_bytes = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    fetch = per_kernel(fetch_db, "FETCH_SIZE")
    write = per_kernel(write_db, "WRITE_SIZE")
    color = [k for k in fetch if k.startswith("k_color")]
    factor = 1.0
    if color:
        factor = (in_bytes * frames) / (fetch[color[0]][0] * 1024.0)
    kernels = {}
    # a kernel may run more than once per encode call (round 6: the AC trellis over two image ranges): bytes are per CALL = the
    # average launch x the launches per call, counted against the colour kernel's one launch per call
    calls_f = fetch[color[0]][1] if color else 0
    calls_w = write[color[0]][1] if color and color[0] in write else 0
    for k in sorted(set(fetch) | set(write)):
        lf = fetch.get(k, (0.0, 0))[1] / calls_f if calls_f else 1.0
        lw = write.get(k, (0.0, 0))[1] / calls_w if calls_w else 1.0
        if 0.9 < lf < 1.1: lf = 1.0
        if 0.9 < lw < 1.1: lw = 1.0
        f = fetch.get(k, (0.0, 0))[0] * 1024.0 * factor * max(lf, 1.0)
        w = write.get(k, (0.0, 0))[0] * 1024.0 * max(lw, 1.0)
        kernels[k] = {"fetch_bytes_corrected": int(f), "write_bytes": int(w), "hbm_bytes": int(f + w),
                      "hbm_bytes_per_frame": int((f + w) / frames), "launches_sampled": fetch.get(k, (0, 0))[1],
                      "launches_per_call": round(max(lf, 1.0), 2)}
    kernels = dict(sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes"]))
    # which tree the passes belong to: bench.py quotes a summary only while the kernel sources still hash to this stamp
    sys.path.insert(0, ROOT)
    import bench
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        if subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "mozjpeg_amd/csrc"], stderr=subprocess.DEVNULL).decode().strip():
            head += "+uncommitted"
    except Exception:
        head = "unknown"
    # ... or, kernel by kernel, while the machine code is still the one that ran here (tools/kernel_isa.py; build() writes the file)
    now = bench.kernel_fingerprints()
    print(json.dumps({
        "profile_head": head, "kernel_source_stamp": bench.kernel_source_stamp(),
        "kernel_isa": {k: now[k] for k in kernels if k in now},
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate --kernel-trace passes; bytes per encode call "
                "(%d frames; a kernel launched more than once per call: all its launches); fetch scaled by the factor calibrated on k_color's known input bytes" % frames,
        "frames_per_launch": frames, "fetch_calibration_factor": round(factor, 4), "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main()
