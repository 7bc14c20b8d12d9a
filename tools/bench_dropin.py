#!/usr/bin/env python3
"""Drop-in throughput (never bench.py's `value`): what an UNCHANGED libjpeg client gets.
  1. tests/native/mt_bench (public libjpeg API, T threads x N images, jpeg_mem_dest) against
       - the reference's libjpeg.so.62 (CPU, oracle/_ref)           -> "reference"
       - the same library with libmozjpeg_hip_jpeg62.so preloaded    -> "preload"
       - mozjpeg_amd/standalone/libjpeg.so.62 instead of it          -> "standalone"
  2. the unchanged cjpeg binary on one 4K PPM: wall time of the whole process (start-up, device context, encode, file).
usage: python tools/bench_dropin.py [--threads 1,4,16] [--images 8] > profiles/rNN_dropin.json"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.path.join(ROOT, "oracle", "_ref")
SHIM = os.path.join(ROOT, "mozjpeg_amd", "libmozjpeg_hip_jpeg62.so")
STANDALONE = os.path.join(ROOT, "mozjpeg_amd", "standalone")
MT = os.path.join(ROOT, "tests", "native", "mt_bench")


def env_for(mode):
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    env["LD_LIBRARY_PATH"] = (STANDALONE if mode == "standalone" else REF) + ":" + env.get("LD_LIBRARY_PATH", "")
    if mode == "preload":
        env["LD_PRELOAD"] = SHIM
    return env


def mt(mode, threads, images, w, h, q, baseline=True, timeout=600):
    cmd = [MT, str(threads), str(images), str(w), str(h), str(q)] + (["baseline"] if baseline else [])
    try:
        r = subprocess.run(cmd, env=env_for(mode), capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"mode": mode, "threads": threads, "error": "timeout"}
    if r.returncode != 0:
        return {"mode": mode, "threads": threads, "error": r.stderr[-400:]}
    d = json.loads(r.stdout.strip().splitlines()[-1])
    d["mode"] = mode
    return d


def cjpeg_wall(mode, ppm, args, out):
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(REF, "cjpeg")] + args + ["-outfile", out, ppm], env=env_for(mode), capture_output=True, text=True)
    dt = time.perf_counter() - t0
    return {"mode": mode, "args": " ".join(args), "wall_s": round(dt, 3), "ok": r.returncode == 0,
            "bytes": os.path.getsize(out) if r.returncode == 0 else 0}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,4,16")
    ap.add_argument("--images", type=int, default=10)
    ap.add_argument("--seconds", type=float, default=3.0, help="target duration of one measured point")
    ap.add_argument("--ref-images", type=int, default=1)
    ap.add_argument("--tmp", default="/tmp")
    a = ap.parse_args()
    res = {"what": "unchanged libjpeg clients on this box; pixels in ordinary host memory, JPEG files in host memory",
           "library_client": [], "cjpeg": []}
    ths = [int(x) for x in a.threads.split(",")]
    for (w, h) in ((3840, 2160), (1920, 1080)):
        same = set()
        for mode in ("preload", "standalone"):
            for t in ths:
                # images PER THREAD, sized for about three seconds per point (0.07-s samples are too short to quote):
                # ~550 (4K) / ~1100 (1080p) images/s from one thread, ~2x from 4, ~3x from 16 in round 3
                per_thread = int(a.seconds * (550 if w > 2000 else 1100) * {1: 1.0, 4: 0.5}.get(t, 3.0 / max(t, 1)))
                d = mt(mode, t, max(a.images, per_thread), w, h, 75)
                res["library_client"].append(d); same.add(d.get("fnv1a_first"))
                print(json.dumps(d), file=sys.stderr, flush=True)
        d = mt("reference", max(ths), a.ref_images, w, h, 75)     # CPU: one image per thread on every requested thread
        res["library_client"].append(d); same.add(d.get("fnv1a_first"))
        print(json.dumps(d), file=sys.stderr, flush=True)
        res["files_identical_%dx%d" % (w, h)] = len(same) == 1
    import numpy as np
    import oracle_lib as O
    ppm = os.path.join(a.tmp, "dropin_4k.ppm")
    img = O.synthetic_frame(3840, 2160, 4242)
    with open(ppm, "wb") as f:
        f.write(b"P6\n3840 2160\n255\n" + np.ascontiguousarray(img).tobytes())
    outs = {}
    for args in (["-quality", "75", "-baseline"], ["-quality", "85"]):
        for mode in ("reference", "preload", "standalone"):
            out = os.path.join(a.tmp, "dropin_%s.jpg" % mode)
            d = cjpeg_wall(mode, ppm, args, out)
            outs[mode] = open(out, "rb").read() if d["ok"] else None
            res["cjpeg"].append(d)
            print(json.dumps(d), file=sys.stderr, flush=True)
        res["cjpeg_files_identical " + " ".join(args)] = outs["reference"] == outs["preload"] == outs["standalone"]
    print(json.dumps(res, indent=1))
