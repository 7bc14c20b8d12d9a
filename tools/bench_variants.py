#!/usr/bin/env python3
"""A/B timing of kernel variants on the metric workload (4K q75 baseline trellis), one process, same frames.
usage: python tools/bench_variants.py [--batch 64] [--variants 0,6,1] [--env NAME]   (NAME defaults to MJH_TRELLIS_VARIANT)
Every variant's files are compared with variant[0]'s (all variants must be bit-identical)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
import mozjpeg_amd as M  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--variants", default="0,6")
    ap.add_argument("--env", default="MJH_TRELLIS_VARIANT")
    ap.add_argument("--config", default="metric")
    ap.add_argument("--prof", type=int, default=0, help="profiling level inside the timed loop (0 off, 2 = the dominant interval bracketed)")
    ap.add_argument("--focus", default="trellis_ac", help="with --prof 2: the interval bracketed in every step of the timed loop")
    a = ap.parse_args()
    cfg = bench.CONFIGS[a.config]
    w, h, kw = cfg["w"], cfg["h"], cfg["kw"]
    frames = bench.make_frames(w, h, [1234 + i for i in range(a.batch)], kw.get("precision", 8) == 12, 1)
    d = torch.from_numpy(frames).cuda()
    base = None
    for v in a.variants.split(","):
        os.environ[a.env] = v
        enc = M.Encoder(M.make_params(w, h, **kw), max_batch=a.batch)
        enc.encode_tensor(d, stream="own"); enc.sync()
        files = [enc.get_jpeg(i) for i in range(a.batch)]
        if base is None:
            base = files
        if a.prof == 2:
            enc.set_profiling(2, focus=a.focus)
        else:
            enc.set_profiling(a.prof)
        for _ in range(4):
            enc.encode_tensor(d, stream="own")
        enc.sync()
        if a.prof == 2:
            enc.set_profiling(2, focus=a.focus)      # (start the averages over: the first bracketed step creates the events)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            enc.encode_tensor(d, stream="own")
        enc.sync()
        dt = (time.perf_counter() - t0) / a.steps
        live = dict(enc.kernel_times()).get(a.focus) if a.prof == 2 else None
        enc.set_profiling(1)
        for _ in range(3):
            enc.encode_tensor(d, stream="own")
        kt = dict(enc.kernel_times())
        print(json.dumps({"variant": "%s=%s" % (a.env, v), "ms_per_batch": round(dt * 1e3, 3), "mpix_per_s": round(w * h * a.batch / dt / 1e6, 1),
                          "identical_to_first": files == base, "focus_ms_live": None if live is None else round(live, 4),
                          "kernel_ms": {k: round(x, 3) for k, x in sorted(kt.items(), key=lambda kv: -kv[1])[:8]}}), flush=True)
        enc.close()
