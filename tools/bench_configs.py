#!/usr/bin/env python3
"""Secondary timings (not the bench.py contract line): the other BASELINE.json configurations,
each checked bit-exact against the CPU oracle.  usage: python tools/bench_configs.py [--batch N]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mozjpeg_amd as M  # noqa: E402
import oracle_lib as O  # noqa: E402


def run(name, w, h, kw, batch, steps=5):
    import torch
    gen = O.synthetic_frame12 if kw.get("precision") == 12 else O.synthetic_frame
    frames = np.stack([gen(w, h, 1234 + i) for i in range(batch)])
    t = torch.from_numpy(frames.view(np.int16) if frames.dtype == np.uint16 else frames).cuda()
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=batch)
    enc.encode_tensor(t, stream="own"); enc.sync()
    ok = enc.get_jpeg(0) == O.encode(O.make_params(w, h, **kw), frames[0])
    enc.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        enc.encode_tensor(t, stream="own")
    enc.sync()
    dt = (time.perf_counter() - t0) / steps
    kt = enc.kernel_times()
    print(json.dumps({"config": name, "size": "%dx%d" % (w, h), "switches": str(kw), "batch": batch,
                      "ms_per_batch": round(dt * 1e3, 3), "mpix_per_s": round(w * h * batch / dt / 1e6, 1),
                      "bit_exact": ok, "jpeg_bytes": enc.jpeg_size(0),
                      "kernel_ms": {k: round(v, 3) for k, v in kt}}), flush=True)
    enc.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    run("C2 1080p q75 4:2:0 baseline trellis", 1920, 1080, dict(baseline=True), a.batch)
    run("C3 4K q85 4:2:0 progressive + scan search", 3840, 2160, dict(quality=85), a.batch)
    run("4K q75 fastcrush (progressive, no search)", 3840, 2160, dict(fastcrush=True), a.batch)
    run("4K q75 -revert (libjpeg-turbo defaults)", 3840, 2160, dict(revert=True), a.batch)
    run("C4-like batch: 32 x 1080p q75 baseline trellis", 1920, 1080, dict(baseline=True), 32)
    run("C5 8-bit twin: 8192x8192 q90 4:4:4 baseline trellis, restart every MCU row", 8192, 8192,
        dict(baseline=True, quality=90, sample=(1, 1), restart=1), 1, steps=3)
    run("C5: 8192x8192 12-bit q90 4:4:4 baseline, restart every MCU row, -notrellis (12-bit trellis does not exist, F1)", 8192, 8192,
        dict(precision=12, baseline=True, notrellis=True, quality=90, sample=(1, 1), restart=1), 1, steps=3)
