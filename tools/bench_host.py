#!/usr/bin/env python3
"""PCIe-inclusive rate of the metric workload (never bench.py's `value`): pixels start in
ordinary host memory, complete JPEG files end in host memory -- mjh_encode_host (pageable -> pinned staging -> H2D on
the copy stream -> pipeline) followed by mjh_get_jpeg for every frame.  usage: python tools/bench_host.py [--batch N]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401  (initialises the HIP runtime torch ships before ours)
import mozjpeg_amd as M  # noqa: E402
import oracle_lib as O  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    w, h, B = 3840, 2160, a.batch
    frames = np.stack([O.synthetic_frame(w, h, 1234 + i) for i in range(B)])
    enc = M.Encoder(M.make_params(w, h, quality=75, baseline=True), max_batch=B)
    out = enc.encode_host(frames)
    ok = out[0] == O.encode(O.make_params(w, h, quality=75, baseline=True), frames[0])
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = enc.encode_host(frames)          # returns the files as bytes objects (D2H included)
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"what": "host pixels -> host JPEG bytes, 4K q75 baseline trellis", "batch": B, "ms_per_batch": round(dt * 1e3, 2),
                      "mpix_per_s": round(w * h * B / dt / 1e6, 1), "input_GBps": round(w * h * 3 * B / dt / 1e9, 2),
                      "bit_exact": ok}))
