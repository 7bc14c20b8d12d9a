#!/usr/bin/env python3
"""Throughput with 1, 2 and 3 batches in flight (one encoder per batch, each on its own streams, taken round-robin).
usage: python tools/bench_inflight.py [metric|c2|c3|...] [batch]   -- what bench.py reports as `pipelined` comes from the same loop"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench, mozjpeg_amd as M
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "metric"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["batch"]
w, h, kw = cfg["w"], cfg["h"], cfg["kw"]
frames = bench.make_frames(w, h, [1234 + i for i in range(B)], False, 1)
d = torch.from_numpy(frames).cuda()
for nenc in [int(v) for v in os.environ.get("INFLIGHT", "1,2,3").split(",")]:
    encs = [M.Encoder(M.make_params(w, h, **kw), max_batch=B) for _ in range(nenc)]
    for e in encs:
        e.encode_tensor(d, stream="own"); e.sync()
    ref = [encs[0].get_jpeg(i) for i in range(B)]
    same = all([e.get_jpeg(i) for i in range(B)] == ref for e in encs)
    steps = 30
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        encs[k % nenc].encode_tensor(d, stream="own")
    for e in encs: e.sync()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"encoders_in_flight": nenc, "ms_per_step": round(dt * 1e3, 3), "mpix_per_s": round(w * h * B / dt / 1e6, 1), "identical": same}), flush=True)
    for e in encs: e.close()
