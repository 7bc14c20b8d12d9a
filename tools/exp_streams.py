"""Experiment: S encoders of B/S frames each on their own streams vs one encoder of B frames."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mozjpeg_amd as M
import oracle_lib as O
w, h, B = 3840, 2160, 16
frames = np.stack([O.synthetic_frame(w, h, 1234 + i) for i in range(B)])
d = torch.from_numpy(frames).cuda()
params = M.make_params(w, h, quality=75, baseline=True)
for S in (1, 2, 4):
    encs = [M.Encoder(params, max_batch=B // S, device=0) for _ in range(S)]
    parts = [d[i * (B // S):(i + 1) * (B // S)] for i in range(S)]
    def step():
        for e, p in zip(encs, parts):
            e.encode_tensor(p)
    for _ in range(3): step()
    for e in encs: e.sync()
    t0 = time.perf_counter()
    for _ in range(20): step()
    for e in encs: e.sync()
    dt = (time.perf_counter() - t0) / 20
    print("S=%d  %.3f ms/step  %.1f Mpx/s" % (S, dt * 1e3, w * h * B / dt / 1e6), flush=True)
    del encs
