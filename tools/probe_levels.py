"""Per-interval times of successive profiled steps of one configuration (why does the first bracketed pass differ?).
usage: python tools/probe_levels.py c3"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mozjpeg_amd as M
import bench as B
def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
    c = B.CONFIGS[cfg]
    print(c)
    w, h, kw, batch = c["w"], c["h"], c["kw"], c.get("batch", 32)
    frames = B.make_frames(w, h, [1234 + i for i in range(batch)], kw.get("precision") == 12, 1)
    t = torch.from_numpy(frames).cuda()
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=batch)
    def step():
        enc.encode_tensor(t, stream="own"); enc.sync()
    for i in range(6): step()  # (each step syncs: the encoder adapts its trellis tier from the previous batch)
    for rnd in range(4):
        enc.set_profiling(1)
        for i in range(3):
            step()
            kt = dict(enc.kernel_times())
            top = sorted(kt.items(), key=lambda kv: -kv[1])[:4]
            print(rnd, i, [(k, round(v, 3)) for k, v in top])
        enc.set_profiling(0)
        for i in range(5): step()


if __name__ == "__main__":
    main()
