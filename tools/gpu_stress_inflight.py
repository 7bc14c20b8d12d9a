#!/usr/bin/env python3
"""Many consecutive encode calls with two batches in flight (the library's default) at bench.py's frames per call: the files of
sampled calls and of the last two against the first call's.  usage (on the GPU box): python tools/gpu_stress_inflight.py"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, bench, mozjpeg_amd as M


def main():
    for cfgk, batch, iters in (("metric", 96, 600), ("c3", 64, 120), ("c2", 256, 600)):
        cfg = bench.CONFIGS[cfgk]; w, h, kw = cfg["w"], cfg["h"], cfg["kw"]
        frames = bench.make_frames(w, h, [77 + i for i in range(min(batch, 16))], False, 1)
        import numpy as np
        frames = np.concatenate([frames] * (batch // len(frames)))
        d = torch.from_numpy(frames).cuda()
        enc = M.Encoder(M.make_params(w, h, **kw), max_batch=batch)
        enc.encode_tensor(d, stream="own"); enc.sync()
        first = [enc.get_jpeg(i) for i in range(batch)]
        t0 = time.time(); bad = 0
        for it in range(iters):
            enc.encode_tensor(d, stream="own")
            if it % 97 == 96 or it >= iters - 2:
                enc.sync()
                got = [enc.get_jpeg(i) for i in range(batch)]
                bad += sum(1 for a, b in zip(got, first) if a != b)
        enc.sync()
        print(cfgk, batch, iters, "calls,", "%.1f s," % (time.time() - t0), "files differing from the first call's:", bad, flush=True)
        enc.close()


if __name__ == "__main__":
    main()
