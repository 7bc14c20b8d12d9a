"""Per-scan workgroup durations of the progressive path (introspection tap MJH_TAP_PROG_SCAN_US)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import mozjpeg_amd as M
import oracle_lib as O
w, h, B = 3840, 2160, int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, kw in (("q85 scan search", dict(quality=85)), ("q75 fastcrush", dict(fastcrush=True))):
    frames = np.stack([O.synthetic_frame(w, h, 1234 + i) for i in range(B)])
    d = torch.from_numpy(frames).cuda()
    p = M.make_params(w, h, **kw)
    enc = M.Encoder(p, max_batch=B)
    for _ in range(2): enc.encode_tensor(d)
    enc.sync()
    t0 = time.perf_counter()
    for _ in range(3): enc.encode_tensor(d)
    enc.sync()
    dt = (time.perf_counter() - t0) / 3
    us = enc.read_tap(M.TAP_PROG_SCAN_US, 0, 0)
    print("%s: %.2f ms per %d frames (%.0f Mpx/s)" % (name, dt * 1e3, B, w * h * B / dt / 1e6))
    for i in range(p.num_scans):
        s = p.scan_info[i]
        comps = [s.component_index[k] for k in range(s.comps_in_scan)]
        print("  scan %2d comps %s Ss %2d Se %2d Ah %d Al %d : stats %5d us  encode %5d us" % (i, comps, s.Ss, s.Se, s.Ah, s.Al, us[0][i], us[1][i]))
    print("  trellis stats passes:", us[0][64:67])
    enc.set_profiling(1)
    enc.encode_tensor(d); print({k: round(v, 3) for k, v in enc.kernel_times()})
    enc.close()
