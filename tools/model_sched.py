#!/usr/bin/env python3
"""Scheduling model of the AC trellis walk loop (no GPU): issue cost of the tile-sorted passes of a 4K frame under "A then B
every iteration" (today) and "one phase per iteration, B once `thr` lanes wait" (tools/model/trellis_sched.c).
usage: python tools/model_sched.py [w h q]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from model_trellis import ZZ, ehufsi  # noqa: E402


def main():
    w, h, q = (int(a) for a in (sys.argv[1:4] + ["3840", "2160", "75"][len(sys.argv) - 1:]))
    so = os.path.join(ROOT, "tools", "model", "libtrellis_sched.so")
    if not os.path.exists(so):
        os.system("gcc -O2 -shared -fPIC -o %s %s -lm" % (so, so.replace("libtrellis_sched.so", "trellis_sched.c")))
    lib = C.CDLL(so)
    lib.wave_cost.restype = C.c_double
    C.c_int.in_dll(lib, "g_tight").value = int(os.environ.get("TIGHT", "0"))
    img = O.synthetic_frame(w, h, 1234)
    p = O.make_params(w, h, quality=q, baseline=True, **({"sample": (1, 1)} if q >= 90 else {}))
    _, taps = O.encode(p, img, want_taps=True)
    gs, _, _ = O.geometry(p)
    zz = (C.c_int * 64)(*ZZ)
    QN = int(os.environ.get("QN", "16"))
    comps = []
    for ci, g in enumerate(gs):
        uq = np.ascontiguousarray(taps[("coef_uq", ci)][:g.hib, :g.wib].reshape(-1, 64))
        n = uq.shape[0]
        t = p.quant_tbl_no[ci]
        qt = np.array(list(p.qtbl[t]), np.uint16)
        si = ehufsi(taps["ac_bits"][p.ac_tbl_no[ci]], taps["ac_vals"][p.ac_tbl_no[ci]])
        nq = np.zeros(n, np.int32); qmax = np.zeros(n, np.int32)
        steps = np.zeros((n, 64), np.uint8); ncd4 = np.zeros((n, 64), np.uint8)
        lib.trellis_records(uq.ctypes.data_as(C.c_void_p), n, qt.ctypes.data_as(C.c_void_p), si.ctypes.data_as(C.c_void_p),
                            C.c_double(p.lambda_log_scale1), C.c_double(p.lambda_log_scale2), zz,
                            nq.ctypes.data_as(C.c_void_p), steps.ctypes.data_as(C.c_void_p), ncd4.ctypes.data_as(C.c_void_p), qmax.ctypes.data_as(C.c_void_p))
        nq_eff = np.where((nq > QN) | (qmax >= 16), 0, nq).astype(np.int32)      # deferred blocks do not walk here
        comps.append((nq, nq_eff, steps, ncd4))
        print("comp %d: %d blocks, mean nq %.2f, mean pair-steps per record %.2f, deferred %.3f" % (
            ci, n, nq.mean(), steps.sum() / max(1, (steps > 0).sum()), (nq_eff != nq).mean()))
    # instruction counts of the loop's parts, read off the gfx950 assembly of k_trellis_ac_v3<16, 4, true, false>
    cA2, cA4, cB, cTop, cLook = 58.0, 84.0, 98.0, 8.0, 6.0
    T = 256

    def run(policy, thr):
        tot, lanes = 0.0, C.c_double(0.0)
        for nq, nq_eff, steps, ncd4 in comps:
            n = len(nq)
            for t0 in range(0, n, T):
                blk = np.arange(t0, min(n, t0 + T))
                key = np.minimum(nq[blk], 63)
                order = blk[np.argsort(-key, kind="stable")]
                for p0 in range(0, len(order), 64):
                    idx = np.full(64, -1, np.int32)
                    part = order[p0:p0 + 64]
                    idx[:len(part)] = part
                    if nq_eff[part].max() == 0:
                        continue
                    tot += lib.wave_cost(idx.ctypes.data_as(C.c_void_p), nq_eff.ctypes.data_as(C.c_void_p), steps.ctypes.data_as(C.c_void_p),
                                         ncd4.ctypes.data_as(C.c_void_p), QN, policy, thr, C.c_double(cA2), C.c_double(cA4), C.c_double(cB),
                                         C.c_double(cTop), C.c_double(cLook), C.byref(lanes))
        return tot, lanes.value / max(tot, 1.0)
    base, bl = run(0, 0)
    print("today (A then B every iteration):            cost %.3e  lanes/instr %.1f" % (base, bl))
    for thr in (1, 4, 8, 12, 16, 24, 32, 48, 64):
        c, l = run(1, thr)
        print("one phase per iteration, B at >= %2d waiting: cost %.3e (%.3f of today)  lanes/instr %.1f" % (thr, c, c / base, l))
    for pct in (50, 100, 200, 400, 800):
        c, l = run(2, pct)
        print("one phase per iteration, B when waiting >= %3d %% of walking: cost %.3e (%.3f of today)  lanes/instr %.1f" % (pct, c, c / base, l))


if __name__ == "__main__":
    main()
