#!/usr/bin/env python3
"""CPU work model of the AC trellis walk (no GPU): how many pair-steps each block of a synthetic 4K frame needs, and what
a wave costs (max over its lanes) under different block-to-lane assignments.  usage: python tools/model_trellis.py [w h q]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

ZZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
      35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def ehufsi(bits, vals):
    si = np.zeros(256, np.uint8)
    p = 0
    for L in range(1, 17):
        for _ in range(int(bits[L])):
            si[vals[p]] = L
            p += 1
    return si


def main():
    w, h, q = (int(a) for a in (sys.argv[1:4] + ["3840", "2160", "75"][len(sys.argv) - 1:]))
    lib = C.CDLL(os.path.join(ROOT, "tools", "model", "libtrellis_work.so"))
    img = O.synthetic_frame(w, h, 1234)
    p = O.make_params(w, h, quality=q, baseline=True)
    _, taps = O.encode(p, img, want_taps=True)
    gs, _, _ = O.geometry(p)
    zz = (C.c_int * 64)(*ZZ)
    allnq, allst, allev, alllam = [], [], [], []
    for ci, g in enumerate(gs):
        uq = np.ascontiguousarray(taps[("coef_uq", ci)][:g.hib, :g.wib].reshape(-1, 64))
        n = uq.shape[0]
        t = p.quant_tbl_no[ci]
        qt = np.array(list(p.qtbl[t]), np.uint16)
        si = ehufsi(taps["ac_bits"][p.ac_tbl_no[ci]], taps["ac_vals"][p.ac_tbl_no[ci]])
        nq, st, ev, s1 = (np.zeros(n, np.int32) for _ in range(4))
        lam = np.zeros(n, np.float64)
        lib.trellis_work(uq.ctypes.data_as(C.c_void_p), n, qt.ctypes.data_as(C.c_void_p), si.ctypes.data_as(C.c_void_p),
                         C.c_double(p.lambda_log_scale1), C.c_double(p.lambda_log_scale2), zz,
                         nq.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), ev.ctypes.data_as(C.c_void_p), s1.ctypes.data_as(C.c_void_p), lam.ctypes.data_as(C.c_void_p))
        alllam.append(lam)
        allnq.append(nq); allst.append(st); allev.append(ev)
        print("comp %d: %d blocks, mean nq %.2f, mean steps %.2f, mean evals %.2f, share nq>16 %.3f, >12 %.3f, >20 %.3f" % (
            ci, n, nq.mean(), st.mean(), ev.mean(), (nq > 16).mean(), (nq > 12).mean(), (nq > 20).mean()))
    QN = 16

    def pad(a, m):
        r = (-len(a)) % m
        return np.concatenate([a, np.zeros(r, a.dtype)])

    tot_blocks = sum(len(a) for a in allnq)
    for name, fn in [("baseline 64 consecutive", lambda nq, st: np.where(pad(nq, 64) > QN, 0, pad(st, 64)).reshape(-1, 64).max(1).sum())]:
        tot = sum(fn(nq, st) for nq, st in zip(allnq, allst))
        ideal = sum(np.where(nq > QN, 0, st).sum() for nq, st in zip(allnq, allst)) / 64.0
        print("%-40s wave-steps %.0f   ideal %.0f   efficiency %.3f" % (name, tot, ideal, ideal / tot))
    for T in (128, 256, 512, 1024):
        for key in ("nq", "steps"):
            tot = 0
            for nq, st in zip(allnq, allst):
                nqp, stp = pad(nq, T).reshape(-1, T), pad(st, T).reshape(-1, T)
                stp = np.where(nqp > QN, 0, stp)
                k = nqp if key == "nq" else stp
                order = np.argsort(-k, axis=1, kind="stable")
                sst = np.take_along_axis(stp, order, 1)
                tot += sst.reshape(sst.shape[0], T // 64, 64).max(2).sum()
            print("sorted by %-5s tile %4d: wave-steps %.0f  efficiency %.3f" % (key, T, tot, ideal / tot))
        # heavy+light pairing inside a tile (2 blocks per lane, T = 128 only meaningful; larger T: k blocks per lane snake order)
        tot = 0
        for nq, st in zip(allnq, allst):
            nqp, stp = pad(nq, T).reshape(-1, T), pad(st, T).reshape(-1, T)
            stp = np.where(nqp > QN, 0, stp)
            order = np.argsort(-nqp, axis=1, kind="stable")
            sst = np.take_along_axis(stp, order, 1).reshape(-1, T // 64, 64)
            sst[:, 1::2, :] = sst[:, 1::2, ::-1]        # snake: odd passes reversed
            tot += sst.sum(1).max(1).sum()
        print("snake-paired (one loop, %d blocks/lane) tile %4d: wave-steps %.0f  efficiency %.3f" % (T // 64, T, tot, ideal / tot))
    # a better sort key from (nq, lambda): predicted pair-steps = sum_i ceil(min(i, M)/2), M = 1 + c / lambda
    for cc in (0.1, 0.2, 0.3, 0.5, 0.8):
        for T in (256, 512):
            tot = 0
            for nq, st, lam in zip(allnq, allst, alllam):
                M = np.minimum(1 + cc / lam, 64.0)
                i = np.arange(1, 65)[None, :]
                pred = (np.ceil(np.minimum(i, M[:, None]) / 2) * (i <= nq[:, None])).sum(1)
                nqp, stp, kp = pad(nq, T).reshape(-1, T), pad(st, T).reshape(-1, T), pad(pred, T).reshape(-1, T)
                stp = np.where(nqp > QN, 0, stp)
                order = np.argsort(-kp, axis=1, kind="stable")
                sst = np.take_along_axis(stp, order, 1)
                tot += sst.reshape(sst.shape[0], T // 64, 64).max(2).sum()
            print("sorted by pred(c=%.1f) tile %4d: wave-steps %.0f  efficiency %.3f" % (cc, T, tot, ideal / tot))
    # What a producer-side placement could do without knowing the tile's counts in advance (ROUND_NOTES "next" 1): two
    # buckets by a FIXED threshold on nq -- heavy blocks fill the tile from the front, light ones from the back (two atomic
    # counters per tile) -- so that every pass of the trellis reads ONE line per plane row instead of four
    ideal = sum(np.where(nq > QN, 0, st).sum() for nq, st in zip(allnq, allst)) / 64.0
    T = 256
    for thr in (2, 3, 4, 6, 8, 10):
        tot = 0
        for nq, st in zip(allnq, allst):
            nqp, stp = pad(nq, T).reshape(-1, T), pad(st, T).reshape(-1, T)
            stp = np.where(nqp > QN, 0, stp)
            heavy = nqp >= thr
            # stable partition: heavy first in natural order, then light in reverse natural order (filled from the back)
            idx = np.arange(T)[None, :].repeat(nqp.shape[0], 0)
            key = np.where(heavy, idx, 2 * T - idx)
            order = np.argsort(key, axis=1, kind="stable")
            sst = np.take_along_axis(stp, order, 1)
            tot += sst.reshape(sst.shape[0], T // 64, 64).max(2).sum()
        print("two buckets (nq >= %2d in front) tile %4d: wave-steps %.0f  efficiency %.3f" % (thr, T, tot, ideal / tot))
    # three buckets with a guessed split of the tile (heavy from the front, light from the back, middle from slot 96 upward and
    # spilling wherever room is left): modelled as a sort by bucket number only
    for t1, t2 in ((3, 8), (4, 10), (2, 6), (4, 8)):
        tot = 0
        for nq, st in zip(allnq, allst):
            nqp, stp = pad(nq, T).reshape(-1, T), pad(st, T).reshape(-1, T)
            stp = np.where(nqp > QN, 0, stp)
            b = (nqp < t2).astype(np.int32) + (nqp < t1).astype(np.int32)
            order = np.argsort(b, axis=1, kind="stable")
            sst = np.take_along_axis(stp, order, 1)
            tot += sst.reshape(sst.shape[0], T // 64, 64).max(2).sum()
        print("three buckets (nq >= %d | >= %d | rest) tile %4d: wave-steps %.0f  efficiency %.3f" % (t2, t1, T, tot, ideal / tot))
    print("blocks", tot_blocks)


if __name__ == "__main__":
    main()
