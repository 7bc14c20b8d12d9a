#!/usr/bin/env python3
"""CPU work model (no GPU) of a lane-refilling AC trellis walk: instead of NPASS sorted passes with a barrier between them
(a pass lasts as long as its busiest lane), a wave owns a tile of T blocks sorted by descending weight, every lane walks
one block at a time, and whenever at least THR lanes have finished the wave stops, runs the finished lanes' epilogues
(back-track, output) together, hands them the next blocks of the tile and goes on.  Costs in units of one wave pair-step:
P = per-round prologue (loading the new blocks' records), B = per-round epilogue.  Per-block pair-steps come from the same
restatement of the walk as tools/model_trellis.py (tools/model/trellis_work.c).
usage: python tools/model_refill.py [w h q]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import oracle_lib as O  # noqa: E402
from model_trellis import ZZ, ehufsi  # noqa: E402


def block_work(w, h, q, sample=(2, 2)):
    lib = C.CDLL(os.path.join(ROOT, "tools", "model", "libtrellis_work.so"))
    img = O.synthetic_frame(w, h, 1234)
    p = O.make_params(w, h, quality=q, baseline=True, sample=sample)
    _, taps = O.encode(p, img, want_taps=True)
    gs, _, _ = O.geometry(p)
    zz = (C.c_int * 64)(*ZZ)
    out = []
    for ci, g in enumerate(gs):
        uq = np.ascontiguousarray(taps[("coef_uq", ci)][:g.hib, :g.wib].reshape(-1, 64))
        n = uq.shape[0]
        qt = np.array(list(p.qtbl[p.quant_tbl_no[ci]]), np.uint16)
        si = ehufsi(taps["ac_bits"][p.ac_tbl_no[ci]], taps["ac_vals"][p.ac_tbl_no[ci]])
        nq, st, ev, s1 = (np.zeros(n, np.int32) for _ in range(4))
        lam = np.zeros(n, np.float64)
        lib.trellis_work(uq.ctypes.data_as(C.c_void_p), n, qt.ctypes.data_as(C.c_void_p), si.ctypes.data_as(C.c_void_p),
                         C.c_double(p.lambda_log_scale1), C.c_double(p.lambda_log_scale2), zz,
                         nq.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), ev.ctypes.data_as(C.c_void_p), s1.ctypes.data_as(C.c_void_p), lam.ctypes.data_as(C.c_void_p))
        out.append((nq, st))
    return out


def passes(st_sorted, P, B):
    """NPASS sorted passes of 64 (today's kernel): per pass prologue + longest lane + epilogue; all-zero passes are skipped"""
    t = 0.0
    for i in range(0, len(st_sorted), 64):
        m = st_sorted[i:i + 64].max()
        if m > 0:
            t += P + m + B
    return t


def refill(st_sorted, thr, P, B, per_step=0.0):
    """lanes take blocks in sorted order; a round ends when >= thr lanes are idle (blocks left) or all lanes are idle"""
    n = len(st_sorted)
    nxt = min(64, n)
    left = st_sorted[:nxt].astype(np.int64).copy()          # remaining steps per lane
    if nxt < 64:
        left = np.concatenate([left, np.zeros(64 - nxt, np.int64)])
    t = P
    while True:
        busy = left > 0
        if not busy.any():
            t += B
            if nxt >= n:
                return t
        idle = (~busy).sum()
        if nxt < n and (idle >= thr or idle == 64):
            # round boundary: epilogue of the finished lanes, prologue of the new blocks
            t += (B if busy.any() else 0.0) + P
            k = min(idle, n - nxt)
            slots = np.flatnonzero(~busy)[:k]
            left[slots] = st_sorted[nxt:nxt + k]
            nxt += k
            # blocks with zero steps (all-zero blocks) cost nothing more
            continue
        if not busy.any():
            return t
        # advance to the next event: the smallest remaining count among busy lanes that changes the idle count enough
        rem = np.sort(left[busy])
        need = max(thr - idle, 1) if nxt < n else len(rem)
        d = rem[min(need, len(rem)) - 1]
        t += d * (1.0 + per_step)
        left[busy] -= d
        left[left < 0] = 0


def main():
    w, h, q = (int(a) for a in (sys.argv[1:4] + ["3840", "2160", "75"][len(sys.argv) - 1:]))
    sample = (1, 1) if len(sys.argv) > 4 and sys.argv[4] == "444" else (2, 2)
    QN = 16 if q <= 80 else 24 if q <= 87 else 32 if q <= 92 else 48
    comps = block_work(w, h, q, sample)
    ideal = sum(np.where(nq > QN, 0, st).sum() for nq, st in comps) / 64.0
    nblk = sum(len(nq) for nq, _ in comps)
    print("%dx%d q%d: %d blocks, ideal %.0f wave-steps (%.2f per 64 blocks)" % (w, h, q, nblk, ideal, ideal * 64 / nblk))

    def total(fn, T):
        tot = 0.0
        for nq, st in comps:
            stz = np.where(nq > QN, 0, st)
            for i in range(0, len(nq), T):
                order = np.argsort(-nq[i:i + T], kind="stable")
                tot += fn(stz[i:i + T][order])
        return tot
    # calibration of the units: today's kernel spends ~5250 VALU instructions per pass of 64 blocks, of which ~1000 build the
    # records (phase 1) -> with 26.7 wave-steps per pass at ~140 instructions each: P1 = 7, B = 2.4; records read from memory: P = 0.5
    for label, P, B in (("today (phase 1 in the kernel)", 7.0, 2.4), ("records from the FDCT kernel", 0.6, 2.4)):
        base = total(lambda s: passes(s, P, B), 256)
        print("%-32s 4 sorted passes of a 256-tile: %.0f units (walk only: efficiency %.3f)" % (label, base, ideal / total(lambda s: passes(s, 0, 0), 256)))
        if P > 1:
            continue
        for T in (256, 512, 1024):
            for thr in (8, 16, 24, 32, 48):
                c = total(lambda s: refill(s, thr, P, B, 0.03), T)
                cw = total(lambda s: refill(s, thr, 0, 0, 0.0), T)
                print("   refill tile %4d thr %2d: %.0f units = %.3f of today's, %.3f of the 4-pass kernel on records (walk efficiency %.3f)" % (
                    T, thr, c, c / total(lambda s: passes(s, 7.0, 2.4), 256), c / base, ideal / cw))


if __name__ == "__main__":
    main()
