#!/usr/bin/env python3
"""Model of pp_band_nonzeros' SKIPLOW form (mjh_prog_sl.hip, MJH_PP_SKIPLOW=1) on C3's workload, without a GPU:
the walks of the first-pass AC candidate scans of the scan search (jcparam.c:796-847) over the oracle's final
coefficients of one synthetic 4K q85 4:2:0 frame, wave by wave (64 consecutive blocks of a component = one wave of
k_pp_stats<true, 1> / k_pp_emit).  An iteration j of the walk costs the wave
  * P  if some lane's j-th non-zero lies in the band (the visitor runs),
  * D  if some lane still has a j-th non-zero but none of them is in the band (default walk: ctz, clear, compare, branch;
       SKIPLOW: one subtract-and-compare),
  * 0  for SKIPLOW iterations in bursts of 8 below the first burst any lane needs (not even loaded).
Prints visits, the visitor's lane efficiency (as is / with chunks sorted by in-band count) and the modelled wave-instruction
totals for both forms.  usage: python tools/model_skiplow.py [quality]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import oracle_lib as O

ZZ = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
               35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
SPLITS = (2, 8, 5, 12, 18)


def scans(al_max):
    out = [(1, 8, 0), (9, 63, 0)]
    for al in range(1, al_max + 1):
        out += [(1, 8, al), (9, 63, al)]
    out.append((1, 63, 0))
    for f in SPLITS:
        out += [(1, f, 0), (f + 1, 63, 0)]
    return out


def main():
    q = int(sys.argv[1]) if len(sys.argv) > 1 else 85
    w, h = 3840, 2160
    img = O.synthetic_frame(w, h, 1234)
    p = O.make_params(w, h, quality=q, sample=(2, 2))
    _, taps = O.encode(p, img, want_taps=True)
    coefs = O.real_coefficients(p, taps)
    tot = dict(visits=0, inband=0, dropped=0, wave_iters=0, sorted_iters=0)
    cost = {k: dict(default=0.0, skiplow=0.0) for k in ("stats", "emit")}
    P = dict(stats=15.0, emit=25.0 + 45.0)      # visitor instructions per in-band non-zero (emit: sizing walk + writing walk)
    D_DEF, D_SL = 12.0, 3.0
    for ci, c in enumerate(coefs):
        zz = c.reshape(-1, 64)[:, ZZ]                         # blocks in raster order, positions in zig-zag order
        nzm = zz != 0
        nzm[:, 0] = False
        nb = nzm.shape[0]
        pad = (-nb) % 64
        nzp = np.concatenate([nzm, np.zeros((pad, 64), bool)]).reshape(-1, 64, 64)      # [wave, lane, position]
        rank = np.cumsum(nzp, axis=2) - 1                       # index of a non-zero in its block's record
        for ss, se, al in scans(3 if ci == 0 else 2):
            upto = nzp[:, :, :se + 1].sum(axis=2)               # record entries the default walk iterates over
            lo = nzp[:, :, :ss].sum(axis=2)
            tot["visits"] += int(upto.sum()); tot["dropped"] += int(lo.sum()); tot["inband"] += int((upto - lo).sum())
            # per wave and iteration j: does any lane visit (in band) / merely hold a j-th entry
            j = np.arange(64)[None, None, :]
            holds = (j < upto[:, :, None])
            inb = holds & (j >= lo[:, :, None])
            any_in = inb.any(axis=1)                            # [wave, j]
            tot["wave_iters"] += int(any_in.sum())
            # the same blocks with every chunk of 2048 (32 waves) sorted by in-band count: what a wave would then iterate over
            cnt = (upto - lo).reshape(-1)
            padc = (-cnt.size) % 2048
            cs = np.sort(np.concatenate([cnt, np.zeros(padc, cnt.dtype)]).reshape(-1, 2048), axis=1).reshape(-1, 64)
            tot["sorted_iters"] += int(cs.max(axis=1).sum())
            any_hold = holds.any(axis=1)
            live = (upto > lo)
            first = np.where(live, lo, 10 ** 6).min(axis=1)     # first entry any lane of the wave visits
            base0 = np.where(first < 10 ** 6, (first // 8) * 8, 64)
            jj = np.arange(64)[None, :]
            for k in ("stats", "emit"):
                cost[k]["default"] += float((any_in * P[k] + (any_hold & ~any_in) * D_DEF).sum())
                cost[k]["skiplow"] += float(((any_in * P[k] + (any_hold & ~any_in) * D_SL) * (jj >= base0[:, None])).sum())
    print("quality %d: record entries walked %d, in band %d, below the band %d (%.1f %%)" % (q, tot["visits"], tot["inband"], tot["dropped"], 100.0 * tot["dropped"] / tot["visits"]))
    print("  lane efficiency of the visitor (in-band values / 64 x wave iterations with any): %.3f; with every chunk of 2048 blocks sorted by in-band count: %.3f"
          % (tot["inband"] / (64.0 * tot["wave_iters"]), tot["inband"] / (64.0 * tot["sorted_iters"])))
    for k in ("stats", "emit"):
        d, s = cost[k]["default"], cost[k]["skiplow"]
        print("  %-5s walk, modelled wave-instructions per frame: default %.3g, SKIPLOW %.3g (%.1f %% fewer)" % (k, d, s, 100.0 * (1 - s / d)))


if __name__ == "__main__":
    main()
