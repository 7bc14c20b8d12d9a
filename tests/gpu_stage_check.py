"""Stage-by-stage comparison of the HIP path against the CPU oracle (run on a GPU box).
usage: python tests/gpu_stage_check.py            (prints one line per case; exit 1 on mismatch)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_lib as O  # noqa: E402
import mozjpeg_amd as M  # noqa: E402

ZZ = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34,
               27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
               58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])


def oracle_to_gpu_layout(a, wib, hib):
    """oracle [hpad][wpad][64 natural] -> gpu [64 zig-zag][hib*wib]"""
    return np.ascontiguousarray(a[:hib, :wib, :][:, :, ZZ].reshape(hib * wib, 64).T)


def check_case(img, kw, verbose=True):
    h, w = img.shape[:2]
    po = O.make_params(w, h, **kw)
    ref_bytes, taps = O.encode(po, img, want_taps=True)
    pg = M.make_params(w, h, **kw)
    enc = M.Encoder(pg, max_batch=2)
    enc.set_debug_taps(True)
    batch = np.stack([img, img[::-1].copy()])
    outs = enc.encode_host(batch)
    status = []
    ok = True
    p12 = kw.get("precision", 8) == 12
    for ci in range(pg.num_components):
        wib, hib, pw, ph = enc.geometry(ci)
        if not p12:   # (12-bit: planes are uint16 and the raw DCT is not kept -- there is no trellis to feed)
            pl = enc.read_tap(M.TAP_PLANE, 0, ci)
            if not np.array_equal(pl, taps[("planes", ci)]):
                status.append("plane%d DIFF(%d)" % (ci, int((pl != taps[("planes", ci)]).sum())))
                ok = False
            uq = enc.read_tap(M.TAP_COEF_UQ, 0, ci)
            if not np.array_equal(uq, oracle_to_gpu_layout(taps[("coef_uq", ci)], wib, hib)):
                status.append("uq%d DIFF" % ci)
                ok = False
        if pg.trellis_quant:
            q0 = enc.read_tap(M.TAP_COEF_Q0, 0, ci)
            if not np.array_equal(q0, oracle_to_gpu_layout(taps[("coef_q0", ci)], wib, hib)):
                status.append("q0_%d DIFF" % ci)
                ok = False
        q = enc.read_tap(M.TAP_COEF_Q, 0, ci)
        oq = oracle_to_gpu_layout(taps[("coef_q", ci)], wib, hib)
        if not np.array_equal(q, oq):
            nd_dc = int((q[0] != oq[0]).sum())
            nd_ac = int((q[1:] != oq[1:]).any(axis=0).sum())
            status.append("q%d DIFF(dc blocks %d, ac blocks %d of %d)" % (ci, nd_dc, nd_ac, wib * hib))
            ok = False
    bits = enc.read_tap(M.TAP_HUFF_BITS, 0)
    vals = enc.read_tap(M.TAP_HUFF_VALS, 0)
    if pg.optimize_coding and pg.num_scans == 0 and not kw.get("arithmetic"):   # (the arithmetic coder has no tables)
        # tables a component refers to (RGB output: only 0; table numbers of the application's own: any of the four, DC and AC apart)
        used_dc = sorted({pg.dc_tbl_no[i] for i in range(pg.num_components)})
        used_ac = sorted({pg.ac_tbl_no[i] for i in range(pg.num_components)})
        for nm, used, off, tb, tv in (("dc", used_dc, 0, taps["dc_bits"], taps["dc_vals"]), ("ac", used_ac, 1, taps["ac_bits"], taps["ac_vals"])):
            for t in used:
                if t >= len(tb) or 2 * t + off >= len(bits):
                    continue      # (the taps hold table numbers 0 and 1; the bytes below cover the rest)
                gb, gv, ob, ov = bits[2 * t + off], vals[2 * t + off], tb[t], tv[t]
                n = int(ob[1:].sum())
                if not (np.array_equal(gb[1:], ob[1:]) and np.array_equal(gv[:n], ov[:n])):
                    status.append("%s-table%d DIFF" % (nm, t))
                    ok = False
    if outs[0] != ref_bytes:
        status.append("BYTES DIFF (%d vs %d)" % (len(outs[0]), len(ref_bytes)))
        ok = False
    ref2 = O.encode(po, batch[1])
    if outs[1] != ref2:
        status.append("BYTES[1] DIFF (%d vs %d)" % (len(outs[1]), len(ref2)))
        ok = False
    enc.close()
    if verbose:
        print("%-4s %dx%d %s %s" % ("OK" if ok else "FAIL", w, h, kw, "; ".join(status)), flush=True)
    return ok


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    testorig = O.read_ppm(os.path.join(here, "golden", "testorig.ppm"))
    big = O.synthetic_frame(640, 480, 1234)
    rng = np.random.default_rng(3)
    imgs = [testorig, big[100:287, 50:300].copy(), big[:33, :17].copy(), big[:1, :1].copy(),
            rng.integers(0, 256, (75, 121, 3), dtype=np.uint8), big]
    cases = [dict(revert=True), dict(revert=True, optimize=True),
             dict(baseline=True, notrellis=True, noovershoot=True), dict(baseline=True, notrellis=True),
             dict(baseline=True, notrellis_dc=True), dict(baseline=True),
             dict(baseline=True, quality=90, sample=(1, 1)), dict(baseline=True, sample=(2, 1)),
             dict(revert=True, sample=(1, 2)), dict(baseline=True, gray=True), dict(baseline=True, quality=30),
             dict(baseline=True, restart=1), dict(baseline=True, restart="5b"), dict(revert=True, restart=2),
             dict(baseline=True, restart="1b", sample=(1, 1)),
             dict(revert=True, progressive=True), dict(fastcrush=True), dict(), dict(quality=85),
             dict(gray=True), dict(fastcrush=True, notrellis=True), dict(quality=5, fastcrush=True),
             dict(quality=92, sample=(1, 1))]
    bad = 0
    for img in imgs:
        for kw in cases:
            try:
                if not check_case(img, kw):
                    bad += 1
            except Exception as e:  # noqa: BLE001
                bad += 1
                print("EXC ", img.shape, kw, repr(e), flush=True)
    # 12-bit samples (uint16): only -notrellis has reference behaviour
    big12 = O.synthetic_frame12(640, 480, 1234)
    imgs12 = [big12[100:287, 50:300].copy(), big12[:33, :17].copy(), big12[:1, :1].copy(),
              rng.integers(0, 4096, (75, 121, 3)).astype(np.uint16), big12]
    cases12 = [dict(baseline=True, notrellis=True, quality=90, sample=(1, 1)), dict(baseline=True, notrellis=True),
               dict(baseline=True, notrellis=True, noovershoot=True, quality=90, sample=(1, 1), restart=1),
               dict(notrellis=True), dict(notrellis=True, fastcrush=True), dict(revert=True, quality=90),
               dict(baseline=True, notrellis=True, gray=True), dict(baseline=True, notrellis=True, sample=(2, 1), quality=40)]
    for img in imgs12:
        for kw in cases12:
            kw = dict(kw, precision=12)
            try:
                if not check_case(img, kw):
                    bad += 1
            except Exception as e:  # noqa: BLE001
                bad += 1
                print("EXC ", img.shape, kw, repr(e), flush=True)
    print("mismatching cases:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
