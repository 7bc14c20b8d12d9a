"""GPU (-m gpu): seeded random configurations -- sizes, quality, sampling, restart intervals, sequential / progressive /
scan search, every trellis option the device path carries -- against the CPU oracle, byte for byte.  The switch sets
are drawn once from fixed seeds, so a failure names a reproducible case."""
import numpy as np
import pytest

import mozjpeg_amd as M
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(20240921)
    out = []
    samplings = [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (1, 4), (4, 2), (2, 4)]
    for i in range(72):
        w = int(rng.integers(1, 400)); h = int(rng.integers(1, 300))
        kw = dict(quality=int(rng.choice([5, 20, 40, 60, 75, 85, 92, 98, 100])), sample=samplings[int(rng.integers(0, len(samplings)))])
        mode = int(rng.integers(0, 4))
        if mode == 0:
            kw["baseline"] = True
        elif mode == 1:
            kw["fastcrush"] = True
        elif mode == 2:
            kw["revert"] = True
            if rng.random() < 0.5:
                kw["progressive"] = True
        if rng.random() < 0.25:
            kw["gray"] = True
            kw["sample"] = (1, 1)
        if rng.random() < 0.3:
            kw["restart"] = int(rng.integers(1, 4)) if rng.random() < 0.5 else "%db" % int(rng.integers(1, 40))
        if not kw.get("revert"):
            r = rng.random()
            if r < 0.12:
                kw["notrellis"] = True
            elif r < 0.2:
                kw["notrellis_dc"] = True
            if not kw.get("notrellis"):
                if rng.random() < 0.2:
                    kw["trellis_loops"] = int(rng.integers(2, 4))
                if rng.random() < 0.2:
                    kw["use_scans_in_trellis"] = True
                    kw["trellis_freq_split"] = int(rng.choice([0, 1, 5, 8, 30, 62, 63]))
                if rng.random() < 0.2:
                    kw["trellis_eob_opt"] = True
                if rng.random() < 0.25 and kw["quality"] >= 40:   # (8-bit tables: the refused combination stays out)
                    kw["trellis_q_opt"] = True
                if rng.random() < 0.2:
                    kw["dc_ver_weight"] = float(rng.choice([0.25, 1.0, 3.0]))
            if rng.random() < 0.2 and not kw.get("baseline"):
                kw["dc_scan_opt"] = int(rng.integers(1, 3))
            if rng.random() < 0.15:
                kw["smooth"] = int(rng.integers(1, 101))
                if kw["sample"] not in ((1, 1), (2, 2)):
                    kw["sample"] = (2, 2)
        if rng.random() < 0.15:
            kw["noovershoot"] = True
        out.append((i, w, h, kw, int(rng.integers(0, 3))))
    return out


@pytest.mark.parametrize("idx,w,h,kw,kind", _cases(), ids=lambda v: str(v) if isinstance(v, int) else None)
def test_random_configuration_matches_the_oracle(idx, w, h, kw, kind):
    rng = np.random.default_rng(1000 + idx)
    if kind == 0:
        img = O.synthetic_frame(max(w, 8), max(h, 8), 3000 + idx)[:h, :w].copy()
    elif kind == 1:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)                 # noise: every coefficient non-zero
    else:
        img = np.full((h, w, 3), 255, np.uint8)                               # saturated flats + a few dark spots: deringing, all-zero blocks
        for _ in range(4):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            img[y:y + 9, x:x + 9] = rng.integers(0, 64, 3, dtype=np.uint8)
    if kw.get("gray") and rng.random() < 0.5:
        kw = dict(kw, grayin=True)
        img = img[:, :, 1].copy()
    want = O.encode(O.make_params(w, h, **kw), img)
    mkw = {k: v for k, v in kw.items()}
    enc = M.Encoder(M.make_params(w, h, **mkw), max_batch=2)
    got = enc.encode_host(np.stack([img, img]))
    enc.close()
    assert got[0] == want and got[1] == want, (idx, w, h, kw)


def _arith_cases():
    """the same idea for arithmetic coding: its own seed, so the Huffman cases above keep their numbers"""
    rng = np.random.default_rng(20260921)
    out = []
    samplings = [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (1, 4), (4, 2), (2, 4)]
    for i in range(40):
        w = int(rng.integers(1, 400)); h = int(rng.integers(1, 300))
        kw = dict(arithmetic=True, quality=int(rng.choice([5, 20, 40, 60, 75, 85, 92, 98, 100])), sample=samplings[int(rng.integers(0, len(samplings)))])
        mode = int(rng.integers(0, 4))        # 3 = cjpeg's default: progressive + scan search
        if mode == 0:
            kw["baseline"] = True
        elif mode == 1:
            kw["fastcrush"] = True
        elif mode == 2:
            kw["revert"] = True
            if rng.random() < 0.5:
                kw["progressive"] = True
        if rng.random() < 0.25:
            kw["gray"] = True
            kw["sample"] = (1, 1)
        if rng.random() < 0.35:
            kw["restart"] = int(rng.integers(1, 4)) if rng.random() < 0.5 else "%db" % int(rng.integers(1, 40))
        if not kw.get("revert"):
            r = rng.random()
            if r < 0.2:
                kw["notrellis"] = True
            elif r < 0.35:
                kw["notrellis_dc"] = True
            if not kw.get("notrellis"):
                if rng.random() < 0.2:
                    kw["use_scans_in_trellis"] = True          # (with arithmetic coding: only the first band is ever trellised)
                    kw["trellis_freq_split"] = int(rng.choice([0, 1, 5, 8, 30, 62, 63]))
                if rng.random() < 0.25:
                    kw["dc_ver_weight"] = float(rng.choice([0.25, 1.0, 3.0]))
                if rng.random() < 0.15:
                    kw["trellis_loops"] = 2
            if rng.random() < 0.2 and not kw.get("baseline"):
                kw["dc_scan_opt"] = int(rng.integers(1, 3))
        out.append((i, w, h, kw, int(rng.integers(0, 3))))
    return out


@pytest.mark.parametrize("idx,w,h,kw,kind", _arith_cases(), ids=lambda v: str(v) if isinstance(v, int) else None)
def test_random_arithmetic_configuration_matches_the_oracle(idx, w, h, kw, kind):
    rng = np.random.default_rng(5000 + idx)
    if kind == 0:
        img = O.synthetic_frame(max(w, 8), max(h, 8), 7000 + idx)[:h, :w].copy()
    elif kind == 1:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    else:
        img = np.full((h, w, 3), 255, np.uint8)
        for _ in range(4):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            img[y:y + 9, x:x + 9] = rng.integers(0, 64, 3, dtype=np.uint8)
    if kw.get("gray") and rng.random() < 0.5:
        kw = dict(kw, grayin=True)
        img = img[:, :, 1].copy()
    want = O.encode(O.make_params(w, h, **kw), img)
    assert len(want) > 0, (idx, kw)
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=2)
    got = enc.encode_host(np.stack([img, img[::-1].copy()]))
    enc.close()
    assert got[0] == want, (idx, w, h, kw)
    assert got[1] == O.encode(O.make_params(w, h, **kw), img[::-1].copy()), (idx, w, h, kw)


def _round5_cases():
    """what round 5 added to the configuration space, its own seed: per-component sampling factors (chroma other than 1x1, luma
    smaller than chroma, 3:1 ratios), scan scripts of the application's own (sequential files of several whole-block scans in every
    grouping of three components, small progressive scripts), arithmetic conditioning values"""
    rng = np.random.default_rng(20260922)
    out = []
    factor_sets = [((2, 2), (2, 1), (1, 1)), ((2, 1), (1, 1), (1, 2)), ((1, 2), (2, 2), (1, 1)), ((3, 1), (1, 1), (1, 1)), ((2, 1), (2, 1), (2, 1)),
                   ((1, 1), (2, 2), (2, 2)), ((2, 2), (1, 2), (2, 1)), ((1, 3), (1, 1), (1, 3)), ((4, 1), (2, 1), (1, 1)), ((2, 2), (2, 2), (1, 1))]
    seq_scripts = [[(0,), (1, 2)], [(0,), (1,), (2,)], [(0, 1), (2,)], [(0, 2), (1,)], [(0, 1, 2)], [(0,), (1,), (2,)][::1]]
    prog_scripts = [[((0, 1, 2), 0, 0, 0, 0), ((0,), 1, 63, 0, 0), ((1,), 1, 63, 0, 0), ((2,), 1, 63, 0, 0)],
                    [((0,), 0, 0, 0, 1), ((1, 2), 0, 0, 0, 0), ((0,), 1, 5, 0, 2), ((0,), 6, 63, 0, 2), ((1,), 1, 63, 0, 1), ((2,), 1, 63, 0, 1),
                     ((0,), 1, 63, 2, 1), ((0,), 0, 0, 1, 0), ((0,), 1, 63, 1, 0), ((1,), 1, 63, 1, 0), ((2,), 1, 63, 1, 0)]]
    for i in range(48):
        w = int(rng.integers(1, 360)); h = int(rng.integers(1, 280))
        kw = dict(quality=int(rng.choice([10, 40, 75, 85, 95])))
        what = int(rng.integers(0, 4))
        kw["sample"] = factor_sets[int(rng.integers(0, len(factor_sets)))] if what != 1 or rng.random() < 0.5 else (2, 2)
        if what == 1:          # a sequential script
            groups = seq_scripts[int(rng.integers(0, len(seq_scripts)))]
            kw["scans"] = [(g, 0, 63, 0, 0) for g in groups]
            if rng.random() < 0.4:
                kw["revert"] = True
        elif what == 2:        # a progressive script
            kw["scans"] = prog_scripts[int(rng.integers(0, len(prog_scripts)))]
        elif what == 3:        # arithmetic coding with conditioning values
            lo = int(rng.integers(0, 4))
            kw.update(arithmetic=True, arith_cond=((lo, lo + int(rng.integers(0, 8)), int(rng.integers(1, 64))), (0, int(rng.integers(0, 3)), int(rng.integers(1, 64)))))
            m = int(rng.integers(0, 3))
            kw.update({0: dict(baseline=True), 1: dict(fastcrush=True), 2: dict()}[m])
        else:
            m = int(rng.integers(0, 3))
            kw.update({0: dict(baseline=True), 1: dict(fastcrush=True), 2: dict(revert=True)}[m])
        if rng.random() < 0.35:
            kw["restart"] = int(rng.integers(1, 3)) if rng.random() < 0.6 else "%db" % int(rng.integers(1, 30))
        if not kw.get("revert") and rng.random() < 0.2:
            kw["notrellis"] = True
        out.append((i, w, h, kw, int(rng.integers(0, 3))))
    return out


@pytest.mark.parametrize("idx,w,h,kw,kind", _round5_cases(), ids=lambda v: str(v) if isinstance(v, int) else None)
def test_random_sampling_factors_scripts_and_conditioning_match_the_oracle(idx, w, h, kw, kind):
    rng = np.random.default_rng(5000 + idx)
    if kind == 0:
        img = O.synthetic_frame(max(w, 8), max(h, 8), 7000 + idx)[:h, :w].copy()
    elif kind == 1:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    else:
        img = np.full((h, w, 3), 255, np.uint8)
        for _ in range(4):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            img[y:y + 9, x:x + 9] = rng.integers(0, 64, 3, dtype=np.uint8)
    want = O.encode(O.make_params(w, h, **kw), img)
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=2)
    got = enc.encode_host(np.stack([img, img[::-1].copy()]))
    enc.close()
    assert got[0] == want, (idx, w, h, kw)


def _arith_qopt_cases():
    """trellis_q_opt with the arithmetic coder (round 5, late): gray and colour, 1-5 trellis loops, with and without the two
    trellis bands -- whether and how often the reference re-estimates component 0's table depends on all three (and a gray
    image with an odd number of loops gets its last estimate in the DQT marker only)"""
    rng = np.random.default_rng(20260923)
    out = []
    for i in range(24):
        w = int(rng.integers(1, 300)); h = int(rng.integers(1, 220))
        kw = dict(arithmetic=True, trellis_q_opt=True, quality=int(rng.choice([3, 10, 25, 50, 75, 90, 97])), trellis_loops=int(rng.integers(1, 6)),
                  sample=[(1, 1), (2, 1), (2, 2), (1, 2)][int(rng.integers(0, 4))])
        kw.update({0: dict(baseline=True), 1: dict(fastcrush=True), 2: dict()}[int(rng.integers(0, 3))])
        if rng.random() < 0.45:
            kw["gray"] = True
            kw["sample"] = (1, 1)
        if rng.random() < 0.35:
            kw["use_scans_in_trellis"] = True
            kw["trellis_freq_split"] = int(rng.choice([0, 1, 8, 30, 63]))
        if rng.random() < 0.3:
            kw["restart"] = int(rng.integers(1, 3)) if rng.random() < 0.6 else "%db" % int(rng.integers(1, 30))
        if rng.random() < 0.2:
            kw["notrellis_dc"] = True
        out.append((i, w, h, kw, int(rng.integers(0, 3))))
    return out


@pytest.mark.parametrize("idx,w,h,kw,kind", _arith_qopt_cases(), ids=lambda v: str(v) if isinstance(v, int) else None)
def test_random_arithmetic_q_opt_configuration_matches_the_oracle(idx, w, h, kw, kind):
    rng = np.random.default_rng(9000 + idx)
    if kind == 0:
        img = O.synthetic_frame(max(w, 8), max(h, 8), 9100 + idx)[:h, :w].copy()
    elif kind == 1:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    else:
        img = np.full((h, w, 3), 255, np.uint8)
        for _ in range(4):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            img[y:y + 9, x:x + 9] = rng.integers(0, 64, 3, dtype=np.uint8)
    want = O.encode(O.make_params(w, h, **kw), img)
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=2)
    got = enc.encode_host(np.stack([img, img[::-1].copy()]))
    again = enc.encode_host(np.stack([img[::-1].copy(), img]))      # (the tables of a call start from the parameters' again)
    enc.close()
    assert got[0] == want and again[1] == want, (idx, w, h, kw)
    assert got[1] == again[0] == O.encode(O.make_params(w, h, **kw), img[::-1].copy()), (idx, w, h, kw)
