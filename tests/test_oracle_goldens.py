"""CPU: the oracle restatement (oracle/mjoracle.c) against the committed goldens, which were
produced by the real reference (tests/golden/make_goldens.py), and against the constants the
reference's own CTest suite pins."""
import pytest

import oracle_lib as O
from cases import CASES, CASES12, REFERENCE_PINNED, images12


def test_reference_pinned_constants_are_in_goldens(goldens):
    # the generator ran the real reference; it must have reproduced the reference's own MD5s
    for (iname, cname), md5 in REFERENCE_PINNED.items():
        assert goldens["%s/%s" % (iname, cname)]["md5"] == md5


@pytest.mark.parametrize("cname,kw", [(c, kw) for c, kw, _ in CASES])
def test_oracle_matches_goldens(cname, kw, goldens, fixture_images):
    for iname, img in fixture_images.items():
        if iname == "syn640x480" and cname not in ("base", "default_progressive", "revert"):
            continue  # keep the CPU suite short; the big frame is covered for the three main modes
        h, w = img.shape[:2]
        data = O.encode(O.make_params(w, h, **kw), img)
        g = goldens["%s/%s" % (iname, cname)]
        assert len(data) == g["bytes"], (iname, cname)
        assert O.md5(data) == g["md5"], (iname, cname)


@pytest.mark.parametrize("cname,kw", [(c, kw) for c, kw, _ in CASES12])
def test_oracle_matches_12bit_goldens(cname, kw, goldens):
    for iname, img in images12().items():
        h, w = img.shape[:2]
        data = O.encode(O.make_params(w, h, **kw), img)
        g = goldens["%s/%s" % (iname, cname)]
        assert (len(data), O.md5(data)) == (g["bytes"], g["md5"]), (iname, cname)


@pytest.mark.skipif(not O.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_oracle_matches_live_reference_random_sizes():
    import numpy as np
    rng = np.random.default_rng(11)
    base = O.synthetic_frame(320, 240, 99)
    for _ in range(6):
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 150))
        img = base[:h, :w].copy()
        for kw in (dict(baseline=True), dict(), dict(baseline=True, restart=1), dict(fastcrush=True, restart=2),
                   dict(baseline=True, sample=(4, 1), smooth=25), dict(revert=True, sample=(1, 4), optimize=True),
                   dict(baseline=True, trellis_loops=2, sample=(2, 1)),
                   dict(fastcrush=True, trellis_eob_opt=True, use_scans_in_trellis=True, trellis_q_opt=True)):
            a = O.encode(O.make_params(w, h, **kw), img)
            b, _ = O.ref_encode(img, **kw)
            assert a == b, (w, h, kw)


@pytest.mark.skipif(not O.have_ref(), reason="compiled reference (oracle/_ref) not present")
def test_oracle_trellised_coefficients_match_reference_readback(fixture_images):
    """post-trellis quantized coefficients vs jpeg_read_coefficients of the reference's own file"""
    import numpy as np
    img = fixture_images["syn250x187"]
    kw = dict(baseline=True)
    h, w = img.shape[:2]
    _, taps = O.encode(O.make_params(w, h, **kw), img, want_taps=True)
    _, _, coefs = O.ref_encode(img, dumpcoef=True, **kw)
    for ci, c in enumerate(coefs):
        hb, wb = c.shape[:2]
        assert np.array_equal(taps[("coef_q", ci)][:hb, :wb], c)


def test_gen_optimal_table_known_answer():
    """K.2 procedure on a tiny known case: 4 symbols with counts 8,4,2,1 (+pseudo symbol)."""
    import ctypes as C
    freq = (C.c_long * 257)()
    for s, c in ((0, 8), (1, 4), (2, 2), (3, 1)):
        freq[s] = c
    bits = (C.c_uint8 * 17)()
    vals = (C.c_uint8 * 256)()
    O.lib().mjo_gen_optimal_table(freq, bits, vals)
    assert list(bits)[1:6] == [1, 1, 1, 1, 0]
    assert list(vals)[:4] == [0, 1, 2, 3]


def _plane_goldens():
    import json
    import os
    from cases import HERE
    return json.load(open(os.path.join(HERE, "goldens_planes.json")))


def test_oracle_plane_input_matches_goldens():
    """jpeg_write_raw_data path (component planes in, TurboJPEG YUV layout): oracle vs the real reference's bytes"""
    from cases import PLANE_CASES
    g = _plane_goldens()
    for cname, w, h, kw in PLANE_CASES:
        p = O.make_params(w, h, **kw)
        data = O.encode_planes(p, O.synthetic_planes(p, 7))
        assert (len(data), O.md5(data)) == (g[cname]["bytes"], g[cname]["md5"]), cname


def test_oracle_plane_input_equals_pixel_path_on_its_own_planes():
    """feeding the planes the pixel path produced (colour conversion + downsampling taps) must give the same file"""
    img = O.synthetic_frame(120, 88, 5)
    for kw in (dict(baseline=True), dict(revert=True, sample=(2, 1))):
        p = O.make_params(120, 88, **kw)
        data, taps = O.encode(p, img, want_taps=True)
        planes = [taps[("planes", ci)] for ci in range(3)]
        assert O.encode_planes(p, planes) == data


def test_oracle_coefficient_input_matches_jpegtran_goldens(fixture_images):
    """jpeg_write_coefficients path: the oracle re-encodes the coefficients of a file it made; the golden is what the
    real jpegtran wrote for that same file (tests/golden/make_goldens.py)."""
    import json
    import os
    from cases import HERE, TRANSCODE_CASES
    g = json.load(open(os.path.join(HERE, "goldens_transcode.json")))
    for cname, iname, src_kw, _switches, kw in TRANSCODE_CASES:
        img = fixture_images[iname]
        h, w = img.shape[:2]
        ps = O.make_params(w, h, **src_kw)
        src, taps = O.encode(ps, img, want_taps=True)
        assert O.md5(src) == g[cname]["source_md5"], cname
        data = O.encode_coefficients(O.transcode_params(ps, **kw), O.real_coefficients(ps, taps))
        assert (len(data), O.md5(data)) == (g[cname]["bytes"], g[cname]["md5"]), cname
