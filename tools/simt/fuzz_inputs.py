#!/usr/bin/env python3
"""The two other ways into the encoder, on random configurations: component planes (jpeg_write_raw_data) and
quantized coefficients (jpeg_write_coefficients / jpegtran, row 2) -- the kernel SOURCES on the emulator against the CPU
oracle against the reference binaries (oracle/_ref/refenc -yuvin, oracle/_ref/jpegtran).  Development aid, correctness only;
the build container only (it needs the reference).
usage: python tools/simt/fuzz_inputs.py SEED COUNT        prints one line per failure and a summary"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import build_simt  # noqa: E402
import mozjpeg_amd as M  # noqa: E402
M.LIB_PATH = build_simt.build()
os.environ.setdefault("SIMT_STRICT", "1")
import oracle_lib as O  # noqa: E402

TJ_SAMPLINGS = [(1, 1), (2, 1), (2, 2), (1, 2), (4, 1), (1, 4)]


def mode_kw(rng, transcode):
    """an output mode in make_params vocabulary + jpegtran's switches for it"""
    m = int(rng.integers(0, 5))
    if m == 0:
        kw, sw = dict(revert=True), ["-revert"]
    elif m == 1:
        kw, sw = dict(revert=True, optimize=True), ["-revert", "-optimize"]
    elif m == 2:
        kw, sw = dict(revert=True, progressive=True), ["-revert", "-progressive"]
    elif m == 3:
        kw, sw = dict(fastcrush=True), ["-fastcrush", "-progressive"]
    else:
        kw, sw = (dict(), ["-progressive"]) if transcode else (dict(baseline=True), [])
    if rng.random() < 0.3:
        if rng.random() < 0.5:
            n = int(rng.integers(1, 4))
            kw["restart"] = n
            sw += ["-restart", str(n)]
        else:
            n = int(rng.integers(1, 30))
            kw["restart"] = "%db" % n
            sw += ["-restart", "%dB" % n]
    if rng.random() < 0.2:
        kw["arithmetic"] = True
        sw = ["-arithmetic"] + sw
        kw.pop("optimize", None)
        sw = [s for s in sw if s != "-optimize"]
    return kw, sw


def plane_case(rng, seed, i):
    w = int(rng.integers(1, 400)); h = int(rng.integers(1, 300))
    kw, _ = mode_kw(rng, False)
    kw["quality"] = int(rng.choice([5, 30, 60, 75, 85, 92, 98]))
    if rng.random() < 0.2:
        kw.update(gray=True, sample=(1, 1))
    else:
        kw["sample"] = TJ_SAMPLINGS[int(rng.integers(0, len(TJ_SAMPLINGS)))]
    if not kw.get("revert") and rng.random() < 0.3:
        kw["notrellis"] = True
    po = O.make_params(w, h, **kw)
    planes = O.synthetic_planes(po, seed * 1000 + i)
    if rng.random() < 0.3:
        planes = [rng.integers(0, 256, a.shape, dtype=np.uint8) for a in planes]
    want = O.encode_planes(po, planes)
    ref = O.ref_encode_planes(planes, w, h, **kw)
    if ref != want:
        return "ORACLE != REFERENCE (planes)", (w, h, kw)
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=2)
    got = enc.encode_planes_host([np.stack([a, a]) for a in planes])
    enc.close()
    if got[0] != want or got[1] != want:
        return "DIFFERENT (planes)", (w, h, kw)
    return None, None


def transcode_case(rng, seed, i):
    w = int(rng.integers(1, 400)); h = int(rng.integers(1, 300))
    src_kw = dict(quality=int(rng.choice([10, 40, 75, 90, 97])))
    m = int(rng.integers(0, 3))
    src_kw.update({0: dict(baseline=True), 1: dict(revert=True), 2: dict(fastcrush=True)}[m])
    if rng.random() < 0.2:
        src_kw.update(gray=True, sample=(1, 1))
    else:
        src_kw["sample"] = TJ_SAMPLINGS[int(rng.integers(0, 4))]
    kind = int(rng.integers(0, 2))
    img = O.synthetic_frame(max(w, 8), max(h, 8), 5000 + seed + i)[:h, :w].copy() if kind == 0 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ps = O.make_params(w, h, **src_kw)
    src, taps = O.encode(ps, img, want_taps=True)
    coefs = O.real_coefficients(ps, taps)
    kw, sw = mode_kw(rng, True)
    pt = O.transcode_params(ps, **kw)
    want = O.encode_coefficients(pt, coefs)
    ref = O.ref_jpegtran(src, sw)
    if ref != want:
        return "ORACLE != REFERENCE (jpegtran %s)" % " ".join(sw), (w, h, src_kw, kw)
    mp = M.make_params(w, h, notrellis=True, gray=(ps.num_components == 1), grayin=(ps.num_components == 1), sample=(ps.h_samp[0], ps.v_samp[0]), **kw)
    for t in range(4):
        for k in range(64):
            mp.quantval[t][k] = ps.qtbl[t][k]
    enc = M.Encoder(mp, max_batch=2)
    got = enc.encode_coefficients_host([np.stack([a, a]) for a in coefs])
    enc.close()
    if got[0] != want or got[1] != want:
        return "DIFFERENT (coefficients)", (w, h, src_kw, kw)
    return None, None


def main():
    seed, count = int(sys.argv[1]), int(sys.argv[2])
    assert O.have_ref(), "the reference binaries are needed (make -C oracle ref)"
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for i in range(count):
        fn = plane_case if rng.random() < 0.5 else transcode_case
        try:
            what, info = fn(rng, seed, i)
        except Exception as exc:
            what, info = "EXCEPTION %s" % repr(exc)[:300], (fn.__name__,)
        if what:
            bad += 1
            print(what, seed, i, info, flush=True)
    print("seed %d: %d cases (planes / coefficients), %d failures, %.0f s" % (seed, count, bad, time.time() - t0), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
