#!/usr/bin/env python3
"""More random configurations than tests/test_gpu_fuzz.py carries, through the kernel SOURCES on the emulator against the CPU
oracle (development aid: correctness only).  The generator is test_gpu_fuzz.py's with the seed as an argument, plus
arithmetic coding (conditioning values, trellis_q_opt with several loops), round 5's sampling-factor sets and scan scripts,
and, on a share of the cases, the encoder's remaining plan knobs (first-tier capacity and passes of the AC trellis, dense-copy
capacity, where the DC chains run, the speculative DC rows) -- every plan has to give the same bytes.
usage: python tools/simt/fuzz_more.py SEED COUNT [--knob-share 0.5] [--ref] [--gpu]        prints one line per failure and a summary
(--ref: also the oracle against the reference binary oracle/_ref/refenc on every case)"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import mozjpeg_amd as M  # noqa: E402
if "--gpu" not in sys.argv:      # (--gpu: the shipped library on the chip instead of the emulator, run through gpurun: the same cases, the same checks)
    import build_simt  # noqa: E402
    M.LIB_PATH = build_simt.build()
    os.environ.setdefault("SIMT_STRICT", "1")
import oracle_lib as O  # noqa: E402

SAMPLINGS = [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (1, 4), (4, 2), (2, 4)]


def draw(rng):
    w = int(rng.integers(1, 500)); h = int(rng.integers(1, 400))
    if rng.random() < 0.1:
        w, h = (int(rng.integers(1, 4000)), int(rng.integers(1, 20))) if rng.random() < 0.5 else (int(rng.integers(1, 20)), int(rng.integers(1, 4000)))
    kw = dict(quality=int(rng.choice([1, 5, 20, 40, 60, 75, 85, 90, 92, 95, 98, 100])), sample=SAMPLINGS[int(rng.integers(0, len(SAMPLINGS)))])
    if rng.random() < 0.2:
        kw["arithmetic"] = True
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
            if not kw.get("arithmetic"):
                if rng.random() < 0.2:
                    kw["use_scans_in_trellis"] = True
                    kw["trellis_freq_split"] = int(rng.choice([0, 1, 5, 8, 30, 62, 63]))
                if rng.random() < 0.2:
                    kw["trellis_eob_opt"] = True
                if rng.random() < 0.25 and kw["quality"] >= 40:
                    kw["trellis_q_opt"] = True
            if rng.random() < 0.2:
                kw["dc_ver_weight"] = float(rng.choice([0.25, 1.0, 3.0]))
        if rng.random() < 0.2 and not kw.get("baseline") and not kw.get("arithmetic"):
            kw["dc_scan_opt"] = int(rng.integers(1, 3))
        if rng.random() < 0.15:
            kw["smooth"] = int(rng.integers(1, 101))
            if kw["sample"] not in ((1, 1), (2, 2)):
                kw["sample"] = (2, 2)
    if rng.random() < 0.15:
        kw["noovershoot"] = True
    # round 5's configuration space (drawn last: the cases of a seed keep everything above)
    r = rng.random()
    if r < 0.12 and not kw.get("gray") and not kw.get("smooth"):
        kw["sample"] = FACTOR_SETS[int(rng.integers(0, len(FACTOR_SETS)))]
    elif r < 0.2 and not kw.get("gray") and not kw.get("progressive") and (kw.get("baseline") or kw.get("revert")):
        groups = SEQ_SCRIPTS[int(rng.integers(0, len(SEQ_SCRIPTS)))]
        kw["scans"] = [(g, 0, 63, 0, 0) for g in groups]
        kw.pop("dc_scan_opt", None)
    if kw.get("arithmetic"):
        if rng.random() < 0.3:
            lo = int(rng.integers(0, 4))
            kw["arith_cond"] = ((lo, lo + int(rng.integers(0, 8)), int(rng.integers(1, 64))), (0, int(rng.integers(0, 3)), int(rng.integers(1, 64))))
        if rng.random() < 0.3 and not kw.get("revert") and not kw.get("notrellis"):
            kw["trellis_q_opt"] = True
            kw["trellis_loops"] = int(rng.integers(1, 5))
    # a progressive scan script of the application's own (any script validate_script, jcmaster.c:270-432, accepts)
    if "scans" not in kw and not kw.get("baseline") and not kw.get("revert") and rng.random() < 0.15:
        kw["scans"] = progressive_script(rng, 1 if kw.get("gray") else 3)
        kw.pop("dc_scan_opt", None)
    # 12-bit samples (drawn last of all): no trellis in the reference (jccoefct.c:132-138), whatever else was drawn stays
    if rng.random() < 0.08 and not kw.get("smooth"):
        kw["precision"] = 12
        kw["notrellis"] = True
        for k in ("notrellis_dc", "trellis_loops", "use_scans_in_trellis", "trellis_freq_split", "trellis_eob_opt", "trellis_q_opt", "dc_ver_weight"):
            kw.pop(k, None)
    # the other base tables of cjpeg -quant-table N (jcparam.c) and the trellis' lambda scales (cjpeg -lambda1 / -lambda2)
    if rng.random() < 0.2:
        kw["quant_table"] = int(rng.integers(0, 9))
    if rng.random() < 0.12 and not kw.get("notrellis") and not kw.get("revert"):
        kw["lambda1"] = float(rng.choice([-2.0, 0.0, 8.5, 14.75, 20.0]))
        kw["lambda2"] = float(rng.choice([0.0, 8.0, 16.5, 22.0]))
    # (drawn after everything else, so that the earlier draws of a seed stay what they were)  input samples that are YCbCr already;
    # one component with sampling factors of its own (cjpeg: 2x1 at qualities 80..89)
    if rng.random() < 0.08 and not kw.get("gray"):
        kw["yccin"] = True
    if kw.get("gray") and rng.random() < 0.3:
        kw["gray_sample"] = (int(rng.integers(1, 5)), int(rng.integers(1, 5)))
    return w, h, kw, int(rng.integers(0, 3))


def progressive_script(rng, ncomp):
    """DC first (interleaved or per component, point transform 0-2), every component's AC positions in 1-3 bands with their own
    point transforms, then the refinement scans bit by bit -- in a random order that keeps every refinement behind its predecessor"""
    first, later = [], []      # later: chains of refinement scans, each chain in order
    al = int(rng.integers(0, 3))
    if ncomp == 1 or rng.random() < 0.6:
        first.append((tuple(range(ncomp)), 0, 0, 0, al))
        later.append([(tuple(range(ncomp)), 0, 0, a + 1, a) for a in range(al - 1, -1, -1)])
    else:
        for c in range(ncomp):
            a0 = int(rng.integers(0, 3))
            first.append(((c,), 0, 0, 0, a0))
            later.append([((c,), 0, 0, a + 1, a) for a in range(a0 - 1, -1, -1)])
    ac_first = []
    for c in range(ncomp):
        cuts = sorted(set(int(v) for v in rng.integers(1, 63, int(rng.integers(0, 3)))))
        lo = 1
        for hi in cuts + [63]:
            if hi < lo:
                continue
            a0 = int(rng.integers(0, 3))
            ac_first.append(((c,), lo, hi, 0, a0))
            later.append([((c,), lo, hi, a + 1, a) for a in range(a0 - 1, -1, -1)])
            lo = hi + 1
    order = [int(v) for v in rng.permutation(len(ac_first))]
    scans = first + [ac_first[j] for j in order]
    chains = [ch for ch in later if ch]
    while chains:
        j = int(rng.integers(0, len(chains)))
        scans.append(chains[j].pop(0))
        chains = [ch for ch in chains if ch]
    return scans


FACTOR_SETS = [((2, 2), (2, 1), (1, 1)), ((2, 1), (1, 1), (1, 2)), ((1, 2), (2, 2), (1, 1)), ((3, 1), (1, 1), (1, 1)), ((2, 1), (2, 1), (2, 1)),
               ((1, 1), (2, 2), (2, 2)), ((2, 2), (1, 2), (2, 1)), ((1, 3), (1, 1), (1, 3)), ((4, 1), (2, 1), (1, 1)), ((2, 2), (2, 2), (1, 1))]
SEQ_SCRIPTS = [[(0,), (1, 2)], [(0,), (1,), (2,)], [(0, 1), (2,)], [(0, 2), (1,)]]


def main():
    seed, count = int(sys.argv[1]), int(sys.argv[2])
    share = float(sys.argv[sys.argv.index("--knob-share") + 1]) if "--knob-share" in sys.argv else 0.5
    rng = np.random.default_rng(seed)
    bad = refused = 0
    t0 = time.time()
    verbose = "--verbose" in sys.argv
    with_ref = "--ref" in sys.argv and O.have_ref()
    first = int(sys.argv[sys.argv.index("--from") + 1]) if "--from" in sys.argv else 0
    for i in range(count):
        w, h, kw, kind = draw(rng)
        r2 = np.random.default_rng(seed * 100003 + i)
        if kind == 0:
            img = O.synthetic_frame(max(w, 8), max(h, 8), 3000 + i + seed)[:h, :w].copy()
        elif kind == 1:
            img = r2.integers(0, 256, (h, w, 3), dtype=np.uint8)
        else:
            img = np.full((h, w, 3), 255, np.uint8)
            for _ in range(4):
                y, x = int(r2.integers(0, h)), int(r2.integers(0, w))
                img[y:y + 9, x:x + 9] = r2.integers(0, 64, 3, dtype=np.uint8)
        if kw.get("precision") == 12:      # the same picture with 12-bit samples (noise: the full range)
            img = r2.integers(0, 4096, img.shape).astype(np.uint16) if kind == 1 else (img.astype(np.uint16) << 4) | (img >> 4)
        if kw.get("gray") and r2.random() < 0.5:
            kw = dict(kw, grayin=True)
            img = img[:, :, 1].copy()
        env = {}
        if r2.random() < share:       # another plan of the same encode: same bytes expected
            if r2.random() < 0.5:
                env["MJH_TRELLIS_VARIANT"] = str(int(r2.choice([0, 2, 3, 4])))
            if r2.random() < 0.4:
                env["MJH_TRELLIS_V3"] = str(int(r2.choice([1, 2, 4, 8])))
            if r2.random() < 0.3:
                env["MJH_DENSE_CAP"] = str(int(r2.integers(0, 30)))
            if r2.random() < 0.3:
                env["MJH_DC_LATE"] = str(int(r2.integers(0, 3)))
            if r2.random() < 0.3:
                env["MJH_DC_SPEC"] = str(int(r2.integers(0, 2)))
        if i < first:
            continue
        if verbose:
            print("case", i, w, h, kind, kw, env, "%.0f s" % (time.time() - t0), flush=True)
        try:
            want = O.encode(O.make_params(w, h, **kw), img)
        except Exception as exc:      # the oracle refuses what the reference refuses
            refused += 1
            continue
        if with_ref:                  # (build container only: the oracle itself against the reference binary on the same case)
            try:
                rkw = {k: v for k, v in kw.items() if k != "grayin"}
                if O.ref_encode(img, **rkw)[0] != want:
                    bad += 1
                    print("ORACLE != REFERENCE:", seed, i, w, h, kw, flush=True)
                    continue
            except Exception as exc:
                print("reference run failed:", seed, i, kw, repr(exc)[:200], flush=True)
        try:
            os.environ.update(env)
            enc = M.Encoder(M.make_params(w, h, **kw), max_batch=3)
        except Exception as exc:
            print("REFUSED by the encoder but not by the oracle:", seed, i, w, h, kw, repr(exc)[:200], flush=True)
            bad += 1
            continue
        finally:
            for k in env:
                os.environ.pop(k, None)
        try:
            got = enc.encode_host(np.stack([img, img[::-1].copy(), img]))
            ok = got[0] == want and got[2] == want
        except Exception as exc:
            ok = False
            print("EXCEPTION", repr(exc)[:200], flush=True)
        enc.close()
        if not ok:
            bad += 1
            print("DIFFERENT:", seed, i, w, h, kw, env, flush=True)
    print("seed %d: %d cases, %d refused by the oracle, %d failures, %.0f s" % (seed, count, refused, bad, time.time() - t0), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
