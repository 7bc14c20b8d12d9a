#!/usr/bin/env python3
"""The TurboJPEG boundary on random calls, without a GPU: oracle/_ref/tjharness (the reference's unchanged libturbojpeg:
tjCompress2 / tjCompressFromYUV) three ways per case -- alone (the expected bytes), with the SHIPPED libjpeg interposing
library underneath libturbojpeg, and with the SHIPPED TurboJPEG-signature library (mozjpeg_amd/libmozjpeg_hip_turbojpeg.so) in
front of it -- the shipped libraries' `libmozjpeg_hip.so` being the kernel sources on the wave64 emulator (see fuzz_cjpeg.py).
Random sizes, every TJPF_ pixel format, every TJSAMP_ subsampling, qualities 1..100, bottom-up / progressive flags, planar
YUV input.  Development aid, correctness only; build container only.
usage: python tools/simt/fuzz_tj.py SEED COUNT [--verbose]"""
import os
import shutil
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import fuzz_cjpeg as F  # noqa: E402
O = F.O

TJH = os.path.join(F.REF, "tjharness")
BOTTOMUP, FASTDCT, ACCURATE, PROGRESSIVE = 2, 2048, 4096, 16384
# TJPF_: RGB BGR RGBX BGRX XBGR XRGB GRAY RGBA BGRA ABGR ARGB  (pixel size, offsets of R G B)
PF = [(3, 0, 1, 2), (3, 2, 1, 0), (4, 0, 1, 2), (4, 2, 1, 0), (4, 3, 2, 1), (4, 1, 2, 3), (1, 0, 0, 0), (4, 0, 1, 2), (4, 2, 1, 0), (4, 3, 2, 1), (4, 1, 2, 3)]
MCU = [(8, 8), (16, 8), (16, 16), (8, 8), (8, 16), (32, 8), (8, 32)]      # TJSAMP_ 444 422 420 GRAY 440 411 441


def main():
    seed, count = int(sys.argv[1]), int(sys.argv[2])
    verbose = "--verbose" in sys.argv
    d = F.dropin_dir()
    tjshim = os.path.join(d, "libmozjpeg_hip_turbojpeg.so")
    src_tj = os.path.join(F.PKG, "libmozjpeg_hip_turbojpeg.so")
    if not os.path.exists(tjshim) or os.path.getmtime(tjshim) < os.path.getmtime(src_tj):
        shutil.copy2(src_tj, tjshim)
    bad = refused = fast = 0
    t0 = time.time()
    for i in range(count):
        rng = np.random.default_rng(seed * 100003 + i)
        w = int(rng.integers(1, 200)); h = int(rng.integers(1, 160))
        if rng.random() < 0.1:
            w, h = (int(rng.integers(1, 1200)), int(rng.integers(1, 10))) if rng.random() < 0.5 else (int(rng.integers(1, 10)), int(rng.integers(1, 1200)))
        ss = int(rng.integers(0, 7))
        q = int(rng.choice([1, 5, 30, 50, 75, 80, 85, 90, 95, 96, 100])) if rng.random() < 0.7 else int(rng.integers(1, 101))
        flags = ACCURATE if rng.random() < 0.9 else (FASTDCT if rng.random() < 0.5 else 0)
        if np.random.default_rng(seed * 7919 + i).random() < 0.5:      # (its own generator: half of the calls as an application makes them -- no TJFLAG_ACCURATEDCT: JDCT_IFAST below quality 96)
            flags &= ~ACCURATE
        if rng.random() < 0.3:
            flags |= PROGRESSIVE
        kind = int(rng.integers(0, 3))
        if kind == 0:
            img = O.synthetic_frame(max(w, 8), max(h, 8), 9000 + i + seed)[:h, :w].copy()
        elif kind == 1:
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        else:
            img = np.full((h, w, 3), 255, np.uint8)
            img[h // 3:h // 3 + 9, w // 3:w // 3 + 9] = rng.integers(0, 64, 3, dtype=np.uint8)
        tmp = tempfile.mkdtemp(prefix="fzt_")
        try:
            raw = os.path.join(tmp, "in.raw")
            if rng.random() < 0.25:        # planar YUV (tjBufSizeYUV2 layout, align 1): planes padded to the MCU size
                pf = -1
                mw, mh = MCU[ss]
                pw = (w + mw // 8 - 1) // (mw // 8) * (mw // 8) if False else (w + (mw // 8) - 1) & ~((mw // 8) - 1)
                ph = (h + (mh // 8) - 1) & ~((mh // 8) - 1)
                planes = [rng.integers(0, 256, (ph, pw), dtype=np.uint8) if kind == 1 else np.resize(img[:, :, 1], (ph, pw))]
                if ss != 3:
                    cw, ch = pw * 8 // mw, ph * 8 // mh
                    planes += [rng.integers(0, 256, (ch, cw), dtype=np.uint8), rng.integers(0, 256, (ch, cw), dtype=np.uint8)]
                with open(raw, "wb") as f:
                    for p in planes:
                        f.write(np.ascontiguousarray(p).tobytes())
            else:
                pf = int(rng.integers(0, len(PF)))
                if rng.random() < 0.3:
                    flags |= BOTTOMUP
                ps, ro, go, bo = PF[pf]
                if ps == 1:
                    px = img[:, :, 1].copy()
                else:
                    px = rng.integers(0, 256, (h, w, ps), dtype=np.uint8)
                    px[..., ro], px[..., go], px[..., bo] = img[..., 0], img[..., 1], img[..., 2]
                px.tofile(raw)
            what = "%dx%d pf %d subsamp %d q %d flags %d" % (w, h, pf, ss, q, flags)
            if verbose:
                print("case", i, what, "%.0f s" % (time.time() - t0), flush=True)

            def cmd(out):
                return [TJH, str(w), str(h), str(pf), str(ss), str(q), str(flags), raw, out]
            ref = os.path.join(tmp, "ref.jpg")
            r0 = F.run(cmd(ref), {})
            if r0.returncode != 0:
                refused += 1
                if verbose:
                    print("   reference refuses:", r0.stderr.decode(errors="replace").strip()[-120:], flush=True)
                continue
            want = open(ref, "rb").read()
            msgs = []
            for name, pre in (("libjpeg shim under libturbojpeg", os.path.join(d, "libmozjpeg_hip_jpeg62.so")), ("TurboJPEG-signature library", tjshim)):
                out = os.path.join(tmp, "o.jpg")
                if os.path.exists(out):
                    os.remove(out)
                r = F.run(cmd(out), {}, preload=pre)
                fastdct = not (flags & ACCURATE) and q < 96 and (flags & FASTDCT or True)
                if r.returncode != 0 and fastdct and (b"JDCT_ISLOW" in r.stderr or b"fast" in r.stderr.lower()):
                    fast += 1       # the fast / default-fast DCT is outside the device path: refused with the reason (never emulated)
                    continue
                if r.returncode != 0:
                    msgs.append("%s: exit %d %s" % (name, r.returncode, r.stderr.decode(errors="replace").strip()[-300:]))
                elif open(out, "rb").read() != want:
                    msgs.append("%s: DIFFERENT (%d vs %d bytes)" % (name, os.path.getsize(out), len(want)))
            if msgs:
                bad += 1
                print("FAIL seed %d case %d: %s\n     %s" % (seed, i, what, "\n     ".join(msgs)), flush=True)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    print("seed %d: %d cases, %d refused by the reference itself, %d runs refused for the fast DCT, %d failures, %.0f s" % (seed, count, refused, fast, bad, time.time() - t0), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
