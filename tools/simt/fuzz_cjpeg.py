#!/usr/bin/env python3
"""The drop-in boundary on random command lines, without a GPU: the reference's UNCHANGED cjpeg / jpegtran binaries run three
times per case -- against the reference's own libjpeg (the expected bytes), with the SHIPPED interposing library
(mozjpeg_amd/libmozjpeg_hip_jpeg62.so) in front of it, and against the SHIPPED stand-alone libjpeg.so.62 -- where the shipped
libraries' DT_NEEDED `libmozjpeg_hip.so` is resolved (rpath $ORIGIN) to the kernel sources on the wave64 emulator
(tools/simt/_build/dropin/libmozjpeg_hip.so -> ../libmozjpeg_hip_simt.so).  What this exercises that fuzz_more.py does not:
the host C code between the libjpeg API and the C ABI (jpeg_shim.c capture_params / markers / destination managers,
jpeg_api.c's parameter setters and its readers of cjpeg's -qtables / -qslots / -sample / -scans / -icc state).
Development aid, correctness only; build container only (it needs oracle/_ref).
usage: python tools/simt/fuzz_cjpeg.py SEED COUNT [--verbose] [--from N] [--keep DIR]     one line per failure and a summary"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
import build_simt  # noqa: E402
import oracle_lib as O  # noqa: E402

REF = O.REF_DIR
CJPEG = os.path.join(REF, "cjpeg")
JPEGTRAN = os.path.join(REF, "jpegtran")
PKG = os.path.join(ROOT, "mozjpeg_amd")


def dropin_dir():
    """the shipped shim binaries next to a `libmozjpeg_hip.so` that is the emulator build"""
    simt = build_simt.build()
    d = os.path.join(os.path.dirname(simt), "dropin")
    os.makedirs(os.path.join(d, "standalone"), exist_ok=True)
    for src, dst in ((os.path.join(PKG, "libmozjpeg_hip_jpeg62.so"), os.path.join(d, "libmozjpeg_hip_jpeg62.so")),
                     (os.path.join(PKG, "standalone", "libjpeg.so.62"), os.path.join(d, "standalone", "libjpeg.so.62"))):
        if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
            shutil.copy2(src, dst)
    link = os.path.join(d, "libmozjpeg_hip.so")
    if not os.path.islink(link):
        os.symlink(os.path.join("..", os.path.basename(simt)), link)
    return d


KNOWN_REFUSALS = []

SAMPLE_SETS = ["1x1", "2x1", "1x2", "2x2", "4x1", "1x4", "4x2", "2x4", "2x2,2x1,1x1", "2x1,1x1,1x2", "1x2,2x2,1x1", "3x1,1x1,1x1",
               "2x1,2x1,2x1", "1x1,2x2,2x2", "2x2,1x2,2x1", "1x3,1x1,1x3", "4x1,2x1,1x1", "2x2,2x2,1x1", "2x2,1x1", "1x1,2x1"]


def script_text(scans):
    out = []
    for comps, ss, se, ah, al in scans:
        s = ",".join(str(c) for c in comps)
        if (ss, se, ah, al) != (0, 63, 0, 0):
            s += ": %d-%d, %d, %d" % (ss, se, ah, al)
        out.append(s + ";\n")
    return "".join(out)


def progressive_script(rng, ncomp):
    """(the generator of fuzz_more.py, restated here so that this tool does not load the emulator into its own process)"""
    first, later = [], []
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
    scans = first + [ac_first[int(j)] for j in rng.permutation(len(ac_first))]
    chains = [ch for ch in later if ch]
    while chains:
        j = int(rng.integers(0, len(chains)))
        scans.append(chains[j].pop(0))
        chains = [ch for ch in chains if ch]
    return scans


def write_image(rng, path_base, seed, i, twelve):
    """a small picture as cjpeg reads it: PPM / PGM (maxval 255, another maxval, or 12-bit), BMP or Targa"""
    w = int(rng.integers(1, 200)); h = int(rng.integers(1, 160))
    if rng.random() < 0.1:
        w, h = (int(rng.integers(1, 1500)), int(rng.integers(1, 12))) if rng.random() < 0.5 else (int(rng.integers(1, 12)), int(rng.integers(1, 1500)))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        img = O.synthetic_frame(max(w, 8), max(h, 8), 7000 + i + seed)[:h, :w].copy()
    elif kind == 1:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    else:
        img = np.full((h, w, 3), 255, np.uint8)
        for _ in range(4):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            img[y:y + 9, x:x + 9] = rng.integers(0, 64, 3, dtype=np.uint8)
    gray_in = rng.random() < 0.2
    fmt = "pnm"
    r = rng.random()
    if not twelve and r < 0.12:
        fmt = "bmp"
    elif not twelve and r < 0.24:
        fmt = "tga"
    if fmt == "pnm":
        if twelve:
            maxval = 4095
            data = (rng.integers(0, 4096, img.shape).astype(">u2") if kind == 1 else ((img.astype(np.uint16) << 4) | (img >> 4)).astype(">u2"))
        else:
            maxval = 255 if rng.random() < 0.8 else int(rng.choice([1, 15, 100, 254, 1023, 65535]))
            if maxval == 255:
                data = img
            elif maxval < 255:
                data = (img.astype(np.uint32) * maxval // 255).astype(np.uint8)
            else:
                data = (img.astype(np.uint32) * maxval // 255).astype(">u2")
        if gray_in:
            data = data[:, :, 1].copy()
        path = path_base + (".pgm" if gray_in else ".ppm")
        with open(path, "wb") as f:
            f.write(b"%s\n%d %d\n%d\n" % (b"P5" if gray_in else b"P6", w, h, maxval))
            f.write(data.tobytes())
        return path, w, h, gray_in, fmt
    if fmt == "bmp":       # 24-bit bottom-up BMP
        path = path_base + ".bmp"
        row = (w * 3 + 3) & ~3
        body = bytearray()
        for y in range(h - 1, -1, -1):
            line = img[y, :, ::-1].tobytes()
            body += line + b"\0" * (row - len(line))
        hdr = b"BM" + (54 + len(body)).to_bytes(4, "little") + b"\0\0\0\0" + (54).to_bytes(4, "little")
        hdr += (40).to_bytes(4, "little") + w.to_bytes(4, "little") + h.to_bytes(4, "little") + (1).to_bytes(2, "little") + (24).to_bytes(2, "little")
        hdr += (0).to_bytes(4, "little") + len(body).to_bytes(4, "little") + (2835).to_bytes(4, "little") * 2 + (0).to_bytes(4, "little") * 2
        with open(path, "wb") as f:
            f.write(hdr + bytes(body))
        return path, w, h, False, fmt
    path = path_base + ".tga"     # uncompressed true-colour (type 2) or gray (type 3) Targa, top-down
    hdr = bytearray(18)
    hdr[2] = 3 if gray_in else 2
    hdr[12:14] = w.to_bytes(2, "little"); hdr[14:16] = h.to_bytes(2, "little")
    hdr[16] = 8 if gray_in else 24
    hdr[17] = 0x20
    with open(path, "wb") as f:
        f.write(bytes(hdr) + (img[:, :, 1].tobytes() if gray_in else img[:, :, ::-1].tobytes()))
    return path, w, h, gray_in, fmt


def draw_cjpeg(rng, tmp, gray_in, fmt):
    """a random cjpeg command line (switches in an order cjpeg accepts: -revert first, it resets what came before)"""
    a = []
    twelve = False
    gray = gray_in or rng.random() < 0.15
    rgb = (not gray) and rng.random() < 0.08
    mode = int(rng.integers(0, 5))          # 0 baseline  1 fastcrush  2 revert  3 revert progressive  4 default
    revert = mode in (2, 3)
    if revert:
        a += ["-revert"]
        if rng.random() < 0.4:
            a += ["-optimize"]
    if rng.random() < 0.18:
        a += ["-arithmetic"]
    arith = "-arithmetic" in a
    if rng.random() < 0.2:
        a += ["-quant-table", str(int(rng.integers(0, 9)))]
    q = int(rng.choice([1, 5, 20, 40, 60, 75, 85, 90, 92, 95, 98, 100]))
    a += ["-quality", ("%d,%d" % (q, int(rng.choice([10, 50, 80, 100]))) if rng.random() < 0.15 else str(q))]
    if mode == 0:
        a += ["-baseline"]
    elif mode == 1:
        a += ["-fastcrush"]
    elif mode == 3:
        a += ["-progressive"]
    elif rng.random() < 0.1:
        a += ["-quant-baseline"]
    if gray and not gray_in:
        a += ["-grayscale"]
    if rgb:
        a += ["-rgb"]
    ncomp = 1 if gray else 3
    if not gray and rng.random() < 0.7:
        s = SAMPLE_SETS[int(rng.integers(0, len(SAMPLE_SETS)))]
        a += ["-sample", s]
    elif gray and rng.random() < 0.3:      # one component with factors of its own (cjpeg itself sets 2x1 at qualities 80..89)
        a += ["-sample", SAMPLE_SETS[int(rng.integers(0, 8))]]
    if rng.random() < 0.3:
        a += ["-restart", (str(int(rng.integers(1, 4))) if rng.random() < 0.5 else "%dB" % int(rng.integers(1, 40)))]
    if not revert:
        r = rng.random()
        if r < 0.12:
            a += ["-notrellis"]
        elif r < 0.2:
            a += ["-notrellis-dc"]
        elif r < 0.25:
            a += ["-trellis-dc"]
        if rng.random() < 0.2:
            a += [str(rng.choice(["-tune-psnr", "-tune-ssim", "-tune-ms-ssim", "-tune-hvs-psnr"]))]
        if rng.random() < 0.12:
            a += ["-lambda1", str(float(rng.choice([-2.0, 0.0, 8.5, 14.75, 20.0]))), "-lambda2", str(float(rng.choice([0.0, 8.0, 16.5, 22.0])))]
        if rng.random() < 0.15:
            a += ["-trellis-dc-ver-weight", str(float(rng.choice([0.25, 1.0, 3.0])))]
        if rng.random() < 0.2 and mode != 0:
            a += ["-dc-scan-opt", str(int(rng.integers(0, 3)))]
    if rng.random() < 0.12:
        a += ["-smooth", str(int(rng.integers(1, 101)))]
    if rng.random() < 0.15:
        a += ["-noovershoot"]
    if rng.random() < 0.1:
        a += ["-nojfif"]
    if rng.random() < 0.12:       # tables of the application's own (read_quant_tables rdswitch.c) and their slots
        path = os.path.join(tmp, "qt.txt")
        ntab = int(rng.integers(1, 4))
        with open(path, "w") as f:
            for t in range(ntab):
                hi = int(rng.choice([16, 255, 255, 2000]))
                f.write(" ".join(str(int(v)) for v in rng.integers(1, hi + 1, 64)) + "\n")
        a += ["-qtables", path]
        if not gray and rng.random() < 0.6:
            a += ["-qslots", ",".join(str(int(v)) for v in rng.integers(0, ntab, 3))]
    elif not gray and rng.random() < 0.06:
        a += ["-qslots", str(rng.choice(["0,1,1", "0,0,0", "0,1,0"]))]
    if rng.random() < 0.15 and "-baseline" not in a:
        path = os.path.join(tmp, "scans.txt")
        if mode in (2, 4) and rng.random() < 0.4 and ncomp == 3:
            groups = [[(0,), (1, 2)], [(0,), (1,), (2,)], [(0, 1), (2,)], [(0, 2), (1,)]][int(rng.integers(0, 4))]
            scans = [(g, 0, 63, 0, 0) for g in groups]
        else:
            scans = progressive_script(rng, ncomp)
        with open(path, "w") as f:
            f.write(script_text(scans))
        a += ["-scans", path]
    if rng.random() < 0.08:
        path = os.path.join(tmp, "profile.icc")
        with open(path, "wb") as f:
            f.write(rng.integers(0, 256, int(rng.choice([1, 300, 65519, 65520, 70000])), dtype=np.uint8).tobytes())
        a += ["-icc", path]
    if fmt == "tga":
        a += ["-targa"]
    if rng.random() < 0.15:
        a += ["-memdst"]
    return a, twelve


def run(cmd, env_extra, preload=None, libpath=None):
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD",)}
    env["LD_LIBRARY_PATH"] = libpath or REF
    if preload:
        env["LD_PRELOAD"] = preload
    env.setdefault("SIMT_STRICT", "1")
    env.update(env_extra)
    try:
        return subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    except subprocess.TimeoutExpired as exc:
        return subprocess.CompletedProcess(cmd, -999, b"", b"TIMEOUT " + (exc.stderr or b""))


def three_ways(cmd_of, d, tmp, verbose, standalone=True):
    """cmd_of(outfile) -> argv.  Returns a list of complaints.  (cjpeg -memdst leaves the file empty and reports the size of the
    memory destination's buffer on stderr: the reports are compared too.)"""
    ref, shim, alone = (os.path.join(tmp, n) for n in ("ref.jpg", "shim.jpg", "alone.jpg"))
    r0 = run(cmd_of(ref), {})
    if r0.returncode != 0 or not os.path.exists(ref):
        return None, r0           # the reference itself refuses the command line: nothing to compare
    want = open(ref, "rb").read()
    bad = []
    for name, out, kw in (("shim", shim, dict(preload=os.path.join(d, "libmozjpeg_hip_jpeg62.so"))),
                          ("stand-alone", alone, dict(libpath=os.path.join(d, "standalone")))):
        if name == "stand-alone" and not standalone:
            continue
        r = run(cmd_of(out), {}, **kw)
        if r.returncode != 0 and b"vertical sampling factor" in r.stderr and b"no CPU fallback" in r.stderr:
            KNOWN_REFUSALS.append(name)       # (until the end of round 5: one component, V > 1, trellis on; encoded now -- the count has to stay 0)
            continue
        if r.returncode == 0 and [l for l in r.stderr.splitlines() if l.startswith(b"Compressed size")] != [l for l in r0.stderr.splitlines() if l.startswith(b"Compressed size")]:
            bad.append("%s: -memdst reports %r, the reference %r" % (name, r.stderr[-60:], r0.stderr[-60:]))
        elif r.returncode != 0:
            bad.append("%s: exit %d %s" % (name, r.returncode, r.stderr.decode(errors="replace").strip()[-300:]))
        elif open(out, "rb").read() != want:
            got = open(out, "rb").read()
            k = next((j for j in range(min(len(got), len(want))) if got[j] != want[j]), min(len(got), len(want)))
            bad.append("%s: DIFFERENT (%d vs %d bytes, first difference at %d)" % (name, len(got), len(want), k))
    return bad, r0


def main():
    seed, count = int(sys.argv[1]), int(sys.argv[2])
    verbose = "--verbose" in sys.argv
    first = int(sys.argv[sys.argv.index("--from") + 1]) if "--from" in sys.argv else 0
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    d = dropin_dir()
    bad = ref_refused = 0
    t0 = time.time()
    for i in range(count):
        rng = np.random.default_rng(seed * 100003 + i)
        tmp = tempfile.mkdtemp(prefix="fzc_")
        try:
            transcode = rng.random() < 0.25
            twelve = (not transcode) and rng.random() < 0.06
            src, w, h, gray_in, fmt = write_image(rng, os.path.join(tmp, "in"), seed, i, twelve)
            a, _ = draw_cjpeg(rng, tmp, gray_in, fmt)
            if twelve:
                a = [x for x in a if x not in ("-trellis-dc",)] + ["-precision", "12", "-notrellis"]
            if i < first:
                continue
            if not transcode:
                # (a generator of its own for the DCT method: the cases of older seeds keep their other switches)
                dct = "fast" if np.random.default_rng(seed * 7919 + i).random() < 0.25 else "int"
                def cmd_of(out, a=a, src=src, dct=dct):
                    return [CJPEG, "-dct", dct] + a + ["-outfile", out, src]
                what = " ".join((["-dct", "fast"] if dct == "fast" else []) + a)
            else:
                # jpegtran: a reference-made file of the drawn settings, re-coded with random jpegtran switches
                mid = os.path.join(tmp, "mid.jpg")
                a = [x for x in a if x != "-memdst"]
                r = run([CJPEG, "-dct", "int"] + a + ["-outfile", mid, src], {})
                if r.returncode != 0:
                    ref_refused += 1
                    continue
                t = []
                if rng.random() < 0.3:
                    t += ["-revert"]
                if rng.random() < 0.3:
                    t += ["-optimize"]
                if rng.random() < 0.4:
                    t += ["-progressive"]
                if rng.random() < 0.2:
                    t += ["-arithmetic"]
                if rng.random() < 0.2:
                    t += ["-fastcrush"]
                if rng.random() < 0.25:
                    t += ["-restart", (str(int(rng.integers(1, 4))) if rng.random() < 0.5 else "%dB" % int(rng.integers(1, 40)))]
                if rng.random() < 0.3:
                    t += [str(rng.choice(["-flip horizontal", "-flip vertical", "-rotate 90", "-rotate 180", "-rotate 270", "-transpose", "-transverse"]))]
                    t = [y for x in t for y in x.split(" ")]
                    if rng.random() < 0.5:
                        t += [str(rng.choice(["-trim", "-perfect"]))]
                if rng.random() < 0.15:
                    t += ["-grayscale"]
                if rng.random() < 0.15:
                    t += ["-crop", "%dx%d+%d+%d" % (max(1, w // 2), max(1, h // 2), int(rng.integers(0, max(1, w // 2))), int(rng.integers(0, max(1, h // 2))))]
                t += ["-copy", str(rng.choice(["none", "all", "icc"]))]

                def cmd_of(out, t=t, mid=mid):
                    return [JPEGTRAN] + t + ["-outfile", out, mid]
                what = "jpegtran " + " ".join(t) + "   <- cjpeg " + " ".join(a)
            if verbose:
                print("case", i, w, h, fmt, what, "%.0f s" % (time.time() - t0), flush=True)
            res, r0 = three_ways(cmd_of, d, tmp, verbose, standalone=not transcode)    # (the stand-alone library is the compress API only: jpegtran decompresses)
            if res is None:
                ref_refused += 1
                if verbose:
                    print("   reference refuses:", r0.stderr.decode(errors="replace").strip()[-160:], flush=True)
                continue
            if res:
                bad += 1
                print("FAIL seed %d case %d (%dx%d %s): %s\n     %s" % (seed, i, w, h, fmt, what, "\n     ".join(res)), flush=True)
                if keep:
                    shutil.copytree(tmp, os.path.join(keep, "s%d_c%d" % (seed, i)), dirs_exist_ok=True)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    print("seed %d: %d cases, %d refused by the reference itself, %d runs refused for the one documented reason (one component, V > 1, trellis), %d failures, %.0f s"
          % (seed, count, ref_refused, len(KNOWN_REFUSALS), bad, time.time() - t0), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
