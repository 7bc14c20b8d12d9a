"""GPU (-m gpu): the libjpeg drop-in boundary.  An UNCHANGED cjpeg binary (built from the reference
sources, oracle/_ref/cjpeg, dynamically linked to the reference libjpeg.so.62) is run with
libmozjpeg_hip_jpeg62.so in front of it; its output files must equal the reference goldens."""
import hashlib
import os
import subprocess
import sys

import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "mozjpeg_amd", "libmozjpeg_hip_jpeg62.so")
CJPEG = os.path.join(O.REF_DIR, "cjpeg")
PPM = os.path.join(ROOT, "tests", "golden", "testorig.ppm")

needs = pytest.mark.skipif(not (os.path.exists(SHIM) and os.path.exists(CJPEG)),
                           reason="shim or reference cjpeg not built")

CJPEG_CASES = [
    ("revert", ["-revert", "-quality", "75", "-sample", "2x2"]),
    ("revert_opt", ["-revert", "-optimize", "-quality", "75", "-sample", "2x2"]),
    ("base", ["-quality", "75", "-baseline", "-sample", "2x2"]),
    ("base_notrellis", ["-quality", "75", "-baseline", "-notrellis", "-sample", "2x2"]),
    ("base_notrellis_dc", ["-quality", "75", "-baseline", "-notrellis-dc", "-sample", "2x2"]),
    ("base_noover", ["-quality", "75", "-baseline", "-noovershoot", "-sample", "2x2"]),
    ("base_q90_444", ["-quality", "90", "-baseline", "-sample", "1x1"]),
    ("base_422", ["-quality", "75", "-baseline", "-sample", "2x1"]),
    ("revert_440", ["-revert", "-quality", "75", "-sample", "1x2"]),
    ("revert_gray", ["-revert", "-quality", "75", "-grayscale"]),
    ("base_gray", ["-quality", "75", "-baseline", "-grayscale"]),
    ("base_restart1", ["-quality", "75", "-baseline", "-restart", "1", "-sample", "2x2"]),
    ("default_progressive", ["-quality", "75", "-sample", "2x2"]),          # progressive + scan search
    ("fastcrush", ["-quality", "75", "-fastcrush", "-sample", "2x2"]),
    ("q85_420_progressive", ["-quality", "85", "-sample", "2x2"]),
    ("revert_progressive", ["-revert", "-progressive", "-quality", "75", "-sample", "2x2"]),
    ("prog_search_restart1", ["-quality", "75", "-restart", "1", "-sample", "2x2"]),      # restart markers inside progressive scans
    ("fastcrush_restart2", ["-quality", "75", "-fastcrush", "-restart", "2", "-sample", "2x2"]),
    ("revert_prog_restart3b", ["-revert", "-progressive", "-quality", "75", "-restart", "3B", "-sample", "2x2"]),
    ("revert_opt_smooth1", ["-revert", "-optimize", "-quality", "75", "-smooth", "1", "-sample", "2x2"]),   # MD5_JPEG_420S_IFAST_OPT
    ("base_smooth30", ["-quality", "75", "-baseline", "-smooth", "30", "-sample", "2x2"]),
    ("dc_scan_opt1", ["-quality", "75", "-dc-scan-opt", "1", "-sample", "2x2"]),
    ("dc_scan_opt2", ["-quality", "75", "-dc-scan-opt", "2", "-sample", "2x2"]),
    ("fastcrush_dc_scan_opt2", ["-quality", "75", "-fastcrush", "-dc-scan-opt", "2", "-sample", "2x2"]),
    ("base_dc_ver_weight1", ["-quality", "75", "-baseline", "-trellis-dc-ver-weight", "1.0", "-sample", "2x2"]),
    ("q60_progressive_dc_ver_weight2", ["-quality", "60", "-trellis-dc-ver-weight", "2.0", "-sample", "2x2"]),
    # cjpeg -arithmetic: sequential, progressive with and without the scan search, restarts
    ("arith_revert", ["-revert", "-arithmetic", "-quality", "75", "-sample", "2x2"]),      # (-revert resets arith_code: it has to come first)
    ("arith_base", ["-arithmetic", "-quality", "75", "-baseline", "-sample", "2x2"]),
    ("arith_base_notrellis", ["-arithmetic", "-quality", "75", "-baseline", "-notrellis", "-sample", "2x2"]),
    ("arith_base_restart1", ["-arithmetic", "-quality", "75", "-baseline", "-restart", "1", "-sample", "2x2"]),
    ("arith_fastcrush", ["-arithmetic", "-quality", "75", "-fastcrush", "-sample", "2x2"]),
    ("arith_default_progressive", ["-arithmetic", "-quality", "75", "-sample", "2x2"]),
    ("arith_revert_progressive", ["-revert", "-arithmetic", "-progressive", "-quality", "75", "-sample", "2x2"]),
]


def run_cjpeg(args, out, env_extra=None):
    env = dict(os.environ)
    env["LD_PRELOAD"] = SHIM
    if env_extra:
        env.update(env_extra)
    return subprocess.run([CJPEG, "-dct", "int"] + args + ["-outfile", out, PPM], env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE)


@needs
@pytest.mark.parametrize("cname,args", CJPEG_CASES)
def test_unchanged_cjpeg_through_the_shim_matches_reference(cname, args, goldens, tmp_path):
    out = str(tmp_path / "o.jpg")
    r = run_cjpeg(args, out)
    assert r.returncode == 0, r.stderr.decode()
    data = open(out, "rb").read()
    g = goldens["testorig/%s" % cname]
    assert (len(data), hashlib.md5(data).hexdigest()) == (g["bytes"], g["md5"])


@needs
@pytest.mark.parametrize("args", [["-precision", "12", "-quality", "90", "-baseline", "-notrellis", "-sample", "1x1"],
                                  ["-precision", "12", "-quality", "75", "-notrellis"],
                                  ["-precision", "12", "-revert", "-quality", "90"]])
def test_unchanged_cjpeg_12bit_through_the_shim(args, tmp_path):
    """cjpeg -precision 12 rescales the 8-bit PPM to 12 bits (rdppm.c:844-848) and calls jpeg12_write_scanlines;
    reference = the same binary without the shim"""
    ref, gpu = str(tmp_path / "ref.jpg"), str(tmp_path / "gpu.jpg")
    r1 = run_cjpeg(args, gpu)
    env = dict(os.environ)
    r0 = subprocess.run([CJPEG, "-dct", "int"] + args + ["-outfile", ref, PPM], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r0.returncode == 0, r0.stderr.decode()
    assert r1.returncode == 0, r1.stderr.decode()
    assert open(gpu, "rb").read() == open(ref, "rb").read()


SCRIPT_SEQ = "0;\n1,2;\n"                                   # sequential: luma, then both chroma components interleaved
SCRIPT_SEQ_EACH = "0;\n1;\n2;\n"
SCRIPT_PROG = "0,1,2: 0-0, 0, 1;\n0: 1-63, 0, 1;\n1: 1-63, 0, 0;\n2: 1-63, 0, 0;\n0,1,2: 0-0, 1, 0;\n0: 1-63, 1, 0;\n"


@needs
@pytest.mark.parametrize("script,args", [(SCRIPT_SEQ, ["-quality", "75", "-sample", "2x2"]),
                                         (SCRIPT_SEQ_EACH, ["-quality", "75", "-restart", "1", "-sample", "2x2"]),
                                         (SCRIPT_SEQ, ["-revert", "-quality", "75", "-sample", "2x1"]),
                                         (SCRIPT_PROG, ["-quality", "75", "-sample", "2x2"]),
                                         (None, ["-quality", "75", "-baseline", "-sample", "2x2,2x1,1x1"]),
                                         (None, ["-quality", "75", "-sample", "2x1,1x1,1x2"]),
                                         (None, ["-revert", "-quality", "75", "-sample", "1x2,2x2,1x1"])])
def test_unchanged_cjpeg_scan_scripts_and_sampling_factors_through_the_shim(script, args, tmp_path):
    """cjpeg -scans FILE (read_scan_script rdswitch.c: sequential files of several whole-block scans, a progressive script of the
    application's own) and cjpeg -sample HxV,HxV,HxV (chroma other than 1x1, luma smaller than chroma) -- configurations the shim
    refused until round 5.  Reference = the same binary without the shim."""
    ref, gpu = str(tmp_path / "ref.jpg"), str(tmp_path / "gpu.jpg")
    if script is not None:
        f = str(tmp_path / "script.txt")
        with open(f, "w") as fh:
            fh.write(script)
        args = args + ["-scans", f]
    r1 = run_cjpeg(args, gpu)
    r0 = subprocess.run([CJPEG, "-dct", "int"] + args + ["-outfile", ref, PPM], env=dict(os.environ), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r0.returncode == 0, r0.stderr.decode()
    assert r1.returncode == 0, r1.stderr.decode()
    assert open(gpu, "rb").read() == open(ref, "rb").read()


@needs
def test_unsupported_configuration_is_an_error_without_fallback(tmp_path):
    out = str(tmp_path / "o.jpg")
    r = run_cjpeg(["-quality", "75", "-dct", "float"], out)        # the float DCT is outside the GPU path
    assert r.returncode != 0
    assert b"no CPU fallback" in r.stderr


@needs
def test_explicit_passthrough_is_logged(goldens, tmp_path):
    out = str(tmp_path / "o.jpg")
    r = run_cjpeg(["-quality", "75", "-dct", "float"], out, {"MOZJPEG_HIP_PASSTHROUGH": "1"})   # the float DCT: outside the GPU path
    assert r.returncode == 0, r.stderr.decode()
    assert b"handing over to the host libjpeg" in r.stderr
    assert open(out, "rb").read()[:2] == b"\xff\xd8"


# ---- the stand-alone library: mozjpeg_amd/standalone/libjpeg.so.62 replaces the reference's libjpeg
# for the compress API; the unchanged cjpeg binary finds it through LD_LIBRARY_PATH and nothing of the reference's
# library is in the process -------------------------------------------------------------------------------------------
STANDALONE_DIR = os.path.join(ROOT, "mozjpeg_amd", "standalone")
needs_sa = pytest.mark.skipif(not (os.path.exists(os.path.join(STANDALONE_DIR, "libjpeg.so.62")) and os.path.exists(CJPEG)),
                              reason="stand-alone library or reference cjpeg not built")


def run_cjpeg_standalone(args, out, inp=PPM):
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    env["LD_LIBRARY_PATH"] = STANDALONE_DIR
    # prove which library the binary gets: the loader's own trace names every object it maps
    env["LD_DEBUG"] = "libs"
    r = subprocess.run([CJPEG, "-dct", "int"] + args + ["-outfile", out, inp], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    loaded = [ln for ln in r.stderr.decode(errors="replace").splitlines() if "calling init:" in ln]
    assert any(STANDALONE_DIR in ln for ln in loaded), loaded
    assert not any(O.REF_DIR in ln and "libjpeg" in ln for ln in loaded), "the reference's libjpeg was loaded"
    return r


@needs_sa
@pytest.mark.parametrize("cname,args", CJPEG_CASES)
def test_unchanged_cjpeg_against_the_standalone_library(cname, args, goldens, tmp_path):
    out = str(tmp_path / "o.jpg")
    r = run_cjpeg_standalone(args, out)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    data = open(out, "rb").read()
    g = goldens["testorig/%s" % cname]
    assert (len(data), hashlib.md5(data).hexdigest()) == (g["bytes"], g["md5"])


# ---- cjpeg's other readers and one-component sampling (found by tools/simt/fuzz_cjpeg.py, round 5): the BMP / bottom-up Targa
# readers keep the picture in a virtual array that jpeg_start_compress has to realise (rdbmp.c:605-609, jcinit.c:143); a gray image
# at quality 80..89 gets component 0 sampled 2x1 (set_quality_ratings rdswitch.c:566-570).  The same cases run on the emulator in
# tests/test_simt_dropin.py; expected bytes = the same binary on the reference's own library ----------------------------------
from test_simt_dropin import CASES as READER_CASES, write_bmp, write_tga  # noqa: E402


@needs
@needs_sa
@pytest.mark.parametrize("kind,args", READER_CASES)
def test_unchanged_cjpeg_readers_and_one_component_sampling(kind, args, tmp_path, fixture_images):
    img = fixture_images["testorig"]
    src = str(tmp_path / ("in." + kind.split("_")[0]))
    if kind == "bmp":
        write_bmp(src, img)
    elif kind.startswith("tga"):
        write_tga(src, img, kind.endswith("bottom_up"))
    elif kind == "pgm":
        with open(src, "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]) + img[:, :, 1].tobytes())
    else:
        src = PPM
    ref, gpu, alone = (str(tmp_path / n) for n in ("ref.jpg", "gpu.jpg", "alone.jpg"))
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    env["LD_LIBRARY_PATH"] = O.REF_DIR
    r0 = subprocess.run([CJPEG, "-dct", "int"] + args + ["-outfile", ref, src], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r0.returncode == 0, r0.stderr.decode()
    env["LD_PRELOAD"] = SHIM
    r1 = subprocess.run([CJPEG, "-dct", "int"] + args + ["-outfile", gpu, src], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r1.returncode == 0, r1.stderr.decode()
    assert open(gpu, "rb").read() == open(ref, "rb").read()
    r2 = run_cjpeg_standalone(args, alone, inp=src)
    assert r2.returncode == 0, r2.stderr.decode()[-2000:]
    assert open(alone, "rb").read() == open(ref, "rb").read()


ICC = os.path.join(ROOT, "tests", "golden", "test1.icc")


@needs
@pytest.mark.parametrize("standalone", [False, True])
def test_rgb_output_with_icc_profile_reproduces_the_references_pinned_md5(standalone, tmp_path):
    """cjpeg -revert -rgb -dct int -icc test1.icc testorig.ppm = the reference's own bit test `rgb-islow`
    (CMakeLists.txt:1347,1427: MD5_JPEG_RGB_ISLOW): JCS_RGB output (null_convert), Adobe APP14, ICC APP2 segments written by
    the application between jpeg_start_compress and the first scanline"""
    out = str(tmp_path / "o.jpg")
    args = ["-revert", "-rgb", "-icc", ICC]
    r = run_cjpeg_standalone(args, out) if standalone else run_cjpeg(args, out)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == "1d44a406f61da743b5fd31c0a9abdca3"


@needs
@pytest.mark.parametrize("standalone", [False, True])
@pytest.mark.parametrize("name,args,md5", [
    ("422-ifast-opt", ["-revert", "-sample", "2x1", "-dct", "fast", "-opt"], "2540287b79d913f91665e660303ab2c8"),
    ("420-q100-ifast-prog", ["-revert", "-sample", "2x2", "-quality", "100", "-dct", "fast", "-scans", "TEST_SCAN"], "0ba15f9dab81a703505f835f9dbbac6d"),
    ("3x2-ifast-prog", ["-revert", "-sample", "3x2", "-dct", "fast", "-prog"], "1ee5d2c1a77f2da495f993c8c7cceca5")])
def test_fast_dct_reproduces_the_references_pinned_md5s(name, args, md5, standalone, tmp_path):
    """the reference's own bit tests of `cjpeg -dct fast` on testorig.ppm (CMakeLists.txt:1459, :1498, :1561: MD5_JPEG_422_IFAST_OPT,
    MD5_JPEG_420_IFAST_Q100_PROG with testimages/test.scan, MD5_JPEG_3x2_IFAST_PROG) through the unchanged cjpeg on the GPU path"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from cases import TEST_SCAN
    scan = str(tmp_path / "test.scan")
    with open(scan, "w") as f:
        for comps, ss, se, ah, al in TEST_SCAN:
            f.write("%s: %d %d %d %d;\n" % (" ".join(str(c) for c in comps), ss, se, ah, al))
    args = [scan if a == "TEST_SCAN" else a for a in args]
    out = str(tmp_path / "o.jpg")
    r = run_cjpeg_standalone(args, out) if standalone else run_cjpeg(args, out)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == md5


HARNESS = os.path.join(ROOT, "tests", "native", "shim_harness")
needs_h = pytest.mark.skipif(not (os.path.exists(HARNESS) and os.path.exists(SHIM)), reason="tests/native/shim_harness not built")


@needs_h
@pytest.mark.parametrize("mode", ["preload", "standalone"])
@pytest.mark.parametrize("scenario", ["interleaved", "abort_reuse", "abort_midway", "markers", "stdio", "ext_params", "color_spaces", "custom_huffman", "abbreviated"])
def test_libjpeg_client_scenarios(scenario, mode):
    """two interleaved compress objects on one thread; error_exit longjmp -> jpeg_abort_compress -> reuse, and hundreds
    of start/abort and create/destroy cycles without growth; COM / APPn / ICC markers and JFIF density fields; the stdio
    destination; Huffman tables of the application's own with optimize_coding off; abbreviated datastreams (jpeg_write_tables,
    jpeg_suppress_tables, write_all_tables FALSE over several frames of one object).  Expected output = the same binary on the
    reference's libjpeg."""
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    env["LD_LIBRARY_PATH"] = O.REF_DIR
    want = subprocess.run([HARNESS, scenario], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert want.returncode == 0, want.stderr.decode()
    if mode == "preload":
        env["LD_PRELOAD"] = SHIM
    else:
        env["LD_LIBRARY_PATH"] = STANDALONE_DIR
    got = subprocess.run([HARNESS, scenario], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert got.returncode == 0, got.stderr.decode()[-2000:]
    assert got.stdout == want.stdout, (got.stdout, want.stdout, got.stderr[-500:])
    import json
    pinned = json.load(open(os.path.join(ROOT, "tests", "golden", "goldens_calls.json"))).get("shim_harness " + scenario)      # (reference-made: make_goldens.py --calls)
    assert pinned is None or got.stdout.decode() == pinned


@needs
def test_device_selection_by_environment(goldens, tmp_path):
    """MOZJPEG_HIP_DEVICE pins the drop-in to a GPU (taken modulo the device count, so any value works on a 1-GPU box)"""
    out = str(tmp_path / "o.jpg")
    r = run_cjpeg(["-quality", "75", "-baseline", "-sample", "2x2"], out, {"MOZJPEG_HIP_DEVICE": "5"})
    assert r.returncode == 0, r.stderr.decode()
    g = goldens["testorig/base"]
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == g["md5"]


# ---- TurboJPEG boundary: the reference's own tjCompress2 (turbojpeg.c:1169, unchanged, built into
# oracle/_ref/libturbojpeg.so.0) with the libjpeg drop-in in front of it --------------------------
TJH = os.path.join(O.REF_DIR, "tjharness")
needs_tj = pytest.mark.skipif(not (os.path.exists(SHIM) and os.path.exists(TJH)), reason="shim or tjharness not built")
TJPF = {"RGB": (0, [0, 1, 2], 3), "BGR": (1, [2, 1, 0], 3), "RGBX": (2, [0, 1, 2], 4), "BGRX": (3, [2, 1, 0], 4),
        "XBGR": (4, [3, 2, 1], 4), "XRGB": (5, [1, 2, 3], 4)}
TJSAMP = {"444": 0, "422": 1, "420": 2, "GRAY": 3, "440": 4, "411": 5, "441": 6}
ACCURATE, BOTTOMUP, PROGRESSIVE = 4096, 2, 16384


TJSHIM = os.path.join(ROOT, "mozjpeg_amd", "libmozjpeg_hip_turbojpeg.so")
# (subsampling, quality, flags) of a legacy tjCompress2 call.  Without TJFLAG_ACCURATEDCT and below quality 96 the call selects
# JDCT_FASTEST = JDCT_IFAST (processFlags turbojpeg.c:522-527) -- its most common form: every set is run both ways
_TJ_ACCURATE_CALLS = [("420", 75, ACCURATE), ("444", 96, 0), ("422", 80, ACCURATE | BOTTOMUP), ("GRAY", 75, ACCURATE), ("440", 60, ACCURATE),
                      ("420", 85, ACCURATE | PROGRESSIVE), ("411", 75, ACCURATE), ("441", 90, ACCURATE | PROGRESSIVE)]
TJ_DEFAULT_CALLS = [(ss, q if q < 96 else 95, f & ~ACCURATE) for ss, q, f in _TJ_ACCURATE_CALLS]
TJ_CALLS = _TJ_ACCURATE_CALLS + TJ_DEFAULT_CALLS


def tj_run(raw, w, h, pf, ss, q, flags, out, preload):
    """preload: False = the reference alone; True = the libjpeg drop-in in front of the reference's libjpeg (underneath its
    unchanged libturbojpeg); "tj" = the TurboJPEG-signature library in front of libturbojpeg itself (what a STOCK
    libturbojpeg with its private libjpeg needs)"""
    env = dict(os.environ)
    if preload == "tj":
        env["LD_PRELOAD"] = TJSHIM
    elif preload:
        env["LD_PRELOAD"] = SHIM
    return subprocess.run([TJH, str(w), str(h), str(pf), str(ss), str(q), str(flags), raw, out], env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE)


@needs_tj
@pytest.mark.parametrize("pfname", list(TJPF))
@pytest.mark.parametrize("ssname,q,flags", TJ_CALLS)
def test_unchanged_tjcompress2_through_the_shim(pfname, ssname, q, flags, tmp_path):
    import numpy as np
    rgb = O.read_ppm(PPM)
    h, w = rgb.shape[:2]
    pf, offs, ps = TJPF[pfname]
    px = np.full((h, w, ps), 0x5A, np.uint8)
    for ch in range(3):
        px[..., offs[ch]] = rgb[..., ch]
    raw = str(tmp_path / "in.raw")
    px.tofile(raw)
    ref, gpu = str(tmp_path / "ref.jpg"), str(tmp_path / "gpu.jpg")
    r0 = tj_run(raw, w, h, pf, TJSAMP[ssname], q, flags, ref, preload=False)
    r1 = tj_run(raw, w, h, pf, TJSAMP[ssname], q, flags, gpu, preload=True)
    assert r0.returncode == 0, r0.stderr.decode()
    assert r1.returncode == 0, r1.stderr.decode()
    assert open(gpu, "rb").read() == open(ref, "rb").read()


@needs_tj
@pytest.mark.parametrize("pfname", ["RGB", "BGRX", "XRGB"])
@pytest.mark.parametrize("ssname,q,flags", TJ_CALLS)
def test_turbojpeg_signature_exports_match_the_reference_turbojpeg(pfname, ssname, q, flags, tmp_path):
    """libmozjpeg_hip_turbojpeg.so: tjInitCompress / tjCompress2 / tjDestroy served directly by the batch encoder"""
    if not os.path.exists(TJSHIM):
        pytest.skip("TurboJPEG-signature library not built")
    import numpy as np
    rgb = O.read_ppm(PPM)
    h, w = rgb.shape[:2]
    pf, offs, ps = TJPF[pfname]
    px = np.full((h, w, ps), 0x5A, np.uint8)
    for ch in range(3):
        px[..., offs[ch]] = rgb[..., ch]
    raw = str(tmp_path / "in.raw")
    px.tofile(raw)
    ref, gpu = str(tmp_path / "ref.jpg"), str(tmp_path / "gpu.jpg")
    r0 = tj_run(raw, w, h, pf, TJSAMP[ssname], q, flags, ref, preload=False)
    r1 = tj_run(raw, w, h, pf, TJSAMP[ssname], q, flags, gpu, preload="tj")
    assert r0.returncode == 0, r0.stderr.decode()
    assert r1.returncode == 0, r1.stderr.decode()
    assert open(gpu, "rb").read() == open(ref, "rb").read()


@needs_tj
@pytest.mark.parametrize("w,h,ssname,q,flags", [(227, 149, "420", 75, ACCURATE), (64, 48, "444", 96, 0), (50, 33, "GRAY", 75, ACCURATE),
                                                (229, 151, "411", 75, ACCURATE), (227, 149, "420", 85, ACCURATE | PROGRESSIVE),
                                                (227, 149, "420", 75, 0), (101, 77, "422", 50, 0), (229, 151, "411", 85, PROGRESSIVE)])
def test_turbojpeg_signature_yuv_exports_match_the_reference_turbojpeg(w, h, ssname, q, flags, tmp_path):
    if not os.path.exists(TJSHIM):
        pytest.skip("TurboJPEG-signature library not built")
    samp = {"444": (1, 1), "422": (2, 1), "420": (2, 2), "GRAY": (1, 1), "440": (1, 2), "411": (4, 1), "441": (1, 4)}[ssname]
    po = O.make_params(w, h, revert=True, quality=q, sample=samp, gray=(ssname == "GRAY"), progressive=bool(flags & PROGRESSIVE))
    raw = str(tmp_path / "in.yuv")
    with open(raw, "wb") as f:
        for a in O.synthetic_planes(po, 11):
            f.write(a.tobytes())
    ref, gpu = str(tmp_path / "ref.jpg"), str(tmp_path / "gpu.jpg")
    r0 = tj_run(raw, w, h, -1, TJSAMP[ssname], q, flags, ref, preload=False)
    r1 = tj_run(raw, w, h, -1, TJSAMP[ssname], q, flags, gpu, preload="tj")
    assert r0.returncode == 0, r0.stderr.decode()
    assert r1.returncode == 0, r1.stderr.decode()
    assert open(gpu, "rb").read() == open(ref, "rb").read()


@needs_tj
@pytest.mark.parametrize("preload", [True, "tj"])
@pytest.mark.parametrize("ssname,q,flags", TJ_DEFAULT_CALLS)
def test_tjcompress2_default_flags_select_the_fast_dct_bit_exactly(ssname, q, flags, preload, tmp_path):
    """tjCompress2(..., flags without TJFLAG_ACCURATEDCT) below quality 96 = JDCT_IFAST (turbojpeg.c:522-527, jfdctfst.c): the
    reference TurboJPEG's bytes, through the libjpeg drop-in underneath it and through the TurboJPEG-signature library"""
    rgb = O.read_ppm(PPM)
    h, w = rgb.shape[:2]
    raw = str(tmp_path / "in.raw")
    rgb.tofile(raw)
    ref, gpu = str(tmp_path / "ref.jpg"), str(tmp_path / "gpu.jpg")
    r0 = tj_run(raw, w, h, 0, TJSAMP[ssname], q, flags, ref, preload=False)
    r1 = tj_run(raw, w, h, 0, TJSAMP[ssname], q, flags, gpu, preload=preload)
    assert r0.returncode == 0, r0.stderr.decode()
    assert r1.returncode == 0, r1.stderr.decode()
    assert open(gpu, "rb").read() == open(ref, "rb").read()


@needs_tj
@pytest.mark.parametrize("w,h,ssname,q,flags", [(227, 149, "420", 75, ACCURATE), (64, 48, "444", 96, 0), (131, 77, "422", 80, ACCURATE),
                                                (50, 33, "GRAY", 75, ACCURATE), (101, 77, "440", 60, ACCURATE),
                                                (227, 149, "420", 85, ACCURATE | PROGRESSIVE), (1920, 1080, "420", 75, ACCURATE),
                                                (229, 151, "411", 75, ACCURATE), (99, 203, "441", 80, ACCURATE)])
def test_unchanged_tjcompressfromyuv_through_the_shim(w, h, ssname, q, flags, tmp_path):
    """the reference's tjCompressFromYUV -> tj3CompressFromYUVPlanes8 (turbojpeg.c:1222) ->
    jpeg_write_raw_data (jcapistd.c:145) chain, unchanged, lands in the GPU plane-input path; the file equals
    the one the reference library writes for the same planar YUV image (and the oracle's)."""
    samp = {"444": (1, 1), "422": (2, 1), "420": (2, 2), "GRAY": (1, 1), "440": (1, 2), "411": (4, 1), "441": (1, 4)}[ssname]
    kw = dict(revert=True, quality=q, sample=samp, gray=(ssname == "GRAY"), progressive=bool(flags & PROGRESSIVE))
    po = O.make_params(w, h, **kw)
    planes = O.synthetic_planes(po, 11)
    raw = str(tmp_path / "in.yuv")
    with open(raw, "wb") as f:
        for a in planes:
            f.write(a.tobytes())
    ref, gpu = str(tmp_path / "ref.jpg"), str(tmp_path / "gpu.jpg")
    r0 = tj_run(raw, w, h, -1, TJSAMP[ssname], q, flags, ref, preload=False)
    r1 = tj_run(raw, w, h, -1, TJSAMP[ssname], q, flags, gpu, preload=True)
    assert r0.returncode == 0, r0.stderr.decode()
    assert r1.returncode == 0, r1.stderr.decode()
    data = open(gpu, "rb").read()
    assert data == open(ref, "rb").read()
    assert data == O.encode_planes(po, planes)


# ---- jpegtran boundary: the reference's own jpegtran binary, unchanged, with the drop-in in
# front: jpeg_write_coefficients (jctrans.c:44) -> GPU entropy-coding passes ---------------------------------------
JPEGTRAN = os.path.join(O.REF_DIR, "jpegtran")
needs_jt = pytest.mark.skipif(not (os.path.exists(SHIM) and os.path.exists(JPEGTRAN)), reason="shim or jpegtran not built")


@needs_jt
@pytest.mark.parametrize("src_kw", [dict(baseline=True), dict(revert=True, sample=(2, 1)), dict(baseline=True, gray=True)])
@pytest.mark.parametrize("switches", [["-progressive"], ["-revert"], ["-revert", "-optimize"], ["-fastcrush", "-progressive"],
                                      ["-revert", "-restart", "2"], ["-progressive", "-restart", "1"], ["-progressive", "-rotate", "90"],
                                      ["-revert", "-optimize", "-flip", "horizontal", "-trim"]])
def test_unchanged_jpegtran_through_the_shim(src_kw, switches):
    """same bytes as the reference jpegtran, including after lossless transforms (which rewrite the coefficient arrays
    between jpeg_write_coefficients and jpeg_finish_compress)"""
    img = O.read_ppm(PPM)
    h, w = img.shape[:2]
    src = O.encode(O.make_params(w, h, **src_kw), img)
    ref = O.ref_jpegtran(src, switches)
    gpu = O.ref_jpegtran(src, switches, preload=SHIM)
    assert gpu == ref
