"""CPU suite: the drop-in boundary without a GPU.  The reference's UNCHANGED cjpeg / jpegtran run with the SHIPPED interposing
library (mozjpeg_amd/libmozjpeg_hip_jpeg62.so) in front of the reference's libjpeg, and against the SHIPPED stand-alone
libjpeg.so.62, while those libraries' `libmozjpeg_hip.so` is the kernel sources on the wave64 emulator (tools/simt: test
infrastructure like oracle/; see tests/test_simt_kernels.py for what that is and is not).  What is under test is the host C
code between the libjpeg API and the C ABI -- jpeg_shim.c, jpeg_api.c -- on command lines the fixed lists of
tests/test_gpu_dropin.py do not hold; expected bytes = the same binary on the reference's own library.
Build container only (needs oracle/_ref); the same cases run on the chip in tests/test_gpu_dropin.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "simt"))
CJPEG = os.path.join(O.REF_DIR, "cjpeg")
SHIM = os.path.join(ROOT, "mozjpeg_amd", "libmozjpeg_hip_jpeg62.so")
STANDALONE = os.path.join(ROOT, "mozjpeg_amd", "standalone", "libjpeg.so.62")

pytestmark = pytest.mark.skipif(not (os.path.exists(CJPEG) and os.path.exists(SHIM) and os.path.exists(STANDALONE)),
                                reason="reference binaries (oracle/_ref) or the shim libraries are not built")


@pytest.fixture(scope="module")
def fz():
    import fuzz_cjpeg
    return fuzz_cjpeg, fuzz_cjpeg.dropin_dir()


def write_bmp(path, img):
    """24-bit bottom-up BMP: cjpeg's reader keeps the whole picture in a virtual array it REQUESTS before jpeg_start_compress
    and that jpeg_start_compress has to realise (rdbmp.c:605-609, jcinit.c:143)"""
    h, w = img.shape[:2]
    row = (w * 3 + 3) & ~3
    body = b"".join(img[y, :, ::-1].tobytes() + b"\0" * (row - 3 * w) for y in range(h - 1, -1, -1))
    le = lambda v, n: int(v).to_bytes(n, "little")   # noqa: E731
    hdr = b"BM" + le(54 + len(body), 4) + le(0, 4) + le(54, 4) + le(40, 4) + le(w, 4) + le(h, 4) + le(1, 2) + le(24, 2) + le(0, 4) + le(len(body), 4) + le(2835, 4) * 2 + le(0, 4) * 2
    with open(path, "wb") as f:
        f.write(hdr + body)


def write_tga(path, img, bottom_up):
    h, w = img.shape[:2]
    hdr = bytearray(18)
    hdr[2] = 2
    hdr[12:14] = w.to_bytes(2, "little"); hdr[14:16] = h.to_bytes(2, "little")
    hdr[16] = 24
    hdr[17] = 0 if bottom_up else 0x20
    with open(path, "wb") as f:
        f.write(bytes(hdr) + (img[::-1] if bottom_up else img)[:, :, ::-1].tobytes())


CASES = [
    ("bmp", ["-quality", "75", "-baseline"]),                       # the reader's virtual array
    ("bmp", ["-quality", "75"]),
    ("tga_bottom_up", ["-quality", "75", "-baseline", "-targa"]),   # the same in rdtarga.c
    ("tga", ["-revert", "-quality", "75", "-targa"]),
    ("pgm", ["-quality", "85"]),                                    # a gray image at quality 80..89: component 0 sampled 2x1 (rdswitch.c:566-570)
    ("ppm", ["-quality", "85", "-grayscale", "-baseline"]),
    ("ppm", ["-quality", "88", "-grayscale", "-arithmetic", "-restart", "1"]),
    ("pgm", ["-revert", "-quality", "75", "-sample", "2x2"]),       # V > 1 on one component
    ("pgm", ["-quality", "75", "-baseline", "-sample", "2x2"]),     # ... with the trellis: its passes walk iMCU rows of V block rows (jccoefct.c:418-441)
    ("ppm", ["-quality", "75", "-grayscale", "-sample", "1x2", "-trellis-dc-ver-weight", "2.0"]),
    ("pgm", ["-quality", "75", "-baseline", "-sample", "4x1", "-smooth", "20"]),
]


@pytest.mark.parametrize("kind,args", CASES)
def test_unchanged_cjpeg_on_the_emulator_readers_and_one_component_sampling(fz, kind, args, tmp_path, fixture_images):
    F, d = fz
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
        src = os.path.join(ROOT, "tests", "golden", "testorig.ppm")
    bad, r0 = F.three_ways(lambda out: [CJPEG, "-dct", "int"] + args + ["-outfile", out, src], d, str(tmp_path), False)
    assert bad is not None, r0.stderr.decode()
    assert bad == []


HARNESS = os.path.join(ROOT, "tests", "native", "shim_harness")


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="tests/native/shim_harness not built")
@pytest.mark.parametrize("scenario", ["interleaved", "abort_reuse", "abort_midway", "markers", "stdio", "ext_params", "color_spaces", "custom_huffman", "abbreviated"])
def test_libjpeg_client_scenarios_on_the_emulator(fz, scenario):
    """tests/native/shim_harness.c (two objects interleaved, abort + reuse, markers, stdio destination, the extension parameters,
    in_color_space = JCS_YCbCr / an extended pixel order / one component with the application's sampling factors, Huffman tables of
    the application's own with optimize_coding off, abbreviated datastreams over several frames of one object) against the
    reference's library, the shipped shim in front of it, and the shipped stand-alone library"""
    F, d = fz
    want = F.run([HARNESS, scenario], {})
    assert want.returncode == 0, want.stderr.decode()
    for kw in (dict(preload=os.path.join(d, "libmozjpeg_hip_jpeg62.so")), dict(libpath=os.path.join(d, "standalone"))):
        got = F.run([HARNESS, scenario], {}, **kw)
        assert got.returncode == 0, got.stderr.decode()[-2000:]
        assert got.stdout == want.stdout, (kw, got.stdout, want.stdout)


MT_BENCH = os.path.join(ROOT, "tests", "native", "mt_bench")


@pytest.mark.skipif(not os.path.exists(MT_BENCH), reason="tests/native/mt_bench not built")
@pytest.mark.parametrize("cfg", [["8", "6", "97", "61", "75", "baseline"], ["6", "4", "200", "130", "85"], ["16", "3", "64", "64", "60", "baseline"]])
def test_many_client_threads_through_the_batcher_on_the_emulator(fz, cfg):
    """tests/native/mt_bench.c: T threads, each with its own compress object, N images each -- through the shim's cross-thread
    batcher (and with MOZJPEG_HIP_BATCH=0) and through the stand-alone library; every thread's first file and the byte total equal
    the reference's"""
    import json
    F, d = fz
    want = F.run([MT_BENCH] + cfg, {})
    assert want.returncode == 0, want.stderr.decode()
    w = json.loads(want.stdout.decode().strip().splitlines()[-1])
    for kw, env in ((dict(preload=os.path.join(d, "libmozjpeg_hip_jpeg62.so")), {}), (dict(preload=os.path.join(d, "libmozjpeg_hip_jpeg62.so")), {"MOZJPEG_HIP_BATCH": "0"}),
                    (dict(libpath=os.path.join(d, "standalone")), {})):
        got = F.run([MT_BENCH] + cfg, env, **kw)
        assert got.returncode == 0, got.stderr.decode()[-1500:]
        g = json.loads(got.stdout.decode().strip().splitlines()[-1])
        assert (g["jpeg_bytes_per_image"], g["fnv1a_first"]) == (w["jpeg_bytes_per_image"], w["fnv1a_first"]), (kw, env)


def test_random_command_lines_through_the_shipped_libraries_on_the_emulator():
    """a slice of tools/simt/fuzz_cjpeg.py (random cjpeg / jpegtran command lines, three ways each); the tool's longer runs are
    recorded in profiles/r05z_*"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "simt", "fuzz_cjpeg.py"), "11", "60"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert b" 0 failures" in r.stdout


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "tests", "native", "api_fuzz")), reason="tests/native/api_fuzz not built")
def test_random_libjpeg_calls_through_the_shipped_libraries_on_the_emulator():
    """a slice of tools/simt/fuzz_api.py (tests/native/api_fuzz.c: parameters and calls drawn from a seed -- abbreviated datastreams,
    several images from ONE object on the reference's side too, tables of its own, the fast DCT); refusals with a reason are tallied
    by the tool, anything else is a failure"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "simt", "fuzz_api.py"), "7", "50"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert b" 0 failures" in r.stdout

