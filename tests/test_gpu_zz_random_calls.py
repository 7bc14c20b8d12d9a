"""GPU (-m gpu), last file of the suite: a slice of the random-call runs of tools/simt (which need no GPU and are where the long
runs happen, profiles/r05z_dropin_fuzz.md) on the chip -- tests/native/api_fuzz (a libjpeg client whose parameters and calls are
drawn from a seed) and random cjpeg command lines, each against the reference's library (expected bytes), with the interposing
library in front of it, and against the stand-alone library."""
import os
import re
import subprocess
import sys

import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "mozjpeg_amd", "libmozjpeg_hip_jpeg62.so")
STANDALONE_DIR = os.path.join(ROOT, "mozjpeg_amd", "standalone")
API_FUZZ = os.path.join(ROOT, "tests", "native", "api_fuzz")
CJPEG = os.path.join(O.REF_DIR, "cjpeg")

needs = pytest.mark.skipif(not (os.path.exists(SHIM) and os.path.exists(os.path.join(STANDALONE_DIR, "libjpeg.so.62")) and os.path.exists(CJPEG)),
                           reason="shim libraries or the reference binaries (oracle/_ref) are not built")


def run(cmd, preload=None, libpath=None, extra=None):
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    env["LD_LIBRARY_PATH"] = libpath or O.REF_DIR
    if preload:
        env["LD_PRELOAD"] = preload
    env.update(extra or {})
    return subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)


@needs
@pytest.mark.skipif(not os.path.exists(API_FUZZ), reason="tests/native/api_fuzz not built")
@pytest.mark.parametrize("index", range(40))
def test_random_libjpeg_calls(index):
    """the same binary on the reference's library gives the expected lines -- ONE object for all images of a case, as the client
    is written: what the object carries from image to image (table flags and contents, cinfo->Ah / Al) is part of the bytes, and so
    are abbreviated datastreams (jpeg_write_tables, jpeg_suppress_tables, write_all_tables FALSE), Huffman tables of the client's
    own and the fast DCT, all of which the seed draws.  A run refused with a reason is not a failure (INTEGRATION.md 1a' lists what)"""
    cmd = [API_FUZZ, "2026", str(index)]
    want = run(cmd)
    import json
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "goldens_calls.json")))["api_fuzz 2026 %d" % index]      # (made by the reference: make_goldens.py --calls)
    if want.returncode != 0:      # the reference itself refuses this draw (an ERREXIT of its own): so must the libraries under test
        assert golden is None
        for kw in (dict(preload=SHIM), dict(libpath=STANDALONE_DIR)):
            assert run(cmd, **kw).returncode != 0, kw
        return
    assert want.stdout.decode() == golden
    for kw in (dict(preload=SHIM), dict(libpath=STANDALONE_DIR)):
        got = run(cmd, **kw)
        if got.returncode != 0 and re.search(rb"unsupported configuration \(.*\); no CPU fallback", got.stderr) and want.stdout.startswith(got.stdout):
            continue
        assert got.returncode == 0, got.stderr.decode(errors="replace")[-1500:]
        assert got.stdout == want.stdout, (kw, got.stdout, want.stdout)


@needs
def test_random_cjpeg_command_lines(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools", "simt"))
    import numpy as np
    import fuzz_cjpeg as F          # (the generator only: nothing of the emulator is built or loaded here)
    checked = 0
    for i in range(28):
        rng = np.random.default_rng(2025 * 100003 + i)
        d = tmp_path / ("c%d" % i)
        d.mkdir()
        if rng.random() < 0.25:      # (the jpegtran share of the tool: not here)
            continue
        twelve = rng.random() < 0.06
        src, w, h, gray_in, fmt = F.write_image(rng, str(d / "in"), 2025, i, twelve)
        a, _ = F.draw_cjpeg(rng, str(d), gray_in, fmt)
        if twelve:
            a = [x for x in a if x != "-trellis-dc"] + ["-precision", "12", "-notrellis"]
        a = [x for x in a if x != "-memdst"]
        outs = []
        for name, kw in (("ref", {}), ("shim", dict(preload=SHIM)), ("alone", dict(libpath=STANDALONE_DIR))):
            out = str(d / (name + ".jpg"))
            r = run([CJPEG, "-dct", "fast" if i % 4 == 1 else "int"] + a + ["-outfile", out, src], **kw)
            if name == "ref" and r.returncode != 0:
                break
            assert r.returncode == 0, (name, a, r.stderr.decode(errors="replace")[-1000:])
            outs.append(open(out, "rb").read())
        if len(outs) == 3:
            assert outs[0] == outs[1] == outs[2], (i, a)
            checked += 1
    assert checked >= 12
