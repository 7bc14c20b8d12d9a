"""GPU (-m gpu): the libjpeg drop-in boundary.  An UNCHANGED cjpeg binary (built from the reference
sources, oracle/_ref/cjpeg, dynamically linked to the reference libjpeg.so.62) is run with
libmozjpeg_hip_jpeg62.so in front of it; its output files must equal the reference goldens."""
import hashlib
import os
import subprocess

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
def test_unsupported_configuration_is_an_error_without_fallback(tmp_path):
    out = str(tmp_path / "o.jpg")
    r = run_cjpeg(["-quality", "75", "-arithmetic"], out)          # arithmetic coding is outside the GPU path
    assert r.returncode != 0
    assert b"no CPU fallback" in r.stderr


@needs
def test_explicit_passthrough_is_logged(goldens, tmp_path):
    out = str(tmp_path / "o.jpg")
    r = run_cjpeg(["-quality", "75", "-smooth", "10"], out, {"MOZJPEG_HIP_PASSTHROUGH": "1"})
    assert r.returncode == 0, r.stderr.decode()
    assert b"handing over to the host libjpeg" in r.stderr
    assert open(out, "rb").read()[:2] == b"\xff\xd8"
