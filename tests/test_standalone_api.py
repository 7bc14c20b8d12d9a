"""CPU: the stand-alone libjpeg replacement (mozjpeg_amd/standalone/libjpeg.so.62 = jpeg_shim.c + jpeg_api.c, SURVEY 8f
row 3) against the reference's library on everything that happens on the host: tests/native/api_probe prints what the
COMPRESS API leaves in a compress object after every setter (parameters, quantization / Huffman tables, scan scripts incl.
dc_scan_opt_mode and the search script, message texts, memory-manager protocol, tables-only datastream through a growing
memory destination); the same binary must print the same text with either library.  No pixel is compressed, no GPU."""
import os
import subprocess

import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "native", "api_probe")
STANDALONE_DIR = os.path.join(ROOT, "mozjpeg_amd", "standalone")
have = os.path.exists(PROBE) and os.path.exists(os.path.join(STANDALONE_DIR, "libjpeg.so.62")) and \
    os.path.exists(os.path.join(O.REF_DIR, "libjpeg.so.62"))


def run(libdir):
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    env["LD_LIBRARY_PATH"] = libdir
    r = subprocess.run([PROBE], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    return r.stdout.decode()


@pytest.mark.skipif(not have, reason="api_probe / stand-alone library / reference library not built")
def test_host_side_api_is_indistinguishable_from_the_reference():
    want, got = run(O.REF_DIR), run(STANDALONE_DIR)
    assert len(want.splitlines()) > 250
    assert got == want


@pytest.mark.skipif(not os.path.exists(os.path.join(STANDALONE_DIR, "libjpeg.so.62")), reason="stand-alone library not built")
def test_standalone_library_exports_what_an_unchanged_cjpeg_needs_and_nothing_of_the_reference():
    cj = os.path.join(O.REF_DIR, "cjpeg")
    lib = os.path.join(STANDALONE_DIR, "libjpeg.so.62")
    exported = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode().splitlines() if ln.strip()}
    if os.path.exists(cj):
        wanted = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--undefined-only", cj]).decode().splitlines()
                  if ln.split()[-1].startswith(("jpeg", "jround", "jdiv", "jcopy", "jzero", "jinit"))}
        assert wanted and not (wanted - exported), wanted - exported
    deps = subprocess.check_output(["ldd", lib]).decode()
    assert "oracle" not in deps and "libjpeg" not in deps.replace(lib, "")
