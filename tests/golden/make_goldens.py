#!/usr/bin/env python3
"""Generate tests/golden/goldens.json from the REAL reference (oracle/_ref/refenc = mozjpeg compiled
from /root/reference by oracle/Makefile).  Run in the build container:
    make -C oracle ref && python tests/golden/make_goldens.py
The JSON (MD5 + size of every fixture x switch set) is committed; /root/reference is not needed to
run the tests."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import oracle_lib as O  # noqa: E402
from cases import CASES, CASES12, PLANE_CASES, TRANSCODE_CASES, images, images12  # noqa: E402


def main():
    assert O.have_ref(), "build the reference first: make -C oracle ref"
    if len(sys.argv) > 2 and sys.argv[1] == "--add":      # only the named cases, merged into the committed file
        names = set(sys.argv[2:])
        path = os.path.join(HERE, "goldens.json")
        out = json.load(open(path))
        for iname, img in images().items():
            for cname, kw, _ in CASES:
                if cname in names:
                    data, _info = O.ref_encode(img, **kw)
                    out["%s/%s" % (iname, cname)] = {"md5": O.md5(data), "bytes": len(data)}
        for iname, img in images12().items():
            for cname, kw, _ in CASES12:
                if cname in names:
                    data, _info = O.ref_encode(img, **kw)
                    out["%s/%s" % (iname, cname)] = {"md5": O.md5(data), "bytes": len(data)}
        with open(path, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        print("goldens.json now holds %d entries" % len(out))
        return
    out = {}
    for iname, img in images().items():
        for cname, kw, _ in CASES:
            data, _info = O.ref_encode(img, **kw)
            out["%s/%s" % (iname, cname)] = {"md5": O.md5(data), "bytes": len(data)}
    for iname, img in images12().items():
        for cname, kw, _ in CASES12:
            data, _info = O.ref_encode(img, **kw)
            out["%s/%s" % (iname, cname)] = {"md5": O.md5(data), "bytes": len(data)}
    with open(os.path.join(HERE, "goldens.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %d goldens" % len(out))
    # planar input through jpeg_write_raw_data (refenc -yuvin): separate file
    outp = {}
    for cname, w, h, kw in PLANE_CASES:
        p = O.make_params(w, h, **kw)
        data = O.ref_encode_planes(O.synthetic_planes(p, 7), w, h, **kw)
        outp[cname] = {"md5": O.md5(data), "bytes": len(data)}
    with open(os.path.join(HERE, "goldens_planes.json"), "w") as f:
        json.dump(outp, f, indent=1, sort_keys=True)
    print("wrote %d plane goldens" % len(outp))
    # transcoding: the real jpegtran on source files made by the oracle
    outt = {}
    imgs = images()
    for cname, iname, src_kw, switches, _kw in TRANSCODE_CASES:
        img = imgs[iname]
        h, w = img.shape[:2]
        src = O.encode(O.make_params(w, h, **src_kw), img)
        data = O.ref_jpegtran(src, switches)
        outt[cname] = {"md5": O.md5(data), "bytes": len(data), "source_md5": O.md5(src)}
    with open(os.path.join(HERE, "goldens_transcode.json"), "w") as f:
        json.dump(outt, f, indent=1, sort_keys=True)
    print("wrote %d transcode goldens" % len(outt))


def calls():
    """goldens_calls.json: what the test clients of the libjpeg API (tests/native) print when they run on the REFERENCE's library --
    call sequences no switch set reaches: abbreviated datastreams, several images from one object, tables of the client's own"""
    import subprocess
    root = os.path.dirname(os.path.dirname(HERE))
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    env["LD_LIBRARY_PATH"] = O.REF_DIR
    out = {}
    for sc in ("abbreviated", "custom_huffman"):
        out["shim_harness " + sc] = subprocess.check_output([os.path.join(root, "tests", "native", "shim_harness"), sc], env=env).decode()
    for i in range(40):
        r = subprocess.run([os.path.join(root, "tests", "native", "api_fuzz"), "2026", str(i)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        out["api_fuzz 2026 %d" % i] = r.stdout.decode() if r.returncode == 0 else None      # None: the reference itself refuses the draw
    with open(os.path.join(HERE, "goldens_calls.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %d call-sequence goldens" % len(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--calls":
        calls()
    else:
        main()
