#!/usr/bin/env python3
"""The libjpeg API on random call sequences, without a GPU: tests/native/api_fuzz (a client whose parameters and calls are drawn
from a seed) three ways per case -- the reference's libjpeg (expected output), the SHIPPED interposing library in front of it,
the SHIPPED stand-alone library -- with the kernels on the emulator (see fuzz_cjpeg.py).  A run the device path refuses with a
reason ("unsupported configuration (...); no CPU fallback") is tallied by reason, not counted as a failure: that list is what an
application can still ask the reference for and not this library.  Development aid, correctness only; build container only.
usage: python tools/simt/fuzz_api.py SEED COUNT [--verbose] [--fresh]"""
import collections
import os
import re
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fuzz_cjpeg as F  # noqa: E402

BIN = os.path.join(F.ROOT, "tests", "native", "api_fuzz")


def main():
    seed, count = int(sys.argv[1]), int(sys.argv[2])
    verbose = "--verbose" in sys.argv
    d = F.dropin_dir()
    bad = ref_refused = 0
    reasons = collections.Counter()
    t0 = time.time()
    for i in range(count):
        cmd = [BIN, str(seed), str(i)]
        r0 = F.run(cmd, {"API_FUZZ_FRESH": "1"} if "--fresh" in sys.argv else {})     # (--fresh: a new object per image, for telling what an object carries from image to image)
        if r0.returncode != 0:
            ref_refused += 1
            if verbose:
                print("case", i, "reference refuses:", r0.stderr.decode(errors="replace").strip()[-100:], flush=True)
            continue
        msgs = []
        for name, kw in (("shim", dict(preload=os.path.join(d, "libmozjpeg_hip_jpeg62.so"))), ("stand-alone", dict(libpath=os.path.join(d, "standalone")))):
            r = F.run(cmd, {}, **kw)
            err = r.stderr.decode(errors="replace")
            m = re.search(r"unsupported configuration \((.*)\); no CPU fallback", err)
            if r.returncode != 0 and m and r0.stdout.startswith(r.stdout):      # (images before the refused one must agree)
                reasons[re.sub(r"\d+", "N", m.group(1))[:110]] += 1
                continue
            if r.returncode != 0:
                msgs.append("%s: exit %d %s" % (name, r.returncode, err.strip()[-300:]))
            elif r.stdout != r0.stdout:
                msgs.append("%s: DIFFERENT %r vs the reference's %r" % (name, r.stdout, r0.stdout))
        if verbose:
            print("case", i, r0.stdout.decode().strip().replace("\n", " | "), "%.0f s" % (time.time() - t0), flush=True)
        if msgs:
            bad += 1
            print("FAIL: api_fuzz %d %d\n     %s" % (seed, i, "\n     ".join(msgs)), flush=True)
    print("seed %d: %d cases, %d refused by the reference itself, %d failures, %.0f s" % (seed, count, ref_refused, bad, time.time() - t0))
    for k, v in reasons.most_common():
        print("   refused %4d x  %s" % (v, k))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
