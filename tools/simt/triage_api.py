#!/usr/bin/env python3
"""One api_fuzz case under the microscope (development aid, build container only): runs tests/native/api_fuzz SEED INDEX against
the reference's libjpeg and against the interposing library on the emulator with API_FUZZ_SAVE, lists each file's marker
segments side by side and names the first segment that differs.
usage: python tools/simt/triage_api.py SEED INDEX [--standalone]"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fuzz_cjpeg as F  # noqa: E402

BIN = os.path.join(F.ROOT, "tests", "native", "api_fuzz")


def segments(b):
    """[(marker, offset, payload bytes incl. entropy data for SOS)]"""
    out, p = [], 0
    while p + 1 < len(b):
        if b[p] != 0xFF:
            out.append((-1, p, b[p:]))
            break
        m = b[p + 1]
        if m in (0xD8, 0xD9):
            out.append((m, p, b""))
            p += 2
            continue
        n = (b[p + 2] << 8) | b[p + 3]
        q = p + 2 + n
        if m == 0xDA:
            while q + 1 < len(b) and not (b[q] == 0xFF and b[q + 1] not in (0, 0xFF) and not 0xD0 <= b[q + 1] <= 0xD7):
                q += 1
        out.append((m, p, b[p + 4:q]))
        p = q
    return out


def describe(m, payload):
    if m == 0xDA:
        n = payload[0]
        return "SOS comps %s Ss %d Se %d Ah %d Al %d, %d data bytes" % ([payload[1 + 2 * i] for i in range(n)], payload[1 + 2 * n], payload[2 + 2 * n], payload[3 + 2 * n] >> 4, payload[3 + 2 * n] & 15, len(payload) - 4 - 2 * n)
    if m == 0xC4:
        ids, q = [], 0
        while q < len(payload):
            nv = sum(payload[q + 1:q + 17])
            ids.append("%02x(%d)" % (payload[q], nv))
            q += 17 + nv
        return "DHT " + " ".join(ids)
    if m == 0xDB:
        ids, q = [], 0
        while q < len(payload):
            ids.append("%02x" % payload[q])
            q += 65 + 64 * (payload[q] >> 4)
        return "DQT " + " ".join(ids)
    return "%02X len %d" % (m & 0xFF, len(payload))


def main():
    seed, index = sys.argv[1], sys.argv[2]
    d = F.dropin_dir()
    tmp = tempfile.mkdtemp(prefix="triage_")
    kw = dict(libpath=os.path.join(d, "standalone")) if "--standalone" in sys.argv else dict(preload=os.path.join(d, "libmozjpeg_hip_jpeg62.so"))
    r0 = F.run([BIN, seed, index], {"API_FUZZ_SAVE": os.path.join(tmp, "ref"), "API_FUZZ_VERBOSE": "1"})
    r1 = F.run([BIN, seed, index], {"API_FUZZ_SAVE": os.path.join(tmp, "our")}, **kw)
    print(r0.stderr.decode(errors="replace"))
    print("reference:", r0.stdout.decode().strip().replace("\n", " | "))
    print("ours     :", r1.stdout.decode().strip().replace("\n", " | "), r1.stderr.decode(errors="replace")[-300:])
    for k in range(8):
        fa, fb = os.path.join(tmp, "ref.%d.jpg" % k), os.path.join(tmp, "our.%d.jpg" % k)
        if not (os.path.exists(fa) and os.path.exists(fb)):
            continue
        a, b = open(fa, "rb").read(), open(fb, "rb").read()
        if a == b:
            print("image %d: identical (%d bytes)" % (k, len(a)))
            continue
        sa, sb = segments(a), segments(b)
        print("image %d: DIFFERENT, %d vs %d bytes" % (k, len(a), len(b)))
        for i in range(max(len(sa), len(sb))):
            x = sa[i] if i < len(sa) else None
            y = sb[i] if i < len(sb) else None
            same = x is not None and y is not None and x[0] == y[0] and x[2] == y[2]
            print("   %s %-60s | %s" % ("  " if same else "!=", describe(x[0], x[2]) if x else "-", describe(y[0], y[2]) if y else "-"))
    print("files in", tmp)


if __name__ == "__main__":
    main()
