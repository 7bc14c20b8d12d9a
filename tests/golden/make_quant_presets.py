#!/usr/bin/env python3
"""tests/golden/quant_presets.json: the nine base quantization tables `cjpeg -quant-table N` selects (jcparam.c), read back from
the REAL reference's DQT marker at quality 50 without the baseline clamp (scale factor 100: jpeg_add_quant_table stores the base
values themselves), natural order.  Run in the build container: python tests/golden/make_quant_presets.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

ZZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
      35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def dqt_tables(jpeg):
    """{table number: 64 values in natural order} of a file's DQT marker(s)"""
    out, j = {}, 2
    while j < len(jpeg) - 1:
        if jpeg[j] == 0xFF and jpeg[j + 1] == 0xDB:
            n = (jpeg[j + 2] << 8) | jpeg[j + 3]
            seg, k = jpeg[j + 4:j + 2 + n], 0
            while k < len(seg):
                wide, t = seg[k] >> 4, seg[k] & 15
                k += 1
                nat = [0] * 64
                for i in range(64):
                    if wide:
                        nat[ZZ[i]] = (seg[k] << 8) | seg[k + 1]
                        k += 2
                    else:
                        nat[ZZ[i]] = seg[k]
                        k += 1
                out[t] = nat
            j += 2 + n
        elif jpeg[j] == 0xFF and jpeg[j + 1] == 0xDA:
            break
        else:
            j += 1
    return out


def main():
    assert O.have_ref(), "build the reference first: make -C oracle ref"
    img = O.synthetic_frame(16, 16, 1)
    out = {}
    for idx in range(9):
        t = dqt_tables(O.ref_encode(img, quality=50, quant_table=idx, fastcrush=True, notrellis=True)[0])
        out[str(idx)] = {"luma": t[0], "chroma": t[1]}
    with open(os.path.join(HERE, "quant_presets.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %d base table pairs" % len(out))


if __name__ == "__main__":
    main()
