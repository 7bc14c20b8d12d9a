#!/usr/bin/env python3
"""Exhaustive check of the multiply-high division the kernels use for 8-bit quantization tables (MjhQuant.mdiv / sdiv):
floor(n / 8q) == ((n << sdiv) * mdiv) >> 32 for every q in 1..255 and every n in [0, 2^16), with both factors below 2^24."""
import numpy as np

n = np.arange(0, 1 << 16, dtype=np.uint64)
for q in range(1, 256):
    d = 8 * q
    k = min(32, 22 + d.bit_length())
    m = (1 << k) // d + 1
    assert m < (1 << 24)
    a = n << np.uint64(32 - k)
    assert int(a.max()) < (1 << 24)
    assert np.array_equal((a * np.uint64(m)) >> np.uint64(32), n // np.uint64(d)), q
print("ok: exact for q = 1..255, n < 65536")
