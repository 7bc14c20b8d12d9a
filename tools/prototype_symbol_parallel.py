#!/usr/bin/env python3
"""Design check for the NEXT step of the progressive path (DESIGN.md section 5 "next", ROUND_NOTES item 1): the chain
kernels code one BLOCK per lane, and PMC counters show 8-10 of 64 lanes active -- a wave runs as long as the block with
the most symbols in the band while most lanes have none.  The remedy is to code one TOKEN per lane.

What a block of an AC-first scan emits on its own (encode_mcu_AC_first jcphuff.c:561-677) is a list of tokens, one per
band position whose value is non-zero after the point transform: r>>4 ZRL symbols, the (run, size) symbol, the value
bits.  This file restates that coder twice on the CPU for a tile of blocks whose bit offsets are already known (the
chain's prefix sums give them today):
  * sequential(): the per-block loop of k_pp_write / k_pp_len / k_pp_stats as they are,
  * token_parallel(): every token computed independently from
        - the tile-wide exclusive prefix sum of the per-block token counts (token t -> block b, index i in the block),
        - the i-th set bit of the block's filtered position mask (its position) and the bit below it (its run),
        - a SEGMENTED exclusive prefix sum of the token lengths (its offset inside the block),
    exactly the quantities a wave computes with popcounts, shuffles and one LDS pass over a compact record
    (non-zero mask + values in position order, mjh_kernels.hip),
and checks that both produce the same (bit offset, length, symbol, value bits) for every token and the same symbol
statistics, on random tiles across bands, point transforms and densities.  Nothing here is product code or oracle; it
needs only numpy.        usage: python tools/prototype_symbol_parallel.py [tiles]"""
import sys

import numpy as np


def size_of(sym):          # any deterministic code-length table will do for an offset check
    return 2 + (sym * 7) % 13


def nbits(a):
    return int(a).bit_length()


def sequential(blocks, offs, Ss, Se, Al):
    """one block after the other, position by position: [(bit offset, length, symbol, value bits, zrl count)]"""
    out, hist = [], {}
    for b, blk in enumerate(blocks):
        pos = offs[b]
        r = 0
        for k in range(Ss, Se + 1):
            v = int(blk[k])
            a = abs(v) >> Al
            if a == 0:
                r += 1
                continue
            z = r >> 4
            r &= 15
            n = nbits(a)
            sym = (r << 4) + n
            length = z * size_of(0xF0) + size_of(sym) + n
            bits = (a if v >= 0 else (~a)) & ((1 << n) - 1)
            out.append((pos, length, sym, bits, z))
            hist[sym] = hist.get(sym, 0) + 1
            if z:
                hist[0xF0] = hist.get(0xF0, 0) + z
            pos += length
            r = 0
    return out, hist


def ith_set_bit(mask, i):
    """position of the i-th (0-based) set bit: on the device a 6-step binary search over popcounts of prefixes"""
    lo, hi = 0, 63
    while lo < hi:
        mid = (lo + hi) >> 1
        if bin(mask & ((2 << mid) - 1)).count("1") > i:
            hi = mid
        else:
            lo = mid + 1
    return lo


def token_parallel(blocks, offs, Ss, Se, Al):
    nb = len(blocks)
    # phase 1 (one lane per block, as today): compact record -> filtered position mask (|v| >> Al != 0 inside the band)
    fmask = []
    for blk in blocks:
        m = 0
        for k in range(Ss, Se + 1):
            if (abs(int(blk[k])) >> Al) != 0:
                m |= 1 << k
        fmask.append(m)
    cnt = np.array([bin(m).count("1") for m in fmask], dtype=np.int64)
    first = np.concatenate([[0], np.cumsum(cnt)[:-1]])          # tile-wide exclusive prefix sum
    total = int(cnt.sum())
    # phase 2 (one lane per TOKEN): everything below is a pure function of the token index t
    blk_of = np.searchsorted(np.cumsum(cnt), np.arange(total), side="right")     # head flags + max-scan on the device
    length = np.zeros(total, dtype=np.int64)
    info = [None] * total
    for t in range(total):
        b = int(blk_of[t])
        i = t - int(first[b])
        k = ith_set_bit(fmask[b], i)
        prev = ith_set_bit(fmask[b], i - 1) if i > 0 else Ss - 1
        run = k - prev - 1
        v = int(blocks[b][k])                                   # device: value at rank popc(record mask below k) of the record
        a = abs(v) >> Al
        n = nbits(a)
        z = run >> 4
        sym = ((run & 15) << 4) + n
        length[t] = z * size_of(0xF0) + size_of(sym) + n
        info[t] = (sym, (a if v >= 0 else (~a)) & ((1 << n) - 1), z)
    # segmented exclusive prefix sum of the token lengths (segments = blocks): scan of all lengths minus the scan value at
    # the block's first token
    ex = np.concatenate([[0], np.cumsum(length)[:-1]]) if total else np.zeros(0, dtype=np.int64)
    out, hist = [], {}
    for t in range(total):
        b = int(blk_of[t])
        inside = int(ex[t] - ex[int(first[b])])
        sym, bits, z = info[t]
        out.append((offs[b] + inside, int(length[t]), sym, bits, z))
        hist[sym] = hist.get(sym, 0) + 1                        # device: one LDS atomic per token, every lane busy
        if z:
            hist[0xF0] = hist.get(0xF0, 0) + z
    return out, hist


def random_tile(rng, n, density, big):
    blocks = np.zeros((n, 64), dtype=np.int64)
    for b in range(n):
        nz = rng.random(64) < density * rng.random()
        mag = np.where(rng.random(64) < big, rng.integers(1, 1024, 64), rng.integers(1, 4, 64))
        blocks[b] = np.where(nz, mag * rng.choice([-1, 1], 64), 0)
    return blocks


def main():
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    rng = np.random.default_rng(7)
    tokens = 0
    for it in range(tiles):
        n = int(rng.integers(1, 65))
        Ss = int(rng.integers(1, 64))
        Se = int(rng.integers(Ss, 64))
        if rng.random() < 0.4:
            Ss, Se = [(1, 63), (1, 8), (9, 63), (1, 2), (3, 63)][int(rng.integers(0, 5))]
        Al = int(rng.integers(0, 4))
        blocks = random_tile(rng, n, float(rng.choice([0.05, 0.3, 0.9])), float(rng.choice([0.0, 0.2])))
        # block bit offsets as the chain's prefix sums deliver them: own bits + an arbitrary flush in front of some blocks
        seq0, _ = sequential(blocks, [0] * n, Ss, Se, Al)
        # (lengths per block from a first sequential pass with offsets 0)
        per_block = [0] * n
        fm_counts = [sum(1 for k in range(Ss, Se + 1) if (abs(int(blocks[i][k])) >> Al) != 0) for i in range(n)]
        idx = 0
        for i in range(n):
            for _ in range(fm_counts[i]):
                per_block[i] += seq0[idx][1]
                idx += 1
        offs, pos = [], int(rng.integers(0, 1000))
        for i in range(n):
            pos += int(rng.integers(0, 40)) if rng.random() < 0.3 else 0     # a flush (EOBRUN symbol + bits) in front of the block
            offs.append(pos)
            pos += per_block[i]
        a, ha = sequential(blocks, offs, Ss, Se, Al)
        p, hp = token_parallel(blocks, offs, Ss, Se, Al)
        assert a == p, (it, n, Ss, Se, Al)
        assert ha == hp, (it, "statistics")
        tokens += len(a)
    print("token-parallel == sequential on %d tiles, %d tokens" % (tiles, tokens))


if __name__ == "__main__":
    main()
