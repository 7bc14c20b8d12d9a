#!/usr/bin/env python3
"""Design check for the NEXT step of the progressive path ("what would move the numbers next",
item 2): coding an AC REFINEMENT scan without walking its blocks in order.

It restates encode_mcu_AC_refine / emit_eobrun (jcphuff.c:918-1000, :409-431) twice on the CPU -- once as the sequential
state machine, once in the chunk-free "every block on its own + prefix sums" form a GPU kernel would use -- and checks
that both place exactly the same tokens at exactly the same bit offsets on random scans, and that the parallel form
detects every case in which the sequential coder forces a flush (EOBRUN == 0x7FFF, BE > 937), where it must hand the
scan to the sequential walk.  Nothing here is product code or oracle; it needs only numpy.
usage: python tools/prototype_refine_parallel.py [seeds]"""
import sys

import numpy as np


def size_of(sym):          # any deterministic code-length table will do for an offset check
    return 2 + (sym * 7) % 13


def block_tokens(blk, Ss, Se, Al):
    """what ONE block emits on its own: (tokens with bit lengths in order, ne, E, tail correction bits).
    A token is ('Z',) ZRL, ('N', run, sign) newly non-zero coefficient, ('C', bit) correction bit."""
    absv = np.abs(blk) >> Al
    eob = 0
    for k in range(Ss, Se + 1):
        if absv[k] == 1:
            eob = k
    toks, r, br = [], 0, []
    for k in range(Ss, Se + 1):
        t = int(absv[k])
        if t == 0:
            r += 1
            continue
        while r > 15 and k <= eob:
            toks.append((('Z',), size_of(0xF0)))
            toks += [(('C', b), 1) for b in br]
            br, r = [], r - 16
        if t > 1:
            br.append(t & 1)
            continue
        toks.append((('N', r, int(blk[k] >= 0)), size_of((r << 4) + 1) + 1))
        toks += [(('C', b), 1) for b in br]
        br, r = [], 0
    return toks, len(toks) > 0, (r > 0 or len(br) > 0), br


def eobrun_token(run):
    run = int(run)
    nb = run.bit_length() - 1
    return (('E', run), size_of(nb << 4) + nb)


def sequential(blocks, Ss, Se, Al):
    out, pos, forced = [], 0, False
    eobrun, be = 0, []

    def emit(tok, n):
        nonlocal pos
        out.append((pos, tok))
        pos += n

    def emit_eobrun():
        nonlocal eobrun, be
        if eobrun > 0:
            emit(*eobrun_token(eobrun))
            for b in be:
                emit(('C', b), 1)
            eobrun, be = 0, []

    for blk in blocks:
        toks, ne, e, tail = block_tokens(blk, Ss, Se, Al)
        if ne:
            emit_eobrun()            # in front of the block's first symbol
            for tok, n in toks:
                emit(tok, n)
        if e:
            eobrun += 1
            be += tail
            if eobrun == 0x7FFF or len(be) > 1000 - 64 + 1:
                forced = True
                emit_eobrun()
    emit_eobrun()
    return out, pos, forced


def parallel(blocks, Ss, Se, Al):
    n = len(blocks)
    per = [block_tokens(b, Ss, Se, Al) for b in blocks]                 # phase A: every block on its own
    ne = np.array([p[1] for p in per]); e = np.array([p[2] for p in per])
    tail_cnt = np.array([len(p[3]) for p in per])
    T = np.concatenate([[0], np.cumsum(tail_cnt)])                       # exclusive prefix sum of the tail counts
    idx = np.arange(n)
    prev_ne = np.maximum.accumulate(np.where(ne, idx, -1))               # last non-empty block <= i
    p_of = np.concatenate([[-1], prev_ne[:-1]])                          # previous non-empty block < i
    # run in front of a non-empty block j: E(p) + the blocks between; its buffered bits: tails of p .. j-1
    run = np.where(p_of >= 0, e[np.maximum(p_of, 0)].astype(int) + (idx - p_of - 1), idx)
    be = np.where(p_of >= 0, T[idx] - T[np.maximum(p_of, 0)], T[idx])
    last = prev_ne[-1] if n else -1
    final_run = int((int(e[last]) + (n - 1 - last)) if last >= 0 else n)
    final_be = (T[n] - T[last]) if last >= 0 else T[n]
    # forced flushes inside a run = not expressible here: the scan goes to the sequential walk
    need_fallback = bool(np.any(ne & ((run >= 0x7FFF) | (be > 937)))) or final_run >= 0x7FFF or final_be > 937
    if need_fallback:
        return None, None, True
    length = np.zeros(n, dtype=np.int64)
    for j in range(n):
        if ne[j]:
            length[j] = (eobrun_token(int(run[j]))[1] + int(be[j]) if run[j] > 0 else 0) + sum(t[1] for t in per[j][0])
    off = np.concatenate([[0], np.cumsum(length)])
    out = []
    nxt = np.full(n, -1)                                                 # next non-empty block > i (or -1)
    cur = -1
    for i in range(n - 1, -1, -1):
        nxt[i] = cur
        if ne[i]:
            cur = i
    end_pos = int(off[n])
    final_tok = eobrun_token(final_run) if final_run > 0 else None
    for i in range(n):
        if ne[i]:
            pos = int(off[i])
            if run[i] > 0:
                tok, nb = eobrun_token(int(run[i]))
                out.append((pos, tok))
                pos += nb + int(be[i])
            for tok, nb in per[i][0]:
                out.append((pos, tok))
                pos += nb
        if tail_cnt[i]:                                                  # my trailing bits go behind the NEXT flush symbol
            j = nxt[i]
            start = prev_ne[i] if ne[i] else p_of[i]                    # the block whose tail opens my run
            rel = int(T[i] - (T[start] if start >= 0 else 0))
            base = int(off[j]) + eobrun_token(int(run[j]))[1] if j >= 0 else end_pos + final_tok[1]
            for q, b in enumerate(per[i][3]):
                out.append((base + rel + q, ('C', b)))
    total = end_pos
    if final_tok:
        out.append((end_pos, final_tok[0]))
        total += final_tok[1] + int(final_be)
    return sorted(out), total, False


def random_scan(rng, n, density, big):
    blocks = np.zeros((n, 64), np.int64)
    mask = rng.random((n, 64)) < density
    vals = rng.integers(1, 4 if not big else 40, (n, 64)) * rng.choice([-1, 1], (n, 64))
    blocks[mask] = vals[mask]
    dead = rng.random(n) < 0.5                                           # long stretches of empty blocks
    blocks[dead] = 0
    return blocks


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    checked = fallbacks = 0
    for seed in range(seeds):
        rng = np.random.default_rng(seed)
        n = int(rng.integers(1, 3000))
        Ss = int(rng.integers(1, 20)); Se = int(rng.integers(Ss, 64)); Al = int(rng.integers(0, 3))
        blocks = random_scan(rng, n, float(rng.choice([0.002, 0.02, 0.2, 0.6])), bool(rng.integers(0, 2)))
        if seed % 7 == 3:                                                # many correction bits in one run: BE > 937
            blocks[:] = 0
            blocks[: n // 2, Ss:Se + 1] = 2 << Al
            blocks[-1, Ss] = 1 << Al
        seq, seq_bits, forced = sequential(blocks, Ss, Se, Al)
        par, par_bits, fb = parallel(blocks, Ss, Se, Al)
        assert fb == forced, (seed, fb, forced)
        if fb:
            fallbacks += 1
            continue
        assert par_bits == seq_bits, (seed, par_bits, seq_bits)
        assert par == sorted(seq), seed
        checked += 1
    print("parallel refinement coding == sequential on %d random scans; %d scans correctly sent to the sequential walk"
          % (checked, fallbacks))


if __name__ == "__main__":
    main()
