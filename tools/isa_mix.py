#!/usr/bin/env python3
"""Static instruction mix of one kernel's loops from `hipcc -S` output (development aid): for every backward branch, the
instructions between its target label and the branch, classed by issue cost as measured by tools/probes/valu_rate.hip on
gfx950 (profiles/r04a_valu_rate_summary.md): 2-cycle VALU (add/sub/logic/right shifts/mov/f32 add, sub, mul), 4-cycle VALU
(compares, cndmask, min/max, left shifts, integer multiplies, conversions, DPP, 3-operand integer ops, packed f32), LDS,
VMEM, SALU.   usage: python tools/isa_mix.py file.s 'substring of the mangled kernel name' [min_loop_instructions]"""
import re
import sys

TWO = re.compile(r"^v_(add|sub|subrev|and|or|xor|not|mov|lshrrev|ashrrev|mul_f32|add_f32|sub_f32|subrev_f32|bfe|accvgpr)")
FOUR_HINT = re.compile(r"^v_(cmp|cmpx|cndmask|min|max|lshlrev|lshl_|mul_|mad_|cvt_|ffbh|ffbl|bfi|perm|alignbit|or3|and_or|add3|lshl_add|lshl_or|add_lshl|xad|pk_|med3|fma|rcp|ceil|floor|trunc|rndne|readlane|writelane|readfirstlane|mbcnt|bcnt|sad)")


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_"):
        if "dpp" in ins or "row_" in ins:
            return "valu4"
        if re.match(r"^v_(mul|add|sub|subrev)_f32", op):
            return "valu2"
        if TWO.match(op) and not FOUR_HINT.match(op):
            return "valu2"
        return "valu4"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    text = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2]
    minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    i = 0
    body = []
    inside = False
    for ln in text:
        m = re.match(r"^(\S+):", ln)
        if m and want in m.group(1) and not m.group(1).startswith(".L"):
            inside = True
            continue
        if inside and re.match(r"^\s*\.Lfunc_end", ln):
            break
        if inside:
            s = ln.split(";", 1)[0].strip()
            if s and not s.startswith((".loc", ".file", ".cfi", ".p2align")):
                body.append(s)
    labels = {}
    for idx, s in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = idx
    tot = {}
    for s in body:
        if not s.endswith(":"):
            c = classify(s)
            tot[c] = tot.get(c, 0) + 1
    print("kernel %s: %d instructions  %s" % (want, sum(tot.values()), tot))
    loops = []
    for idx, s in enumerate(body):
        m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)", s)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < idx and idx - labels[t] >= minlen:
                loops.append((labels[t], idx, t))
    for a, b, t in sorted(loops, key=lambda l: l[0] - l[1])[:12]:
        mix = {}
        for s in body[a:b + 1]:
            if not s.endswith(":"):
                c = classify(s)
                mix[c] = mix.get(c, 0) + 1
        cyc = 2 * mix.get("valu2", 0) + 4 * mix.get("valu4", 0)
        print("loop %-12s lines %5d..%5d: %4d instr  %s   VALU issue cycles if every instruction ran: %d" % (t, a, b, sum(mix.values()), mix, cyc))


if __name__ == "__main__":
    main()
