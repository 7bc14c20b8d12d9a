#!/usr/bin/env python3
"""Per-kernel fingerprint of the gfx950 machine code (test / measurement infrastructure, not product).

`roofline.traffic` in bench.py's line comes from committed PMC passes; a figure may be quoted only for the kernel it was
measured on.  A hash of the SOURCE files (round 4's first stamp) goes stale with every comment, annotation or new,
unrelated kernel in the same file.  This tool fingerprints what actually runs: the assembly the compiler emits for every
`__global__` function (`hipcc -S --cuda-device-only`, the flags of mozjpeg_amd/build.py), cut per function, with the
function-numbered local labels (.LBB<n>_<m>, .Ltmp<n>, .Lfunc_end<n>), comments and debug directives normalised away.  Two
trees give the same fingerprint for a kernel exactly when the compiler produced the same instruction stream, register
allocation and kernel descriptor (.amdhsa_* block) for it.

  python tools/kernel_isa.py                   fingerprints of the working tree -> mozjpeg_amd/kernel_isa.json
  python tools/kernel_isa.py --rev <commit>    the same for a commit (git worktree under /tmp), printed as JSON
  python tools/kernel_isa.py --diff <commit>   which kernels of the working tree differ from that commit's
  python tools/kernel_isa.py --stamp-profiles profiles/r04e_*pmc_hbm_traffic*.json
                                               add to each PMC summary the fingerprints of the kernels it lists, computed on the
                                               tree the passes were taken on (its `profile_head` commit)

bench.py quotes a summary's traffic for an interval while every kernel of the interval has, in mozjpeg_amd/kernel_isa.json
(written by build()), the fingerprint recorded in the summary.
"""
import argparse
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = ["mjh_kernels.hip", "mjh_trellis.hip", "mjh_prog.hip", "mjh_arith.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fno-fast-math", "-w",
         "--cuda-device-only", "-S", "-x", "hip"]
OUT = os.path.join(ROOT, "mozjpeg_amd", "kernel_isa.json")

_label = re.compile(r"\.L(BB|tmp|func_end|func_begin|JTI|CPI)\d+(_\d+)?")


def _normalise(line):
    line = line.split(";", 1)[0].rstrip()            # comments (register statistics, source line notes)
    if not line.strip():
        return None
    s = line.strip()
    if s.startswith((".loc", ".file", ".cfi_", ".ident", ".p2align", ".section", ".text", ".type", ".size", ".weak", ".globl", ".protected", ".hidden")):
        return None
    return _label.sub(lambda m: ".L" + m.group(1) + (m.group(2) or ""), s)


def split_kernels(asm_text):
    """{mangled kernel name: normalised text of its code + its .amdhsa_kernel descriptor block}"""
    lines = asm_text.split("\n")
    # kernels = symbols with an .amdhsa_kernel block
    kernels = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm_text, re.M))
    out = {k: [] for k in kernels}
    cur = None
    in_desc = None
    for ln in lines:
        m = re.match(r"^(\S+):\s*(;.*)?$", ln)
        if m and m.group(1) in kernels:
            cur = m.group(1)
            continue
        if cur and re.match(r"^\s*\.Lfunc_end\d+:", ln):
            cur = None
            continue
        m = re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", ln)
        if m:
            in_desc = m.group(1)
            continue
        if in_desc and re.match(r"^\s*\.end_amdhsa_kernel", ln):
            in_desc = None
            continue
        tgt = cur or in_desc
        if tgt:
            n = _normalise(ln)
            if n is not None:
                out[tgt].append(("D " if in_desc else "") + n)
    return out


def demangle(names):
    filt = shutil.which("llvm-cxxfilt") or shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        res = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, res))
    except Exception:
        return {n: n for n in names}


def short(dem):
    """k_trellis_ac_v3<16, 4, true, false, false>(MjhConst, ...) -> k_trellis_ac_v3<16, 4, true, false, false>"""
    dem = re.sub(r"^void\s+", "", dem)
    depth = 0
    for i, ch in enumerate(dem):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return dem[:i]
    return dem


def fingerprints(tree, jobs=3):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    tmp = tempfile.mkdtemp(prefix="kisa_")
    procs = []
    extra = {}
    try:        # the per-source flags of THAT tree's build (mozjpeg_amd/build.py: EXTRA_FLAGS), so that the fingerprints are the shipped code's
        import runpy
        extra = runpy.run_path(os.path.join(tree, "mozjpeg_amd", "build.py"), run_name="kernel_isa_probe").get("EXTRA_FLAGS", {})
    except Exception:
        extra = {}
    for u in UNITS:
        src = os.path.join(tree, "mozjpeg_amd", "csrc", u)
        if not os.path.exists(src):
            continue
        o = os.path.join(tmp, u + ".s")
        procs.append((u, o, subprocess.Popen([hipcc] + FLAGS + list(extra.get(u, [])) + ["-I" + os.path.join(tree, "include"), src, "-o", o])))
    res = {}
    for u, o, p in procs:
        if p.wait() != 0:
            raise SystemExit("kernel_isa: hipcc failed on " + u)
        parts = split_kernels(open(o).read())
        dm = demangle(list(parts))
        for k, body in parts.items():
            # the kernel descriptor's resources (round 6, third session: two progressive kernels' scratch use went unnoticed for four
            # rounds -- tests/test_abi.py now holds every kernel outside a short list to zero bytes of scratch)
            def field(name, body=body):
                for ln in body:
                    m = re.match(r"^D \.amdhsa_%s\s+(\d+)" % name, ln)
                    if m:
                        return int(m.group(1))
                return None
            res[short(dm[k])] = {"unit": u, "sha": hashlib.sha256("\n".join(body).encode()).hexdigest()[:16], "lines": len(body),
                                 "vgpr": field("next_free_vgpr"), "lds": field("group_segment_fixed_size"),
                                 "scratch": field("private_segment_fixed_size")}
    shutil.rmtree(tmp, ignore_errors=True)
    return res


def checkout(rev):
    d = tempfile.mkdtemp(prefix="kisa_tree_")
    subprocess.check_call(["git", "-C", ROOT, "worktree", "add", "--detach", d, rev], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return d


def drop(d):
    subprocess.call(["git", "-C", ROOT, "worktree", "remove", "--force", d], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def source_stamp(tree=ROOT):
    """sha256 over every file of mozjpeg_amd/csrc that can reach a kernel: kernel_isa.json is trusted only for the sources it
    was computed from"""
    hsh = hashlib.sha256()
    src = os.path.join(tree, "mozjpeg_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h", ".inc")):
            hsh.update(name.encode())
            hsh.update(open(os.path.join(src, name), "rb").read())
    bp = os.path.join(tree, "mozjpeg_amd", "build.py")       # (the compile flags live there)
    if os.path.exists(bp):
        hsh.update(open(bp, "rb").read())
    return hsh.hexdigest()[:16]


def stamp_profiles(paths):
    cache = {}
    for path in paths:
        j = json.load(open(path))
        head = (j.get("profile_head") or "").split("+")[0]
        if not head or "+uncommitted" in (j.get("profile_head") or ""):
            print("skipped (no clean profile_head):", path)
            continue
        if head not in cache:
            d = checkout(head)
            try:
                cache[head] = fingerprints(d)
            finally:
                drop(d)
        fp = cache[head]
        j["kernel_isa"] = {k: fp[k]["sha"] for k in j["kernels"] if k in fp}
        j["kernel_isa_note"] = ("fingerprints of the gfx950 machine code of the kernels above as compiled from the tree at profile_head "
                                "(tools/kernel_isa.py --stamp-profiles; same compiler and flags as the build); bench.py quotes these figures "
                                "while the kernels of the focus interval still compile to the same code")
        json.dump(j, open(path, "w"), indent=1)
        open(path, "a").write("\n")
        print("%s: %d of %d kernels fingerprinted at %s" % (os.path.relpath(path, ROOT), len(j["kernel_isa"]), len(j["kernels"]), head))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rev")
    ap.add_argument("--diff")
    ap.add_argument("--stamp-profiles", nargs="+")
    a = ap.parse_args()
    if a.stamp_profiles:
        stamp_profiles(a.stamp_profiles)
        return
    if a.rev:
        d = checkout(a.rev)
        try:
            json.dump(fingerprints(d), sys.stdout, indent=1, sort_keys=True)
        finally:
            drop(d)
        return
    mine = fingerprints(ROOT)
    if a.diff:
        d = checkout(a.diff)
        try:
            other = fingerprints(d)
        finally:
            drop(d)
        same = [k for k in mine if k in other and other[k]["sha"] == mine[k]["sha"]]
        print("identical machine code: %d kernels" % len(same))
        for k in sorted(mine):
            if k not in other:
                print("  new      ", k)
            elif other[k]["sha"] != mine[k]["sha"]:
                print("  DIFFERENT", k, "(%d -> %d lines)" % (other[k]["lines"], mine[k]["lines"]))
        for k in sorted(other):
            if k not in mine:
                print("  gone     ", k)
        return
    json.dump({"compiler": subprocess.run(["hipcc", "--version"], capture_output=True, text=True).stdout.split("\n")[0],
               "flags": " ".join(FLAGS), "source_stamp": source_stamp(), "kernels": mine}, open(OUT, "w"), indent=1, sort_keys=True)
    print("kernel_isa: %d kernels -> %s" % (len(mine), os.path.relpath(OUT, ROOT)))


if __name__ == "__main__":
    main()
