"""Build tools/simt/_build/libmozjpeg_hip_simt.so: the product's kernel and host sources (mozjpeg_amd/csrc) compiled as plain
C++ against the lock-step wave64 emulator (tools/simt/include/hip/hip_runtime.h + simt.cpp).

Development / test infrastructure: lets a kernel edit be parity-checked against the oracle in a container without a GPU.
Nothing in mozjpeg_amd/ loads this library; tests opt in with `pytest --simt` (tests/conftest.py)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "mozjpeg_amd", "csrc")
OUT = os.environ.get("SIMT_BUILD_DIR") or os.path.join(HERE, "_build")     # (a second directory lets a build go on while tests run from the first)
LIB = os.path.join(OUT, "libmozjpeg_hip_simt.so")
SOURCES = ["mjh_kernels.hip", "mjh_trellis.hip", "mjh_prog.hip", "mjh_arith.hip", "mjh_encoder.cpp", "mjh_pool.cpp", "mjh_guard.cpp", "mjh_numa.cpp"]
# the same floating-point contract as the device build (mozjpeg_amd/build.py): no FMA contraction
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-g1", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-fno-strict-aliasing",
         "-I" + os.path.join(HERE, "include"), "-I" + CSRC]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _sources():
    """The kernel sources to compile.  A tree whose sources carry the emulator's annotations (MJH_DIVERGENT_SCOPE,
    MJH_WAVE_GROUPS, MJH_WAVE_SYNC: nothing on the device, see mjh_device.h) is compiled as it is.  A tree that predates them
    keeps its kernel sources untouched (the PMC traffic summaries under profiles/ are stamped with a hash of those files):
    tools/simt/annotations.patch inserts the same annotations into a COPY under _build/src, which is what gets compiled."""
    patch = os.path.join(HERE, "annotations.patch")
    if "MJH_WAVE_SYNC" in open(os.path.join(CSRC, "mjh_device.h")).read() or not os.path.exists(patch):
        return CSRC
    dst = os.path.join(OUT, "src")
    stamp = os.path.join(dst, ".stamp")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [patch]
    if _newer(stamp, deps):
        shutil.rmtree(dst, ignore_errors=True)
        os.makedirs(os.path.join(dst, "mozjpeg_amd"))
        shutil.copytree(CSRC, os.path.join(dst, "mozjpeg_amd", "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(dst, "include"))     # (the sources include ../../include/mozjpeg_hip.h)
        subprocess.check_call(["patch", "-p1", "-s", "-i", patch], cwd=dst)               # fails loudly when a hunk no longer fits
        open(stamp, "w").close()
    return os.path.join(dst, "mozjpeg_amd", "csrc")


def build(force=False, verbose=False):
    os.makedirs(OUT, exist_ok=True)
    csrc = _sources()
    flags = [f for f in FLAGS if f != "-I" + CSRC] + ["-I" + csrc]
    hdrs = [os.path.join(csrc, h) for h in os.listdir(csrc) if h.endswith(".h")] + \
        [os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "mozjpeg_hip.h")]
    jobs = []
    for s in SOURCES + ["simt.cpp"]:
        src = os.path.join(HERE if s == "simt.cpp" else csrc, s)
        obj = os.path.join(OUT, os.path.splitext(s)[0] + ".o")
        if force or _newer(obj, [src] + hdrs):
            jobs.append(["g++"] + flags + ["-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            for cmd, rc in zip(jobs, ex.map(lambda c: subprocess.call(c), jobs)):
                if verbose:
                    print(" ".join(cmd))
                if rc:
                    raise RuntimeError("simt build failed: " + " ".join(cmd))
    objs = [os.path.join(OUT, os.path.splitext(s)[0] + ".o") for s in SOURCES + ["simt.cpp"]]
    if force or _newer(LIB, objs):
        subprocess.check_call(["g++", "-shared", "-o", LIB] + objs + ["-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
