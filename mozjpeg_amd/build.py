"""Build libmozjpeg_hip.so (HIP kernels + host pipeline + C ABI) for gfx950, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmozjpeg_hip.so")
SHIM = os.path.join(HERE, "libmozjpeg_hip_jpeg62.so")
STANDALONE = os.path.join(HERE, "standalone", "libjpeg.so.62")
TJSHIM = os.path.join(HERE, "libmozjpeg_hip_turbojpeg.so")
SOURCES = ["mjh_kernels.hip", "mjh_trellis.hip", "mjh_prog.hip", "mjh_arith.hip", "mjh_encoder.cpp", "mjh_pool.cpp", "mjh_guard.cpp", "mjh_numa.cpp"]
# -ffp-contract=off: the trellis / deringing float recipes must not be fused into FMAs
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]
# per source: the SLP vectoriser pairs the trellis' and the colour kernel's float adds into v_pk_add_f32 and pays for it with
# v_mov shuffles (packed f32 issues at half rate on gfx950, profiles/r04a_valu_rate_summary.md): without it the metric's step with
# two batches in flight is 3.90 instead of 4.00 ms (profiles/r06p_noslp.md); the float recipes are untouched (no reassociation either way)
EXTRA_FLAGS = {"mjh_kernels.hip": ["-fno-slp-vectorize"],
               # the AC trellis' walks (dependent LDS look-ups + float adds): the max-ILP scheduling strategy; it costs the bit writers
               # and the FDCT kernel more than it buys them, hence a translation unit of its own (profiles/r06q_sched.md)
               "mjh_trellis.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith((".h", ".inc"))] + \
        [os.path.join(HERE, "..", "include", "mozjpeg_hip.h")]
    # one object per source, rebuilt when its source or any shared header / .inc is newer; the translation units compile
    # side by side (the five kernel files take ~40-60 s each)
    hdrs = [d for d in deps if d not in srcs] + [os.path.abspath(__file__)]      # (the flags live in this file)
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(CSRC, os.path.splitext(os.path.basename(s))[0] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(s), []) + ["-x", "hip", "-c", s, "-o", o])
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            for cmd, rc in zip(jobs, ex.map(subprocess.call, jobs)):
                if verbose:
                    print(" ".join(cmd))
                if rc:
                    raise subprocess.CalledProcessError(rc, cmd)
    if force or _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_shim(force=False, verbose=False):
    """libjpeg drop-in entry points (jpeg_start_compress / jpeg_write_scanlines /
    jpeg_finish_compress) on top of the batch encoder.  It is compiled against the libjpeg headers
    of the tree it drops into (struct jpeg_compress_struct is ABI), so it is only built where
    /root/reference (and the generated config headers under oracle/_ref/include) exist."""
    src = os.path.join(CSRC, "jpeg_shim.c")
    ref = os.environ.get("MOZJPEG_REFERENCE", "/root/reference")
    cfg = os.path.join(HERE, "..", "oracle", "_ref", "include")
    if not (os.path.exists(src) and os.path.exists(os.path.join(ref, "jpeglib.h")) and os.path.exists(cfg)):
        if verbose:
            print("shim: reference headers not present, keeping prebuilt", SHIM if os.path.exists(SHIM) else "(none)")
        return SHIM if os.path.exists(SHIM) else None
    hdr = os.path.join(CSRC, "jpeg_shim.h")
    inc = ["-I" + cfg, "-I" + ref, "-I" + os.path.join(HERE, "..", "include"), "-I" + CSRC]
    if force or _newer(SHIM, [src, hdr, LIB]):
        # preload / link-order flavour: the hot-path entry points + abort/destroy hooks in front of a libjpeg
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-Wall"] + inc + ["-o", SHIM, src, "-L" + HERE, "-l:libmozjpeg_hip.so",
                                                                     "-Wl,-rpath,$ORIGIN", "-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    api = os.path.join(CSRC, "jpeg_api.c")
    if force or _newer(STANDALONE, [src, api, hdr, LIB, os.path.join(CSRC, "mjh_quant_presets.h")]):
        # stand-alone flavour: a complete libjpeg.so.62 for the COMPRESS API, nothing of the reference at run time
        os.makedirs(os.path.dirname(STANDALONE), exist_ok=True)
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-DMJH_STANDALONE"] + inc + ["-o", STANDALONE, src, api, "-L" + HERE,
                                                                                         "-l:libmozjpeg_hip.so", "-Wl,-soname,libjpeg.so.62",
                                                                                         "-Wl,-rpath,$ORIGIN/..", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    tj = os.path.join(CSRC, "tj_shim.c")
    if force or _newer(TJSHIM, [tj, LIB]):
        # TurboJPEG-signature entry points (tjCompress2 / tj3Compress8 / ...FromYUV...) straight on the batch encoder
        cmd = ["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-I" + ref, "-I" + os.path.join(HERE, "..", "include"), "-o", TJSHIM, tj,
               "-L" + HERE, "-l:libmozjpeg_hip.so", "-Wl,-rpath,$ORIGIN", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return SHIM


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
