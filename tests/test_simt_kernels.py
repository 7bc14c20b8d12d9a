"""CPU suite: the kernel SOURCES executed by the lock-step wave64 emulator (tools/simt) against the reference-made goldens.

What this is: test infrastructure for a container without a GPU -- mozjpeg_amd/csrc/*.hip|*.cpp compiled as plain C++ against
tools/simt/include/hip/hip_runtime.h (one fiber per lane, rendezvous at cross-lane operations, device buffers that end at an
unmapped page).  It catches what a kernel edit can break before GPU minutes are spent on it: wrong arithmetic or indexing
(bytes differ from the reference's), reads or writes past the end of a buffer (fault), barriers that not every lane reaches
(deadlock report), cross-lane operations in divergent code.  What it is not: a CPU path of the product -- nothing under
mozjpeg_amd/ knows about it (the package loads libmozjpeg_hip.so and nothing else; this module rebinds the ctypes layer for
the duration of its own tests), and timing or the hardware's memory model are out of its reach.  The -m gpu tests remain the
parity tests proper; `pytest -m gpu --simt` runs those same tests against the emulator.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import mozjpeg_amd as M
import oracle_lib as O
from cases import CASES, CASES12, images, images12

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "simt"))


@pytest.fixture(scope="module")
def simt():
    """the ctypes layer bound to the emulator's library for this module only"""
    import build_simt
    path = build_simt.build()
    saved = (M.LIB_PATH, M._lib)
    M.LIB_PATH, M._lib = path, None
    try:
        yield path
    finally:
        M.LIB_PATH, M._lib = saved


# one case per branch of the schedule: sequential with both trellises (the metric's configuration), plain Huffman, 4:4:4 at a
# quality that fills the bigger queues, progressive with the scan search, restart intervals inside progressive scans (the
# all-in-one scan kernel with its token passing), gray, odd sampling, smoothing, the extended trellis options, arithmetic
SIMT_CASES = ["base", "revert", "base_q90_444", "default_progressive", "prog_search_restart1", "gray_prog_restart1",
              "base_4x2_restart1", "base_411_smooth20", "progressive_all_trellis_options_1loop", "base_trellis_q_opt",
              "dc_scan_opt2", "arith_base", "arith_fastcrush"]


@pytest.mark.parametrize("cname", SIMT_CASES)
def test_emulated_kernels_reproduce_the_reference_goldens(simt, cname, goldens):
    kw = [k for c, k, _ in CASES if c == cname][0]
    for iname, img in images().items():
        h, w = img.shape[:2]
        if cname.startswith("arith") and w * h > 250 * 190:
            continue        # (the coding wave is one dependent chain of cross-lane reads: slow in the emulator)
        enc = M.Encoder(M.make_params(w, h, **kw))
        data = enc.encode_host(img)[0]
        enc.close()
        g = goldens["%s/%s" % (iname, cname)]
        assert (len(data), O.md5(data)) == (g["bytes"], g["md5"]), (iname, cname)


@pytest.mark.parametrize("cname", ["p12_base_q90_444", "p12_default_progressive"])
def test_emulated_12bit_kernels_reproduce_the_reference_goldens(simt, cname, goldens):
    kw = [k for c, k, _ in CASES12 if c == cname][0]
    for iname, img in images12().items():
        h, w = img.shape[:2]
        enc = M.Encoder(M.make_params(w, h, **kw))
        data = enc.encode_host(img)[0]
        enc.close()
        g = goldens["%s/%s" % (iname, cname)]
        assert (len(data), O.md5(data)) == (g["bytes"], g["md5"]), (iname, cname)


# kernel variants switched by the environment (A/B knobs that survive: every setting is bit-identical): the files must be the
# default path's = the reference's
KNOBS = [("MJH_TRELLIS_VARIANT", "3", "base_q90_444"), ("MJH_TRELLIS_V3", "2", "base"), ("MJH_DC_SPEC", "0", "default_progressive")]


@pytest.mark.parametrize("knob,value,cname", KNOBS)
def test_emulated_knob_settings_reproduce_the_reference_goldens(simt, knob, value, cname, goldens):
    kw = [k for c, k, _ in CASES if c == cname][0]
    for iname, img in images().items():
        h, w = img.shape[:2]
        try:
            os.environ[knob] = value          # (read when the encoder is made)
            enc = M.Encoder(M.make_params(w, h, **kw))
            data = enc.encode_host(img)[0]
            enc.close()
        finally:
            os.environ.pop(knob, None)
        g = goldens["%s/%s" % (iname, cname)]
        assert (len(data), O.md5(data)) == (g["bytes"], g["md5"]), (knob, iname, cname)


def test_emulated_batch_of_1080p_frames_matches_the_oracle(simt):
    """BASELINE config 2's frame size, a batch of three (distinct frames, one encoder), against the oracle"""
    w, h = 1920, 1080
    frames = np.stack([O.synthetic_frame(w, h, 700 + i) for i in range(3)])
    enc = M.Encoder(M.make_params(w, h, baseline=True), max_batch=3)
    got = enc.encode_host(frames)
    enc.close()
    po = O.make_params(w, h, baseline=True)
    for i in range(3):
        assert got[i] == O.encode(po, frames[i]), i


@pytest.mark.parametrize("chunks", ["2", "4"])
def test_emulated_image_ranges_of_the_ac_trellis_match_the_oracle(simt, chunks):
    """MJH_TRELLIS_CHUNKS: the tile-sorted AC trellis over image ranges with a work list each (the large-batch schedule forced
    onto small frames with MJH_SMALL_BATCH; late DC chains included); a tiny dense-copy budget sends part of every range's
    deferred blocks through the planes"""
    w, h = 227, 149
    frames = np.stack([O.synthetic_frame(w, h, 900 + i) for i in range(9)])
    po = O.make_params(w, h, baseline=True, quality=92)
    want = [O.encode(po, frames[i]) for i in range(9)]
    for dense in (None, "13"):
        try:
            os.environ["MJH_TRELLIS_CHUNKS"] = chunks
            os.environ["MJH_SMALL_BATCH"] = "1"
            os.environ["MJH_TRELLIS_VARIANT"] = "0"       # 16 records at q92: many deferred blocks, both general tiers
            if dense:
                os.environ["MJH_DENSE_CAP"] = dense
            enc = M.Encoder(M.make_params(w, h, baseline=True, quality=92), max_batch=9)
            got = enc.encode_host(frames)
            enc.close()
        finally:
            for k in ("MJH_TRELLIS_CHUNKS", "MJH_SMALL_BATCH", "MJH_TRELLIS_VARIANT", "MJH_DENSE_CAP"):
                os.environ.pop(k, None)
        for i in range(9):
            assert got[i] == want[i], (chunks, dense, i)


@pytest.mark.parametrize("dc", [(0, 2, 2), (1, 3, 3), (2, 0, 0), (3, 1, 3), (0, 2, 0), (2, 3, 2), (1, 1, 0)])
def test_emulated_progressive_images_with_any_two_dc_table_numbers_match_the_oracle(simt, dc):
    """a progressive image may use any two of the four DC table numbers (until round 6: only two with different low bits);
    the interleaved DC scan of the default script codes all three components, each with its table's class"""
    w, h = 83, 61
    img = O.synthetic_frame(w, h, 4242)
    for kw in (dict(quality=80), dict(quality=80, fastcrush=True), dict(quality=70, revert=True, progressive=True)):
        pm, po = M.make_params(w, h, dc_tbl=dc, **kw), O.make_params(w, h, dc_tbl=dc, **kw)
        enc = M.Encoder(pm)
        got = enc.encode_host(img)[0]
        enc.close()
        assert got == O.encode(po, img), (dc, kw)


def test_emulated_gray_trellis_without_optimize_coding_is_the_optimize_schedule(simt, goldens):
    """optimize_coding switched off by hand with the trellis on: one component = the reference's optimize_coding schedule, the same
    bytes (reference-made goldens *_no_optimize, byte-identical to their optimize_coding twins); colour: refused (test_abi)"""
    from cases import images
    for cname in ("base_gray_no_optimize", "base_gray_q90_loops3_no_optimize", "base_gray_restart1_eob_opt_no_optimize", "base_gray_ifast_q_opt_no_optimize"):
        kw = [k for c, k, _ in CASES if c == cname][0]
        for iname, img in images().items():
            h, w = img.shape[:2]
            if w * h > 250 * 190:
                continue
            pm = M.make_params(w, h, **kw)
            assert pm.optimize_coding == 0 and pm.trellis_quant == 1
            enc = M.Encoder(pm)
            data = enc.encode_host(img)[0]
            enc.close()
            g = goldens["%s/%s" % (iname, cname)]
            assert (len(data), O.md5(data)) == (g["bytes"], g["md5"]), (iname, cname)
    pm = M.make_params(64, 64, baseline=True, gray=True, use_scans_in_trellis=True, no_optimize=True)     # two bands per component: not the same schedule
    with pytest.raises(M.MjhError) as ei:
        M.Encoder(pm)
    assert ei.value.code == M.EUNSUPPORTED


def test_three_dc_table_numbers_in_a_progressive_image_are_refused_with_the_reason(simt):
    pm = M.make_params(64, 64, quality=80, dc_tbl=(0, 1, 2))
    with pytest.raises(M.MjhError) as ei:
        M.Encoder(pm)
    assert ei.value.code == M.EUNSUPPORTED and "two" in str(ei.value)


def test_the_emulator_itself(tmp_path):
    """tools/simt/selftest.cpp: cross-lane operations against their documented results (shuffles, ballots under divergence with
    and without MJH_DIVERGENT_SCOPE, DPP row shifts / broadcasts with bound_ctrl, independent rows of 16, __syncthreads_or,
    workgroup-wide LDS, atomics across workgroups), and an overrun of a device buffer must fault."""
    src = os.path.join(ROOT, "tools", "simt", "selftest.cpp")
    exe = str(tmp_path / "simt_selftest")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "tools", "simt", "include"), src,
                           os.path.join(ROOT, "tools", "simt", "simt.cpp"), "-o", exe, "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all ok" in out.stdout
    bad = subprocess.run([exe, "overrun"], capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "all ok" not in bad.stdout          # SIGSEGV at the page behind the buffer
