"""CPU: the product library loads, exports every symbol include/mozjpeg_hip.h declares, the host
helpers mirror the reference's parameter logic, and the encoder refuses to run without a GPU
(no CPU fallback).  No compute is launched here."""
import ctypes as C
import os
import re

import pytest

import mozjpeg_amd as M
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "mozjpeg_hip.h")).read()
    return sorted(set(re.findall(r"\b(mjh_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    L = M.lib()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), "missing export " + s


def test_params_struct_layout_matches_header():
    # sizeof(mjh_params) computed from the header's field list: ints/floats are 4 bytes
    assert C.sizeof(M.Params) == 4 * (4 + 6 * 4) + 2 * 64 * 4 + 4 * 5 + 4 * 2 + 4 * 3 + C.sizeof(M.Scan) * 64 + 8 + 16 + 4 + 4 + 4 + 4 + 6 * 4 + 4 + 6 * 4 + 2 * 4 + 4 + 8 * 17 + 8 * 256 + 4   # (+ arith_code, + arith_dc_L / arith_dc_U / arith_ac_K of two tables; round 6: + trellis_stats_Ah / Al, huff_tables_given, huff_bits, huff_vals, dct_method)
    assert C.sizeof(M.Scan) == 4 * 9


@pytest.mark.parametrize("kw", [dict(baseline=True), dict(revert=True), dict(baseline=True, quality=30),
                                dict(baseline=True, quality=92, sample=(1, 1)), dict(revert=True, quality=10),
                                dict(baseline=True, quant_table=0), dict(baseline=True, gray=True)])
def test_host_parameter_helpers_match_oracle(kw):
    """mjh_params_defaults / mjh_params_set_quality reproduce jpeg_set_defaults / jpeg_set_quality"""
    pg = M.make_params(333, 222, **kw)
    po = O.make_params(333, 222, **kw)
    assert pg.num_components == po.num_components
    for i in range(po.num_components):
        assert (pg.h_samp_factor[i], pg.v_samp_factor[i]) == (po.h_samp[i], po.v_samp[i])
        assert pg.quant_tbl_no[i] == po.quant_tbl_no[i]
        assert list(pg.quantval[pg.quant_tbl_no[i]]) == list(po.qtbl[po.quant_tbl_no[i]])
    assert bool(pg.optimize_coding) == bool(po.optimize_coding)
    assert bool(pg.trellis_quant) == bool(po.trellis_quant)
    assert bool(pg.overshoot_deringing) == bool(po.overshoot_deringing)
    assert (pg.compress_profile == M.PROFILE_FASTEST) == bool(po.fastest_profile)


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_unsupported_configurations_are_errors_not_fallbacks():
    p = M.make_params(64, 64, baseline=True, precision=12)  # 12-bit + trellis: the reference itself aborts
    with pytest.raises(M.MjhError) as ei:
        M.Encoder(p)
    assert ei.value.code == M.EUNSUPPORTED
    p = M.make_params(64, 64, baseline=True, sample=(2, 2))
    p.h_samp_factor[1] = 3                                  # a fractional ratio (2 against 3): JERR_FRACT_SAMPLE_NOTIMPL in the reference
    with pytest.raises(M.MjhError) as ei:
        M.Encoder(p)
    assert ei.value.code == M.EUNSUPPORTED
    p = M.make_params(64, 64, baseline=True, sample=((2, 2), (2, 2), (2, 2)))   # 12 blocks per MCU (at most 10)
    with pytest.raises(M.MjhError) as ei:
        M.Encoder(p)
    assert ei.value.code == M.EINVAL
    p = M.make_params(64, 64, scans=[((0,), 0, 63, 0, 0), ((0, 1, 2), 0, 63, 0, 0)])   # a sequential script that codes component 0 twice
    with pytest.raises(M.MjhError) as ei:
        M.Encoder(p)
    assert ei.value.code == M.EINVAL
    p = M.make_params(64, 64)
    p.scan_info[5].Ss = 2                                  # optimize_scans with a script that is not jpeg_search_progression's
    with pytest.raises(M.MjhError) as ei:
        M.Encoder(p)
    assert ei.value.code == M.EUNSUPPORTED
    p = M.make_params(64, 64, baseline=True)
    p.optimize_coding = 0               # trellis without optimize_coding: jcmaster.c never selects a component
    with pytest.raises(M.MjhError) as ei:
        M.Encoder(p)
    assert ei.value.code == M.EUNSUPPORTED
    p = M.make_params(64, 64, baseline=True, gray=True, gray_sample=(1, 5))   # a sampling factor outside 1..4 (JERR_BAD_SAMPLING)
    with pytest.raises(M.MjhError) as ei:
        M.Encoder(p)
    assert ei.value.code == M.EINVAL
    for dc, ac in (((1, 1, 1), (0, 0, 1)), ((0, 1, 0), (1, 1, 0))):
        # a component reuses an earlier one's DC table with an AC table nobody had: the reference's one-marker DHT writer
        # (emit_multi_dht jcmarker.c:293-401) writes that AC table without its values, outside the marker's length -- a corrupt file
        # (tools/simt/fuzz_api.py; the oracle restates it and the reference's djpeg rejects the result)
        p = M.make_params(64, 64, baseline=True)
        for i in range(3):
            p.dc_tbl_no[i], p.ac_tbl_no[i] = dc[i], ac[i]
        with pytest.raises(M.MjhError) as ei:
            M.Encoder(p)
        assert ei.value.code == M.EUNSUPPORTED and "emit_multi_dht" in str(ei.value)
    p = M.make_params(0, 64, baseline=True)
    with pytest.raises(M.MjhError) as ei:
        M.Encoder(p)
    assert ei.value.code == M.EINVAL
    for field, bad in (("smoothing_factor", 101), ("smoothing_factor", -1), ("trellis_num_loops", 17), ("trellis_num_loops", -2)):
        p = M.make_params(64, 64, baseline=True)
        setattr(p, field, bad)
        with pytest.raises(M.MjhError) as ei:
            M.Encoder(p)
        assert ei.value.code == M.EINVAL, field
    p = M.make_params(64, 64, baseline=True, sample=(4, 4))     # 16 + 2 blocks per MCU > 10 (jcmaster.c:540-544: the reference errors out too)
    with pytest.raises(M.MjhError) as ei:
        M.Encoder(p)
    assert ei.value.code == M.EINVAL


@pytest.mark.skipif(_gpu_present(), reason="only meaningful on a machine without a GPU")
def test_no_gpu_means_loud_failure():
    with pytest.raises(M.MjhError) as ei:
        M.Encoder(M.make_params(64, 64, baseline=True))
    assert ei.value.code == M.EHIP
    assert "no CPU fallback" in str(ei.value)


def test_shim_exports_every_symbol_its_contract_names():
    """include/mozjpeg_hip_jpeglib.h lists the libjpeg entry points the drop-in library must export"""
    import subprocess
    shim = os.path.join(ROOT, "mozjpeg_amd", "libmozjpeg_hip_jpeg62.so")
    if not os.path.exists(shim):
        pytest.skip("shim not built (needs the reference's libjpeg headers)")
    hdr = open(os.path.join(ROOT, "include", "mozjpeg_hip_jpeglib.h")).read()
    names = re.search(r'MOZJPEG_HIP_SHIM_SYMBOLS\s+"([^"]+)"', hdr).group(1).split()
    assert len(names) == 11
    exported = subprocess.check_output(["nm", "-D", "--defined-only", shim]).decode()
    for n in names:
        assert re.search(r"\sT\s+%s\b" % re.escape(n), exported), "shim does not export " + n


@pytest.mark.skipif(_gpu_present(), reason="only meaningful on a machine without a GPU")
def test_drop_in_without_gpu_fails_loudly_and_writes_nothing(tmp_path):
    """the unchanged cjpeg with the drop-in in front, on a machine without a GPU: an error exit, never a silent CPU encode"""
    import subprocess
    shim = os.path.join(ROOT, "mozjpeg_amd", "libmozjpeg_hip_jpeg62.so")
    cjpeg = os.path.join(O.REF_DIR, "cjpeg")
    if not (os.path.exists(shim) and os.path.exists(cjpeg)):
        pytest.skip("shim or reference cjpeg not built")
    out = str(tmp_path / "o.jpg")
    env = dict(os.environ, LD_PRELOAD=shim)
    env.pop("MOZJPEG_HIP_PASSTHROUGH", None)
    r = subprocess.run([cjpeg, "-quality", "75", "-baseline", "-outfile", out, os.path.join(ROOT, "tests", "golden", "testorig.ppm")],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0
    assert b"no CPU fallback" in r.stderr
    assert not os.path.exists(out) or os.path.getsize(out) == 0


def test_batch_shape_normalisation_of_the_binding():
    """gray images may come with or without the channel axis; only the encoder's geometry tells a [n, H, W] gray batch from
    one [H, W, C] image (a seeded fuzz case once encoded a two-image gray batch as one image)"""
    import numpy as np
    pg = M.make_params(7, 5, gray=True, grayin=True, baseline=True)
    for shape, want in (((5, 7), (1, 5, 7, 1)), ((5, 7, 1), (1, 5, 7, 1)), ((2, 5, 7), (2, 5, 7, 1)), ((2, 5, 7, 1), (2, 5, 7, 1)), ((1, 5, 7), (1, 5, 7, 1))):
        assert M._as_batch(pg, np.zeros(shape, np.uint8)).shape == want
    pc = M.make_params(7, 5, baseline=True)
    for shape, want in (((5, 7, 3), (1, 5, 7, 3)), ((2, 5, 7, 3), (2, 5, 7, 3)), ((5, 7, 4), (1, 5, 7, 4))):
        assert M._as_batch(pc, np.zeros(shape, np.uint8)).shape == want
    pw = M.make_params(1, 5, gray=True, grayin=True, baseline=True)   # one pixel wide: the channel axis is ambiguous by value, not by geometry
    for shape, want in (((5, 1), (1, 5, 1, 1)), ((5, 1, 1), (1, 5, 1, 1)), ((3, 5, 1), (3, 5, 1, 1)), ((3, 5, 1, 1), (3, 5, 1, 1))):
        assert M._as_batch(pw, np.zeros(shape, np.uint8)).shape == want
    with pytest.raises(AssertionError):
        M._as_batch(pc, np.zeros((6, 7, 3), np.uint8))


def test_committed_bench_line_carries_the_contract_fields():
    """the newest committed bench line (profiles/r*_bench_batch64.json, written by bench.py on the GPU box) has every field
    the driver contract names, with the metric string of BASELINE.json"""
    import glob
    import json
    lines = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "r*_bench_batch64.json")) if "_c" not in os.path.basename(p).split("_bench")[0])
    assert lines
    j = json.loads(open(lines[-1]).read())
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert j["metric"] == base["metric"]
    for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in j, k
    assert j["higher_is_better"] is True and j["scaling"] in ("weak", "strong") and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    # consistency: the whole job's rate follows from the step time, and the dominant kernel fits inside a step
    px = 3840 * 2160 * j["config"]["frames_per_step_per_gpu"] * j["n_gpus"]
    assert abs(j["value"] - px / (j["ms_per_step"] * 1e-3) / 1e6) / j["value"] < 0.01
    assert r["kernel_ms"] <= j["ms_per_step"]
    assert j["bit_exact"]["ok"] and j["bit_exact"]["checked"] == j["bit_exact"]["identical"]


def test_committed_bench_lines_of_the_other_configurations_name_their_own_dominant_interval():
    """every committed bench line (the metric and the other BASELINE configurations) reports as roofline.kernel the largest
    entry of its OWN per-kernel table (side-stream kernels excluded: they overlap the interval they are listed beside), and
    the traffic figure comes from PMC passes of the same configuration or is null"""
    import glob
    import json
    newest = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]*bench*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]*configs.jsonl")),
                       key=os.path.basename):
        for ln in open(path).read().strip().splitlines():
            try:
                j = json.loads(ln)
            except ValueError:
                continue
            if "roofline" in j and j.get("n_gpus") == 1:
                newest[j["config"]["config_key"]] = (path, j)
    assert "metric" in newest
    for key, (path, j) in newest.items():
        r = j["roofline"]
        table = {k: v for k, v in r["kernel_ms_per_call(untimed pass, every kernel bracketed)"].items()
                 if "side stream" not in k and not k.startswith("join(")}
        top = max(table.values())
        assert table[r["kernel"]] >= 0.9 * top, (path, key, r["kernel"])     # (two passes: near-ties may swap)
        src = r.get("traffic_source") or ""
        if r["traffic"] is not None and key != "metric":
            assert "_%s_" % key in src, (path, src)


def test_traffic_figures_are_quoted_only_for_the_kernels_they_were_measured_on(monkeypatch):
    """roofline.traffic comes from committed PMC passes, so it must not outlive the kernels it was
    taken on.  tools/pmc_traffic.py stamps every summary with a hash of the kernel sources, the git head and a fingerprint
    of each kernel's machine code (tools/kernel_isa.py); bench.py quotes a summary while the sources hash to its stamp, or
    -- when only comments, annotations or other kernels changed -- while every kernel of the focus interval still compiles
    to the fingerprinted code, and says 'stale' otherwise."""
    import glob
    import json
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import kernel_isa
    stamp = bench.kernel_source_stamp()
    traffic, src = bench.dominant_traffic("metric", "trellis_ac", 64)
    if traffic is not None:
        pmc = json.load(open(os.path.join(ROOT, src.split(" ")[0])))
        assert "git" in src
        if pmc["kernel_source_stamp"] != stamp:
            now = bench.kernel_fingerprints()
            focus = [k for k in pmc["kernels"] if k.startswith("k_trellis_ac")]
            assert focus and all(now[k] == pmc["kernel_isa"][k] for k in focus) and "machine code" in src
    else:
        assert src.startswith("stale") or src.startswith("no PMC passes")
    # any other tree: other sources AND other machine code
    monkeypatch.setattr(bench, "kernel_source_stamp", lambda: "0" * 16)
    monkeypatch.setattr(bench, "kernel_fingerprints", lambda: {})
    traffic, src = bench.dominant_traffic("metric", "trellis_ac", 64)
    assert traffic is None and (src.startswith("stale") or src.startswith("no PMC passes"))
    # one kernel of the interval recompiled to something else: stale, whatever the others do
    real = json.load(open(os.path.join(ROOT, "mozjpeg_amd", "kernel_isa.json")))
    fp = {k: v["sha"] for k, v in real["kernels"].items()}
    newest = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_pmc_hbm_traffic_batch64.json") and p.count("_") == 4)[-1]
    fp[next(k for k in json.load(open(os.path.join(ROOT, "profiles", newest)))["kernels"] if k.startswith("k_trellis_ac_qd"))] = "f" * 16
    monkeypatch.setattr(bench, "kernel_fingerprints", lambda: fp)
    traffic, src = bench.dominant_traffic("metric", "trellis_ac", 64)
    assert traffic is None and src.startswith("stale")
    # ... while an interval that kernel does not belong to is still quoted (when its passes are committed at all)
    t2, s2 = bench.dominant_traffic("metric", "dct_quant", 64)
    assert t2 is not None or s2.startswith("no PMC passes") or s2.startswith("stale")
    # the committed fingerprints belong to the sources in the tree (build() refreshes them; `python tools/kernel_isa.py`)
    assert real["source_stamp"] == kernel_isa.source_stamp(), "mozjpeg_amd/kernel_isa.json is older than mozjpeg_amd/csrc: run python tools/kernel_isa.py"
    # the summaries of this round carry the stamps
    new = [p for p in glob.glob(os.path.join(ROOT, "profiles", "r0[4-9]*pmc_hbm_traffic*.json"))]
    for p in new:
        j = json.load(open(p))
        assert len(j.get("kernel_source_stamp", "")) == 16 and j.get("profile_head"), p


def test_no_kernel_of_a_throughput_path_uses_scratch_memory():
    """mozjpeg_amd/kernel_isa.json carries every kernel's descriptor resources (tools/kernel_isa.py, rewritten by build()).  Scratch
    (.amdhsa_private_segment_fixed_size) in a kernel is either a register spill or -- what round 6's third session found in the
    progressive kernels after four rounds -- a local array / struct the compiler could not keep in registers (a struct copy whose
    member arrays are indexed dynamically; an if / else-if on two struct members turned into a read-modify-write through a selected
    address): memory traffic per lane on the hot path that no profile names.  Every kernel outside the two cold ones below stays at 0."""
    import json
    real = json.load(open(os.path.join(ROOT, "mozjpeg_amd", "kernel_isa.json")))
    allowed = {"k_trellis_arith": 1024,       # per-lane DP arrays of the arithmetic coder's trellis (a completeness path)
               "k_prog_scan<1>": 64}          # spills in the walk of progressive scans WITH restart intervals
    seen = 0
    for name, k in real["kernels"].items():
        assert k.get("scratch") is not None and k.get("vgpr"), "kernel_isa.json without resources: run python tools/kernel_isa.py"
        seen += 1
        assert k["scratch"] <= allowed.get(name, 0), "%s uses %d bytes of scratch memory" % (name, k["scratch"])
    assert seen > 100
    # the metric's dominant kernel keeps the occupancy its design counts on: 16 waves per CU = at most 10 240 bytes of LDS and 128 VGPRs
    v3 = real["kernels"]["k_trellis_ac_v3<16, 4, true>"]
    assert v3["lds"] <= 10240 and v3["vgpr"] <= 128


def test_kernel_fingerprints_ignore_labels_and_comments_but_not_code():
    """tools/kernel_isa.py: the per-kernel text that gets hashed"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_isa as K
    asm = """
\t.text
\t.globl\t_Z3k_av
_Z3k_av:                                ; @_Z3k_av
; %bb.0:
\ts_load_dword s0, s[0:1], 0x0          ; a comment
.LBB7_2:
\tv_add_u32_e32 v1, s0, v0
\ts_cbranch_execz .LBB7_2
\ts_endpgm
.Lfunc_end7:
\t.amdhsa_kernel _Z3k_av
\t\t.amdhsa_next_free_vgpr 2
\t.end_amdhsa_kernel
"""
    a = K.split_kernels(asm)["_Z3k_av"]
    b = K.split_kernels(asm.replace(".LBB7_2", ".LBB12_2").replace("Lfunc_end7", "Lfunc_end12").replace("a comment", "another"))["_Z3k_av"]
    c = K.split_kernels(asm.replace("v_add_u32_e32 v1", "v_add_u32_e32 v2"))["_Z3k_av"]
    d = K.split_kernels(asm.replace("next_free_vgpr 2", "next_free_vgpr 3"))["_Z3k_av"]
    assert a == b and a != c and a != d and len(a) == 6
    assert K.short("void k_x<16, 4, true>(MjhConst, int*)") == "k_x<16, 4, true>"


def test_arithmetic_coder_on_the_host_matches_a_plain_restatement_of_jcarith(tmp_path):
    """mozjpeg_amd/csrc/mjh_arith_coder.h compiles for the host (a vector register = an array of 64 ints):
    tests/native/arith_coder_check.cpp runs the product's coder -- bins by kind, one per lane, Qe cached in the bin, bound
    tables, renormalisation by count-leading-zeros -- against a plain restatement of jcarith.c (flat state bytes, bit-by-bit
    renormalisation) on random whole-block / DC / AC first / AC refinement scans with restarts, and compares bytes, registers
    and every statistics bin."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no host C++ compiler")
    exe = str(tmp_path / "arith_coder_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "native", "arith_coder_check.cpp")])
    out = subprocess.run([exe, "300", "4242"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "identical" in out.stdout, out.stdout[-2000:]


def test_numa_placement_helpers_parse_and_degrade():
    """mjh_numa.cpp: the cpulist parser behind mjh_bind_thread_to_device, and "no placement" without a device (this
    container has no GPU: the node is unknown, nothing is bound, the description says so)"""
    L = M.lib()
    import ctypes.util  # noqa: F401

    class CpuSet(C.Structure):
        _fields_ = [("bits", C.c_ulong * 16)]     # cpu_set_t: 1024 bits
    L.mjh_numa_parse_cpulist.argtypes = [C.c_char_p, C.POINTER(CpuSet)]

    def parse(s):
        cs = CpuSet()
        n = L.mjh_numa_parse_cpulist(s.encode(), C.byref(cs))
        return n, [i for i in range(1024) if (cs.bits[i // 64] >> (i % 64)) & 1]
    assert parse("0-7\n") == (8, list(range(8)))
    assert parse("0-3,64-67") == (8, [0, 1, 2, 3, 64, 65, 66, 67])
    assert parse("5") == (1, [5])
    assert parse("3,3,2-4") == (3, [2, 3, 4])
    assert parse("") == (0, [])
    assert parse("7-3")[0] == -1 and parse("0-99999")[0] == -1
    if not _gpu_present():
        assert L.mjh_device_numa_node(0) == -1
        assert L.mjh_bind_thread_to_device(0) == -1
        buf = C.create_string_buffer(256)
        L.mjh_device_placement.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        assert L.mjh_device_placement(0, buf, 256) > 0 and b"no NUMA placement" in buf.value


def test_quantization_table_presets_match_the_reference():
    """mjh_params_set_quality(.., base_idx): the nine tables of `cjpeg -quant-table N` (mjh_quant_presets.h) at several qualities,
    with and without the baseline clamp, against the base tables read back from the reference's DQT markers
    (tests/golden/quant_presets.json) scaled by jpeg_add_quant_table's rule -- and the test harness' own tables (oracle_lib) with them"""
    import json
    import oracle_lib as O
    base = json.load(open(os.path.join(ROOT, "tests", "golden", "quant_presets.json")))
    for idx in range(9):
        for quality in (1, 3, 10, 25, 50, 75, 90, 95, 100):
            scale = int(5000.0 / quality) if quality < 50 else int(200.0 - quality * 2.0)
            for baseline in (False, True):
                kw = dict(quality=quality, quant_table=idx, baseline=baseline)
                if not baseline:
                    kw["fastcrush"] = True
                pm, po = M.make_params(16, 16, **kw), O.make_params(16, 16, **kw)
                for t, name in ((0, "luma"), (1, "chroma")):
                    want = [min(max((b * scale + 50) // 100, 1), 32767) for b in base[str(idx)][name]]
                    if baseline:
                        want = [min(v, 255) for v in want]
                    assert [pm.quantval[t][k] for k in range(64)] == want, (idx, quality, baseline, name)
                    assert [po.qtbl[t][k] for k in range(64)] == want, (idx, quality, baseline, name, "oracle_lib")
