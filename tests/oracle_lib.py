"""ctypes binding of the CPU oracle (oracle/libmjoracle.so) and helpers around the compiled
reference (oracle/_ref).  TEST INFRASTRUCTURE: imported only by tests/, bench.py's cpu_baseline
leg and __graft_entry__.smoke() -- never by the product package."""
import ctypes as C
import hashlib
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
MAXC, MAXS = 4, 64


class Scan(C.Structure):
    _fields_ = [("comps_in_scan", C.c_int), ("component_index", C.c_int * MAXC),
                ("Ss", C.c_int), ("Se", C.c_int), ("Ah", C.c_int), ("Al", C.c_int)]


class Params(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("input_components", C.c_int),
                ("num_components", C.c_int), ("h_samp", C.c_int * MAXC), ("v_samp", C.c_int * MAXC),
                ("quant_tbl_no", C.c_int * MAXC), ("dc_tbl_no", C.c_int * MAXC),
                ("ac_tbl_no", C.c_int * MAXC), ("component_id", C.c_int * MAXC),
                ("qtbl", (C.c_uint16 * 64) * 4), ("fastest_profile", C.c_int),
                ("optimize_coding", C.c_int), ("trellis_quant", C.c_int),
                ("trellis_quant_dc", C.c_int), ("overshoot_deringing", C.c_int),
                ("lambda_log_scale1", C.c_float), ("lambda_log_scale2", C.c_float),
                ("restart_interval", C.c_int), ("restart_in_rows", C.c_int),
                ("num_scans", C.c_int), ("scans", Scan * MAXS), ("optimize_scans", C.c_int),
                ("write_jfif", C.c_int), ("data_precision", C.c_int), ("trellis_num_loops", C.c_int),
                ("smoothing_factor", C.c_int), ("trellis_q_opt", C.c_int),
                ("trellis_eob_opt", C.c_int), ("use_scans_in_trellis", C.c_int), ("trellis_freq_split", C.c_int),
                ("rgb_output", C.c_int), ("trellis_delta_dc_weight", C.c_float), ("dc_scan_opt_mode", C.c_int),
                ("arith_code", C.c_int), ("arith_dc_L", C.c_int * 4), ("arith_dc_U", C.c_int * 4), ("arith_ac_K", C.c_int * 4),
                ("ycc_input", C.c_int), ("dct_method", C.c_int)]


class Geom(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("wib", "hib", "wpad", "hpad", "pw", "ph")]


class Taps(C.Structure):
    _fields_ = [("coef_uq", C.c_void_p * MAXC), ("coef_q0", C.c_void_p * MAXC),
                ("coef_q", C.c_void_p * MAXC), ("planes", C.c_void_p * MAXC),
                ("dc_bits", (C.c_uint8 * 17) * 4), ("dc_vals", (C.c_uint8 * 256) * 4),
                ("ac_bits", (C.c_uint8 * 17) * 4), ("ac_vals", (C.c_uint8 * 256) * 4)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "libmjoracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "port"])
        _lib = C.CDLL(path)
        _lib.mjo_encode.restype = C.c_size_t
        _lib.mjo_encode.argtypes = [C.POINTER(Params), C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.c_size_t, C.POINTER(Taps)]
        _lib.mjo_default_params.argtypes = [C.POINTER(Params)] + [C.c_int] * 10
        _lib.mjo_geometry.argtypes = [C.POINTER(Params), C.POINTER(Geom), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]
        _lib.mjo_gen_optimal_table.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_uint8),
                                               C.POINTER(C.c_uint8)]
    return _lib


def make_params(width, height, *, quality=75, baseline=False, revert=False, optimize=False,
                progressive=False, fastcrush=False, notrellis=False, notrellis_dc=False,
                noovershoot=False, sample=(2, 2), restart=None, gray=False, grayin=False,
                quant_table=-1, lambda1=None, lambda2=None, precision=8, trellis_loops=1, smooth=0, trellis_q_opt=False,
                trellis_eob_opt=False, use_scans_in_trellis=False, trellis_freq_split=0, rgb=False,
                dc_scan_opt=None, dc_ver_weight=None, arithmetic=False, arith_cond=None, scans=None, gray_sample=None, yccin=False, dct=None,
                dc_tbl=None, ac_tbl=None, no_optimize=False):
    """Same switch vocabulary as cjpeg / oracle/refenc.c.  Default (no switch) is cjpeg's default:
    max-compression profile, progressive with scan search."""
    p = Params()
    L = lib()
    per_comp = isinstance(sample[0], (tuple, list))      # ((h, v) of Y, (h, v) of Cb, (h, v) of Cr): cjpeg -sample HxV,HxV,HxV
    s0 = sample[0] if per_comp else sample
    L.mjo_default_params(C.byref(p), width, height, 1 if grayin else 3, 1 if gray else 0, quality,
                         1 if baseline else 0, 1 if revert else 0, s0[0], s0[1], quant_table)
    if per_comp and p.num_components == 3:
        for i in range(3):
            p.h_samp[i], p.v_samp[i] = sample[i]
    if gray_sample is not None and p.num_components == 1:      # (h, v) of a gray image's one component (cjpeg: 2x1 for qualities 80..89)
        p.h_samp[0], p.v_samp[0] = gray_sample
    for i in range(p.num_components):          # table numbers of the application's own (cinfo->comp_info[i].dc_tbl_no / ac_tbl_no)
        if dc_tbl is not None:
            p.dc_tbl_no[i] = dc_tbl[i]
        if ac_tbl is not None:
            p.ac_tbl_no[i] = ac_tbl[i]
    if optimize:
        p.optimize_coding = 1
    if no_optimize:
        # optimize_coding switched off by hand with the trellis on, ONE component: the reference's passes are the schedule of
        # optimize_coding and its file the same bytes (jcmaster.c:451-466, :975-1010; pinned by the *_no_optimize goldens): the
        # restatement keeps the flag on
        assert p.num_components == 1 and p.trellis_quant
    if notrellis:
        p.trellis_quant = 0
    if notrellis_dc:
        p.trellis_quant_dc = 0
    if noovershoot:
        p.overshoot_deringing = 0
    if quant_table not in (-1, 0, 3):
        # the other presets of cjpeg -quant-table N: base tables read back from the reference (tests/golden/quant_presets.json,
        # made by tests/golden/make_quant_presets.py), scaled by jpeg_add_quant_table's rule (jcparam.c:30-68)
        import json
        base = json.load(open(os.path.join(ROOT, "tests", "golden", "quant_presets.json")))[str(quant_table)]
        q = float(min(max(quality, 1), 100))
        scale = int(5000.0 / q) if q < 50 else int(200.0 - q * 2.0)
        for t, name in ((0, "luma"), (1, "chroma")):
            for k in range(64):
                v = (base[name][k] * scale + 50) // 100
                v = min(max(v, 1), 32767)
                p.qtbl[t][k] = min(v, 255) if baseline else v
    if lambda1 is not None:
        p.lambda_log_scale1 = lambda1
    if lambda2 is not None:
        p.lambda_log_scale2 = lambda2
    p.data_precision = precision
    p.trellis_num_loops = trellis_loops
    p.smoothing_factor = smooth
    p.trellis_q_opt = 1 if trellis_q_opt else 0
    p.trellis_eob_opt = 1 if trellis_eob_opt else 0
    p.use_scans_in_trellis = 1 if use_scans_in_trellis else 0
    p.trellis_freq_split = trellis_freq_split
    p.arith_code = 1 if arithmetic else 0
    p.dct_method = 1 if dct == "fast" else 0      # cjpeg -dct fast: JDCT_IFAST
    if arith_cond is not None:     # ((L, U, K) of conditioning table 0, (L, U, K) of table 1)
        for t, (lo, up, kx) in enumerate(arith_cond):
            p.arith_dc_L[t], p.arith_dc_U[t], p.arith_ac_K[t] = lo, up, kx
    if yccin and not grayin:      # in_color_space = JCS_YCbCr (refenc -yccin): the pixels are Y, Cb, Cr already
        p.ycc_input = 1
    if rgb:
        L.mjo_set_rgb_output(C.byref(p))
    if dc_ver_weight is not None:
        p.trellis_delta_dc_weight = dc_ver_weight
    if dc_scan_opt is not None:
        p.dc_scan_opt_mode = dc_scan_opt      # read by the script builders below
    if restart is not None:
        if isinstance(restart, str) and restart.lower().endswith("b"):
            p.restart_interval = int(restart[:-1])
        else:
            p.restart_in_rows = int(restart)
    if revert:
        if progressive:
            L.mjo_simple_progression(C.byref(p))
            p.optimize_coding = 1
    elif not baseline:
        if fastcrush or progressive:
            L.mjo_simple_progression(C.byref(p))
        else:
            L.mjo_search_progression(C.byref(p))
        p.optimize_coding = 1
    if scans is not None:      # cjpeg -scans: [(component indices, Ss, Se, Ah, Al), ...] replaces the script, no scan search
        p.optimize_scans = 0
        p.num_scans = len(scans)
        if not (scans[0][1] == 0 and scans[0][2] == 63):
            p.optimize_coding = 1          # a progressive script forces optimal tables (jcmaster.c:1091-1094)
        for i, (comps, ss, se, ah, al) in enumerate(scans):
            p.scans[i].comps_in_scan = len(comps)
            for j, c in enumerate(comps):
                p.scans[i].component_index[j] = c
            p.scans[i].Ss, p.scans[i].Se, p.scans[i].Ah, p.scans[i].Al = ss, se, ah, al
    return p


def geometry(p):
    g = (Geom * MAXC)()
    mpr, mr = C.c_int(), C.c_int()
    lib().mjo_geometry(C.byref(p), g, C.byref(mpr), C.byref(mr))
    return [g[i] for i in range(p.num_components)], mpr.value, mr.value


def encode(p, pixels, want_taps=False):
    """pixels: uint8 array [H, W, C] (C contiguous).  Returns bytes (and taps dict)."""
    pixels = np.ascontiguousarray(pixels, dtype=np.uint16 if p.data_precision == 12 else np.uint8)
    h, w = pixels.shape[:2]
    assert (w, h) == (p.width, p.height)
    cap = w * h * 8 + 65536
    out = np.empty(cap, np.uint8)
    taps = None
    keep = {}
    if want_taps:
        taps = Taps()
        gs, _, _ = geometry(p)
        for ci, g in enumerate(gs):
            nb = g.hpad * g.wpad * 64
            for name in ("coef_uq", "coef_q0", "coef_q"):
                a = np.zeros(nb, np.int16)
                keep[(name, ci)] = a
                getattr(taps, name)[ci] = a.ctypes.data
            a = np.zeros(g.pw * g.ph, np.uint8)
            keep[("planes", ci)] = a
            taps.planes[ci] = a.ctypes.data
    n = lib().mjo_encode(C.byref(p), pixels.ctypes.data, pixels.strides[0], out.ctypes.data, cap,
                         C.byref(taps) if taps is not None else None)
    assert n > 0, "oracle encode failed"
    data = out[:n].tobytes()
    if not want_taps:
        return data
    res = {}
    gs, _, _ = geometry(p)
    for ci, g in enumerate(gs):
        for name in ("coef_uq", "coef_q0", "coef_q"):
            res[(name, ci)] = keep[(name, ci)].reshape(g.hpad, g.wpad, 64)
        res[("planes", ci)] = keep[("planes", ci)].reshape(g.ph, g.pw)
    res["dc_bits"] = np.ctypeslib.as_array(taps.dc_bits).copy()
    res["dc_vals"] = np.ctypeslib.as_array(taps.dc_vals).copy()
    res["ac_bits"] = np.ctypeslib.as_array(taps.ac_bits).copy()
    res["ac_vals"] = np.ctypeslib.as_array(taps.ac_vals).copy()
    return data, res


def tj_plane_dims(p):
    """Plane sizes of TurboJPEG's planar YUV image for these parameters (tj3YUVPlaneWidth/Height,
    turbojpeg.c:1283-1286): component ci is PAD(W, maxh) * h_i / maxh by PAD(H, maxv) * v_i / maxv."""
    nc = p.num_components
    maxh = max(p.h_samp[i] for i in range(nc))
    maxv = max(p.v_samp[i] for i in range(nc))
    pad = lambda v, m: (v + m - 1) // m * m
    return [(pad(p.width, maxh) * p.h_samp[i] // maxh, pad(p.height, maxv) * p.v_samp[i] // maxv) for i in range(nc)]


def synthetic_planes(p, seed=7):
    """Deterministic YCbCr component planes in TurboJPEG's layout (smooth field + noise + saturated patches)."""
    rng = np.random.default_rng(seed)
    out = []
    for i, (pw, ph) in enumerate(tj_plane_dims(p)):
        y, x = np.mgrid[0:ph, 0:pw].astype(np.float64)
        f = 128 + (90 - 20 * i) * np.sin(x / (23.0 + 7 * i) + seed) * np.cos(y / (17.0 + 5 * i)) + rng.normal(0, 10, (ph, pw))
        a = np.clip(f, 0, 255).astype(np.uint8)
        a[((x.astype(np.int64) // 24 + y.astype(np.int64) // 24) % 5) == 0] = 255
        out.append(a)
    return out


def encode_planes(p, planes):
    """Oracle encode from component planes (jpeg_write_raw_data path).  planes: list of 2-D uint8 arrays."""
    planes = [np.ascontiguousarray(a, dtype=np.uint16 if p.data_precision == 12 else np.uint8) for a in planes]
    n = len(planes)
    src = (C.c_void_p * 4)(*[a.ctypes.data for a in planes] + [None] * (4 - n))
    stride = (C.c_size_t * 4)(*[a.strides[0] for a in planes] + [0] * (4 - n))
    sw = (C.c_int * 4)(*[a.shape[1] for a in planes] + [0] * (4 - n))
    sh = (C.c_int * 4)(*[a.shape[0] for a in planes] + [0] * (4 - n))
    cap = p.width * p.height * 8 + 65536
    out = np.empty(cap, np.uint8)
    L = lib()
    L.mjo_encode_planes.restype = C.c_size_t
    L.mjo_encode_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    nb = L.mjo_encode_planes(C.byref(p), src, stride, sw, sh, out.ctypes.data, cap, None)
    assert nb > 0, "oracle plane encode failed"
    return out[:nb].tobytes()


def ref_encode_planes(planes, width, height, **kw):
    """The REAL reference fed through jpeg_write_raw_data (oracle/_ref/refenc -yuvin); planes in TurboJPEG's layout."""
    with tempfile.TemporaryDirectory() as td:
        raw = os.path.join(td, "in.yuv")
        outp = os.path.join(td, "out.jpg")
        with open(raw, "wb") as f:
            for a in planes:
                f.write(np.ascontiguousarray(a, dtype=np.uint8).tobytes())
        cmd = [os.path.join(REF_DIR, "refenc")] + ref_switches(**kw) + ["-raw", str(width), str(height), "-yuvin", raw, outp]
        subprocess.check_output(cmd)
        return open(outp, "rb").read()


def transcode_params(src, **kw):
    """Parameters of a jpegtran run on a file that was encoded with `src`: jpeg_copy_critical_parameters
    (jctrans.c:70-171) = library defaults for the profile + the source's size, sampling factors and
    quantization tables, trellis off; kw = jpegtran's own switches (revert, optimize, progressive, fastcrush,
    restart) in make_params vocabulary."""
    nc = src.num_components
    p = make_params(src.width, src.height, notrellis=True, gray=(nc == 1), grayin=(nc == 1),
                    sample=(src.h_samp[0], src.v_samp[0]), precision=src.data_precision or 8, **kw)
    for t in range(4):
        for i in range(64):
            p.qtbl[t][i] = src.qtbl[t][i]
    return p


def encode_coefficients(p, coefs):
    """Oracle entropy-coding of existing quantized coefficients (jpeg_write_coefficients path).
    coefs: one int16 array [hib, wib(+pad), 64] (natural order) per component."""
    coefs = [np.ascontiguousarray(a, dtype=np.int16) for a in coefs]
    n = len(coefs)
    ptrs = (C.c_void_p * 4)(*[a.ctypes.data for a in coefs] + [None] * (4 - n))
    bpr = (C.c_size_t * 4)(*[a.shape[1] for a in coefs] + [0] * (4 - n))
    cap = p.width * p.height * 8 + 65536
    out = np.empty(cap, np.uint8)
    L = lib()
    L.mjo_encode_coefficients.restype = C.c_size_t
    L.mjo_encode_coefficients.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    nb = L.mjo_encode_coefficients(C.byref(p), ptrs, bpr, out.ctypes.data, cap)
    assert nb > 0, "oracle coefficient encode failed"
    return out[:nb].tobytes()


def real_coefficients(p, taps):
    """the real (non-dummy) blocks of an encode's final coefficients, as a transcoder would read them back"""
    gs, _, _ = geometry(p)
    return [taps[("coef_q", ci)][:g.hib, :g.wib].copy() for ci, g in enumerate(gs)]


def ref_jpegtran(jpeg, switches=(), preload=None):
    """The REAL reference jpegtran (oracle/_ref/jpegtran) on a JPEG byte string."""
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "in.jpg"), os.path.join(td, "out.jpg")
        open(inp, "wb").write(jpeg)
        env = dict(os.environ)
        if preload:
            env["LD_PRELOAD"] = preload
        r = subprocess.run([os.path.join(REF_DIR, "jpegtran")] + list(switches) + ["-outfile", outp, inp], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()
        return open(outp, "rb").read()


# ---- the compiled reference (oracle/_ref), where present -------------------------------------
def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "refenc"))


def ref_switches(**kw):
    """Translate make_params keywords into refenc/cjpeg switches."""
    sw = ["-quality", str(kw.get("quality", 75))]
    for k in ("baseline", "revert", "optimize", "progressive", "fastcrush", "notrellis",
              "noovershoot", "gray", "grayin", "rgb", "yccin"):
        if kw.get(k):
            sw.append("-" + k)
    if kw.get("notrellis_dc"):
        sw.append("-notrellis-dc")
    s = kw.get("sample", (2, 2))
    sw += ["-sample", ",".join("%dx%d" % tuple(t) for t in s) if isinstance(s[0], (tuple, list)) else "%dx%d" % s]
    if kw.get("gray_sample") is not None:
        sw += ["-graysample", "%dx%d" % tuple(kw["gray_sample"])]
    if kw.get("restart") is not None:
        sw += ["-restart", str(kw["restart"])]
    if kw.get("quant_table", -1) >= 0:
        sw += ["-quant-table", str(kw["quant_table"])]
    if kw.get("lambda1") is not None:
        sw += ["-lambda1", str(kw["lambda1"])]
    if kw.get("lambda2") is not None:
        sw += ["-lambda2", str(kw["lambda2"])]
    if kw.get("precision", 8) == 12:
        sw += ["-precision", "12"]
    if kw.get("trellis_q_opt"):
        sw.append("-trellis-q-opt")
    if kw.get("trellis_eob_opt"):
        sw.append("-trellis-eob-opt")
    if kw.get("use_scans_in_trellis"):
        sw.append("-use-scans-in-trellis")
    if kw.get("trellis_freq_split", 0):
        sw += ["-trellis-freq-split", str(kw["trellis_freq_split"])]
    if kw.get("smooth", 0):
        sw += ["-smooth", str(kw["smooth"])]
    if kw.get("dc_scan_opt") is not None:
        sw += ["-dc-scan-opt", str(kw["dc_scan_opt"])]
    if kw.get("dc_ver_weight") is not None:
        sw += ["-trellis-dc-ver-weight", repr(float(kw["dc_ver_weight"]))]
    if kw.get("trellis_loops", 1) != 1:
        sw += ["-trellis-loops", str(kw["trellis_loops"])]
    if kw.get("arithmetic"):
        sw.append("-arithmetic")
    if kw.get("dct") is not None:
        sw += ["-dct", kw["dct"]]
    if kw.get("scans") is not None:           # (refenc's own switch: cjpeg reads the script from a file)
        sw += ["-scanspec", ";".join("%s:%d-%d:%d:%d" % (",".join(str(c) for c in comps), ss, se, ah, al) for comps, ss, se, ah, al in kw["scans"])]
    if kw.get("no_optimize"):                 # (refenc's own switch: cinfo->optimize_coding = FALSE by hand)
        sw.append("-no-optimize")
    if kw.get("dc_tbl") is not None:          # (refenc's own switches: the table numbers are API-only)
        sw += ["-dctbl", ",".join(str(v) for v in kw["dc_tbl"])]
    if kw.get("ac_tbl") is not None:
        sw += ["-actbl", ",".join(str(v) for v in kw["ac_tbl"])]
    if kw.get("arith_cond") is not None:      # (refenc's own switch: the API fields cinfo->arith_dc_L / arith_dc_U / arith_ac_K have no cjpeg switch)
        sw += ["-arith-cond", ",".join(str(v) for t in kw["arith_cond"] for v in t)]
    return sw


def ref_encode(pixels, reps=1, dumpcoef=False, **kw):
    """Encode with the REAL reference library through oracle/_ref/refenc."""
    pixels = np.ascontiguousarray(pixels, dtype=np.uint16 if kw.get("precision", 8) == 12 else np.uint8)
    h, w = pixels.shape[:2]
    with tempfile.TemporaryDirectory() as td:
        raw = os.path.join(td, "in.rgb")
        outp = os.path.join(td, "out.jpg")
        pixels.tofile(raw)
        cmd = [os.path.join(REF_DIR, "refenc")] + ref_switches(**kw) + ["-raw", str(w), str(h),
                                                                        "-reps", str(reps)]
        if pixels.ndim == 2 or pixels.shape[2] == 1:
            cmd.append("-grayin")
        if dumpcoef:
            cmd += ["-dumpcoef", os.path.join(td, "coef.bin")]
        cmd += [raw, outp]
        info = subprocess.check_output(cmd).decode()
        data = open(outp, "rb").read()
        if dumpcoef:
            blob = open(os.path.join(td, "coef.bin"), "rb").read()
            coefs, off = [], 0
            while off < len(blob):
                hb, wb = np.frombuffer(blob, np.int32, 2, off)
                off += 8
                n = int(hb) * int(wb) * 64
                coefs.append(np.frombuffer(blob, np.int16, n, off).reshape(hb, wb, 64).copy())
                off += 2 * n
            return data, info, coefs
        return data, info


def md5(b):
    return hashlib.md5(b).hexdigest()


def read_ppm(path):
    with open(path, "rb") as f:
        blob = f.read()
    # P6\n W H\n 255\n
    parts = blob.split(None, 4)
    assert parts[0] == b"P6"
    w, h = int(parts[1]), int(parts[2])
    data = blob[len(blob) - w * h * 3:]
    return np.frombuffer(data, np.uint8).reshape(h, w, 3).copy()


def synthetic_frame12(width, height, seed=1234):
    """12-bit variant of the SURVEY 8d frame: same formula x16, clipped to 0..4095, uint16."""
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    s = float(seed)
    f = [np.sin(x / 97 + s) * np.cos(y / 71), np.sin((x + y) / 53), np.cos(x / 31 - y / 43)]
    amp = (100, 90, 80)
    rng = np.random.default_rng(seed)
    img = np.empty((height, width, 3), np.float64)
    for c in range(3):
        img[..., c] = (128 + amp[c] * f[c] + rng.normal(0, 12, (height, width))) * 16
    img = np.clip(img, 0, 4095).astype(np.uint16)
    tiles = ((x.astype(np.int64) // 64 + y.astype(np.int64) // 64) % 7) == 0
    img[tiles] = 4095
    return img


def synthetic_frame(width, height, seed=1234):
    """SURVEY 8d synthetic input: smooth colour fields + Gaussian noise + saturated 64x64 tiles
    (exercise deringing/clipping).  Deterministic (PCG64)."""
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    s = float(seed)
    f = [np.sin(x / 97 + s) * np.cos(y / 71), np.sin((x + y) / 53), np.cos(x / 31 - y / 43)]
    amp = (100, 90, 80)
    rng = np.random.default_rng(seed)
    img = np.empty((height, width, 3), np.float64)
    for c in range(3):
        img[..., c] = 128 + amp[c] * f[c] + rng.normal(0, 12, (height, width))
    img = np.clip(img, 0, 255).astype(np.uint8)
    tiles = ((x.astype(np.int64) // 64 + y.astype(np.int64) // 64) % 7) == 0
    img[tiles] = 255
    return img
