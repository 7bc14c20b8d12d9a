"""mozjpeg_amd -- ctypes binding of libmozjpeg_hip.so (the MI355X-native JPEG encode hot path).

This package is only a thin binding: the product is the shared library built from
mozjpeg_amd/csrc (HIP kernels for gfx950 + C++ host pipeline + C ABI, see include/mozjpeg_hip.h).
There is deliberately NO CPU fallback anywhere: if the library is missing or no GPU is visible the
calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOZJPEG_AMD_LIB") or os.path.join(_HERE, "libmozjpeg_hip.so")   # (the override: A/B runs of a differently built library, tools/gpu_*.sh)

MAX_COMPS, MAX_SCANS = 4, 64
PROFILE_MAX_COMPRESSION = 0x5D083AAD
PROFILE_FASTEST = 0x2AEA5CB4
COLOR_YCC, COLOR_NONE, COLOR_YCC_IN = 0, 1, 2
OK, EINVAL, EUNSUPPORTED, EHIP, ENOMEM, ETOOSMALL = 0, -1, -2, -3, -4, -5
TAP_PLANE, TAP_COEF_UQ, TAP_COEF_Q, TAP_COEF_Q0, TAP_HUFF_BITS, TAP_HUFF_VALS, TAP_PROG_SCAN_US = 1, 2, 3, 4, 5, 6, 7


class MjhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mozjpeg_hip error %d: %s" % (code, msg))
        self.code = code


class Scan(C.Structure):
    _fields_ = [("comps_in_scan", C.c_int), ("component_index", C.c_int * MAX_COMPS),
                ("Ss", C.c_int), ("Se", C.c_int), ("Ah", C.c_int), ("Al", C.c_int)]


class Params(C.Structure):
    """Mirror of mjh_params (include/mozjpeg_hip.h) = the cinfo fields the hot path reads."""
    _fields_ = [("image_width", C.c_int), ("image_height", C.c_int), ("input_components", C.c_int),
                ("num_components", C.c_int), ("h_samp_factor", C.c_int * MAX_COMPS),
                ("v_samp_factor", C.c_int * MAX_COMPS), ("quant_tbl_no", C.c_int * MAX_COMPS),
                ("dc_tbl_no", C.c_int * MAX_COMPS), ("ac_tbl_no", C.c_int * MAX_COMPS),
                ("component_id", C.c_int * MAX_COMPS), ("quantval", (C.c_uint16 * 64) * 4),
                ("compress_profile", C.c_int), ("optimize_coding", C.c_int), ("trellis_quant", C.c_int),
                ("trellis_quant_dc", C.c_int), ("overshoot_deringing", C.c_int),
                ("lambda_log_scale1", C.c_float), ("lambda_log_scale2", C.c_float),
                ("restart_interval", C.c_uint), ("restart_in_rows", C.c_int), ("num_scans", C.c_int),
                ("scan_info", Scan * MAX_SCANS), ("optimize_scans", C.c_int), ("write_JFIF_header", C.c_int),
                ("input_pixel_size", C.c_int), ("rgb_offset", C.c_int * 3), ("data_precision", C.c_int),
                ("trellis_num_loops", C.c_int), ("smoothing_factor", C.c_int), ("color_transform", C.c_int),
                ("dc_scan_opt_mode", C.c_int), ("trellis_delta_dc_weight", C.c_float),
                ("use_scans_in_trellis", C.c_int), ("trellis_freq_split", C.c_int),
                ("trellis_eob_opt", C.c_int), ("trellis_q_opt", C.c_int), ("arith_code", C.c_int),
                ("arith_dc_L", C.c_int * 2), ("arith_dc_U", C.c_int * 2), ("arith_ac_K", C.c_int * 2),
                ("trellis_stats_Ah", C.c_int), ("trellis_stats_Al", C.c_int), ("huff_tables_given", C.c_int),
                ("huff_bits", (C.c_uint8 * 17) * 8), ("huff_vals", (C.c_uint8 * 256) * 8), ("dct_method", C.c_int)]


class Result(C.Structure):
    """mjh_result: one finished file inside the pinned result arena"""
    _fields_ = [("offset", C.c_uint64), ("size", C.c_uint64)]


_lib = None


def lib():
    """Load libmozjpeg_hip.so; raises if it has not been built (python -m mozjpeg_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MjhError(EHIP, "%s not built: run `python -m mozjpeg_amd.build` (needs hipcc)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.mjh_last_error.restype = C.c_char_p
        L.mjh_version.restype = C.c_char_p
        L.mjh_params_size.restype = C.c_size_t
        if L.mjh_params_size() != C.sizeof(Params):     # mjh_params grows at its end between versions (include/mozjpeg_hip.h)
            raise MjhError(EINVAL, "%s (%s) has a %d-byte mjh_params, this binding a %d-byte one: rebuild"
                           % (LIB_PATH, L.mjh_version().decode(), L.mjh_params_size(), C.sizeof(Params)))
        L.mjh_device_placement.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
        L.mjh_params_defaults.argtypes = [C.POINTER(Params)] + [C.c_int] * 7
        L.mjh_params_set_quality.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_int]
        L.mjh_params_simple_progression.argtypes = [C.POINTER(Params)]
        L.mjh_params_search_progression.argtypes = [C.POINTER(Params)]
        L.mjh_encoder_create.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.mjh_encoder_destroy.argtypes = [C.c_void_p]
        L.mjh_encoder_destroy.restype = None
        L.mjh_encode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
        L.mjh_encode_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        L.mjh_encode_planes_device.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
        L.mjh_encode_planes_host.argtypes = [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int]
        L.mjh_encode_coefficients_device.argtypes = [C.c_void_p] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p]
        L.mjh_encode_coefficients_host.argtypes = [C.c_void_p] + [C.c_void_p] * 3 + [C.c_int]
        L.mjh_encoder_sync.argtypes = [C.c_void_p]
        L.mjh_set_inflight.argtypes = [C.c_void_p, C.c_int]
        L.mjh_encoder_params.argtypes = [C.c_void_p]
        L.mjh_encoder_params.restype = C.POINTER(Params)
        L.mjh_pool_create.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
        L.mjh_pool_destroy.argtypes = [C.c_void_p]
        L.mjh_pool_destroy.restype = None
        L.mjh_pool_device_count.argtypes = [C.c_void_p]
        L.mjh_pool_encode_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int,
                                           C.POINTER(C.POINTER(C.c_void_p)), C.POINTER(C.POINTER(C.c_size_t))]
        L.mjh_pool_last_error.argtypes = [C.c_void_p]
        L.mjh_pool_last_error.restype = C.c_char_p
        L.mjh_collect.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.POINTER(Result)), C.POINTER(C.c_int)]
        L.mjh_wait_input.argtypes = [C.c_void_p]
        L.mjh_host_staging.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.mjh_stage_commit.argtypes = [C.c_void_p, C.c_size_t]
        L.mjh_encode_gather.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int]
        L.mjh_host_alloc.argtypes = [C.c_size_t]
        L.mjh_host_alloc.restype = C.c_void_p
        L.mjh_host_free.argtypes = [C.c_void_p]
        L.mjh_host_free.restype = None
        L.mjh_host_register.argtypes = [C.c_void_p, C.c_size_t]
        L.mjh_host_unregister.argtypes = [C.c_void_p]
        L.mjh_get_jpeg_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
        L.mjh_get_jpeg.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mjh_get_output_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                            C.POINTER(C.c_void_p)]
        L.mjh_set_debug_taps.argtypes = [C.c_void_p, C.c_int]
        L.mjh_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.mjh_set_profiling_focus.argtypes = [C.c_void_p, C.c_char_p]
        L.mjh_read_tap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                   C.POINTER(C.c_size_t)]
        L.mjh_component_geometry.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 4
        L.mjh_get_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_char_p)),
                                           C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise MjhError(rc, lib().mjh_last_error().decode())


def make_params(width, height, *, quality=75, baseline=False, revert=False, optimize=False,
                notrellis=False, notrellis_dc=False, noovershoot=False, sample=(2, 2), gray=False,
                grayin=False, quant_table=-1, lambda1=None, lambda2=None, restart=None,
                progressive=False, fastcrush=False, precision=8, trellis_loops=1, smooth=0, rgb=False,
                dc_scan_opt=None, dc_ver_weight=None, use_scans_in_trellis=False, trellis_freq_split=0,
                trellis_eob_opt=False, trellis_q_opt=False, arithmetic=False, arith_cond=None, scans=None, gray_sample=None, yccin=False, dct=None,
                dc_tbl=None, ac_tbl=None, no_optimize=False):
    """Parameters with cjpeg's switch vocabulary (cjpeg.c:371-714).  Without `baseline` or
    `revert` this is cjpeg's default: progressive with scan search (`fastcrush`: fixed 9-scan script)."""
    p = Params()
    L = lib()
    per_comp = isinstance(sample[0], (tuple, list))      # ((h, v) of Y, (h, v) of Cb, (h, v) of Cr): cjpeg -sample HxV,HxV,HxV
    s0 = sample[0] if per_comp else sample
    _chk(L.mjh_params_defaults(C.byref(p), width, height, 1 if grayin else 3, 1 if gray else 0,
                               PROFILE_FASTEST if revert else PROFILE_MAX_COMPRESSION, s0[0], s0[1]))
    if per_comp and p.num_components == 3:
        for i in range(3):
            p.h_samp_factor[i], p.v_samp_factor[i] = sample[i]
    if gray_sample is not None and p.num_components == 1:   # (h, v) of a gray image's one component: cjpeg sets 2x1 for qualities 80..89 (rdswitch.c:566-570)
        p.h_samp_factor[0], p.v_samp_factor[0] = gray_sample
    _chk(L.mjh_params_set_quality(C.byref(p), quality, 1 if baseline else 0, quant_table))
    for i in range(p.num_components):      # table numbers of the application's own (cinfo->comp_info[i].dc_tbl_no / ac_tbl_no)
        if dc_tbl is not None:
            p.dc_tbl_no[i] = dc_tbl[i]
        if ac_tbl is not None:
            p.ac_tbl_no[i] = ac_tbl[i]
        # slots 2 / 3 are empty until the application defines them (the reference aborts on an empty slot): like oracle/refenc.c,
        # a copy of the standard table of the same parity
        for is_ac, t in ((0, p.dc_tbl_no[i]), (1, p.ac_tbl_no[i])):
            if t > 1 and (dc_tbl is not None or ac_tbl is not None):
                bits, vals, nv = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint8)(), C.c_int()
                _chk(L.mjh_std_huffman_table(is_ac, t & 1, C.byref(bits), C.byref(vals), C.byref(nv)))
                k = 2 * t + is_ac
                for j in range(17):
                    p.huff_bits[k][j] = bits[j]
                for j in range(nv.value):
                    p.huff_vals[k][j] = vals[j]
                p.huff_tables_given |= 1 << k
    if optimize:
        p.optimize_coding = 1
    if no_optimize:
        p.optimize_coding = 0               # by hand, whatever the profile set
    if notrellis:
        p.trellis_quant = 0
    if notrellis_dc:
        p.trellis_quant_dc = 0
    if noovershoot:
        p.overshoot_deringing = 0
    if lambda1 is not None:
        p.lambda_log_scale1 = lambda1
    if lambda2 is not None:
        p.lambda_log_scale2 = lambda2
    p.data_precision = precision
    p.trellis_num_loops = trellis_loops
    p.smoothing_factor = smooth
    if dc_scan_opt is not None:
        p.dc_scan_opt_mode = dc_scan_opt        # read by the script builders below
    if dc_ver_weight is not None:
        p.trellis_delta_dc_weight = dc_ver_weight
    p.use_scans_in_trellis = 1 if use_scans_in_trellis else 0
    p.trellis_freq_split = trellis_freq_split
    p.trellis_eob_opt = 1 if trellis_eob_opt else 0
    p.trellis_q_opt = 1 if trellis_q_opt else 0
    p.arith_code = 1 if arithmetic else 0
    if dct == "fast":              # cjpeg -dct fast: JDCT_IFAST
        p.dct_method = 1
    if arith_cond is not None:     # ((L, U, K) of conditioning table 0, (L, U, K) of table 1): cinfo->arith_dc_L / arith_dc_U / arith_ac_K
        for t, (lo, up, kx) in enumerate(arith_cond):
            p.arith_dc_L[t], p.arith_dc_U[t], p.arith_ac_K[t] = lo, up, kx
    if yccin and not grayin:    # in_color_space = JCS_YCbCr: the pixels are Y, Cb, Cr already (null_convert jccolor.c:479)
        p.color_transform = COLOR_YCC_IN
    if rgb:   # cjpeg -rgb: jpeg_set_colorspace(JCS_RGB) (jcparam.c:611-619): all components 1x1 / table 0, ids 'R' 'G' 'B', no JFIF
        p.color_transform = COLOR_NONE
        p.write_JFIF_header = 0
        for i, cid in enumerate(b"RGB"):
            p.component_id[i] = cid
            p.h_samp_factor[i] = p.v_samp_factor[i] = 1
            p.quant_tbl_no[i] = p.dc_tbl_no[i] = p.ac_tbl_no[i] = 0
    if restart is not None:
        if isinstance(restart, str) and restart.lower().endswith("b"):
            p.restart_interval = int(restart[:-1])
        else:
            p.restart_in_rows = int(restart)
    if revert:
        if progressive:
            _chk(L.mjh_params_simple_progression(C.byref(p)))
    elif not baseline:
        # cjpeg's default in the max-compression profile: progressive, scan search unless -fastcrush
        if fastcrush or progressive:
            _chk(L.mjh_params_simple_progression(C.byref(p)))
        else:
            _chk(L.mjh_params_search_progression(C.byref(p)))
    if scans is not None:      # cjpeg -scans: [(component indices, Ss, Se, Ah, Al), ...] replaces the script, no scan search
        p.optimize_scans = 0
        p.num_scans = len(scans)
        if not (scans[0][1] == 0 and scans[0][2] == 63):
            p.optimize_coding = 1          # a progressive script forces optimal tables (jcmaster.c:1091-1094)
        for i, (comps, ss, se, ah, al) in enumerate(scans):
            p.scan_info[i].comps_in_scan = len(comps)
            for j, c in enumerate(comps):
                p.scan_info[i].component_index[j] = c
            p.scan_info[i].Ss, p.scan_info[i].Se, p.scan_info[i].Ah, p.scan_info[i].Al = ss, se, ah, al
    return p


def pinned_empty(shape, dtype=np.uint8):
    """numpy array in pinned host memory (mjh_host_alloc): mjh_encode_host reads it by DMA, without a staging copy.
    The memory is released when the array (and every view of it) is gone."""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = lib().mjh_host_alloc(nbytes)
    if not ptr:
        raise MjhError(ENOMEM, lib().mjh_last_error().decode())
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
    import weakref
    weakref.finalize(buf, lib().mjh_host_free, ptr)
    return arr


def _as_batch(params, a):
    """[n, H, W, C] view of one image or a batch; gray images may come without the channel axis ([H, W] / [n, H, W]), which
    only the encoder's geometry can tell apart from a single [H, W, C] image"""
    h, w, c = params.image_height, params.image_width, (1 if params.input_components == 1 else a.shape[-1])
    if params.input_components == 1:
        if a.shape[-1] != 1 or a.shape[-2:] == (h, w):
            a = a[..., None]                   # no channel axis yet
    if a.ndim == 3:
        a = a[None]
    assert a.ndim == 4 and a.shape[1] == h and a.shape[2] == w, "expected [n, %d, %d, C], got %s" % (h, w, a.shape)
    return a


class Pool:
    """One process driving several GPUs (mjh_pool_*): one encoder + one host thread per device, images dealt
    round-robin (image i -> device i mod N), files returned in image order.  devices=None: every visible device."""

    def __init__(self, params, max_batch_per_device=8, devices=None):
        self._h = C.c_void_p()
        self.params = params
        arr = (C.c_int * len(devices))(*devices) if devices else None
        _chk(lib().mjh_pool_create(C.byref(params), max_batch_per_device, arr, len(devices) if devices else 0, C.byref(self._h)))

    @property
    def device_count(self):
        return lib().mjh_pool_device_count(self._h)

    def encode_host(self, frames):
        """frames: uint8 [n, H, W, C] (or uint16 for 12-bit) C-contiguous.  Returns a list of bytes."""
        frames = _as_batch(self.params, np.ascontiguousarray(frames))
        n = frames.shape[0]
        jp, sz = C.POINTER(C.c_void_p)(), C.POINTER(C.c_size_t)()
        rc = lib().mjh_pool_encode_host(self._h, frames.ctypes.data, frames.strides[1], frames.strides[0], n, C.byref(jp), C.byref(sz))
        if rc != OK:
            raise MjhError(rc, lib().mjh_pool_last_error(self._h).decode() or lib().mjh_last_error().decode())
        return [C.string_at(jp[i], sz[i]) for i in range(n)]

    def close(self):
        if self._h:
            lib().mjh_pool_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Encoder:
    """One parameter set + one GPU + device buffers for up to max_batch images."""

    def __init__(self, params, max_batch=1, device=0):
        self._h = C.c_void_p()
        self.params = params
        self.max_batch = max_batch
        _chk(lib().mjh_encoder_create(C.byref(params), max_batch, device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().mjh_encoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- encode -------------------------------------------------------------------------------
    def encode_host(self, images):
        """images: uint8 ndarray [n, H, W, C] (or [H, W, C]); returns list of bytes."""
        a = _as_batch(self.params, np.ascontiguousarray(images, dtype=np.uint16 if self.params.data_precision == 12 else np.uint8))
        n = a.shape[0]
        _chk(lib().mjh_encode_host(self._h, a.ctypes.data, a.strides[1], a.strides[0], n))
        return [self.get_jpeg(i) for i in range(n)]

    def submit_host(self, images):
        """Asynchronous mjh_encode_host: queues copy + kernels + hand-over and returns.  Pinned arrays (pinned_empty)
        are read in place and must stay untouched until wait_input()/collect(); others are staged by the library."""
        a = images if images.ndim == 4 else images[None]
        assert a.flags.c_contiguous or (a.strides[3] == a.itemsize and a.strides[2] == a.shape[3] * a.itemsize)
        _chk(lib().mjh_encode_host(self._h, a.ctypes.data, a.strides[1], a.strides[0], a.shape[0]))
        return a.shape[0]

    def wait_input(self):
        _chk(lib().mjh_wait_input(self._h))

    def collect(self, age=0, copy=True):
        """Files of the most recent submit_host batch (age 0) or of the one before it (age 1).  copy=False: memoryviews
        into the encoder's pinned result arena (reused by the second submit_host call after the batch's own)."""
        base, res, cnt = C.c_void_p(), C.POINTER(Result)(), C.c_int()
        _chk(lib().mjh_collect(self._h, age, C.byref(base), C.byref(res), C.byref(cnt)))
        out = []
        for i in range(cnt.value):
            mv = (C.c_uint8 * res[i].size).from_address(base.value + res[i].offset)
            out.append(bytes(mv) if copy else memoryview(mv))
        return out

    def encode_device_ptr(self, ptr, row_pitch, image_stride, n, stream=None):
        _chk(lib().mjh_encode_device(self._h, ptr, row_pitch, image_stride, n, stream))

    def encode_tensor(self, t, stream=None):
        """t: torch CUDA tensor [n, H, W, C] (uint8; int16/uint16 storage for 12-bit), rows contiguous.  Asynchronous.
        stream: None = the torch stream current on t's device (ordered after whatever produced t); "own" = the encoder's
        private stream (the caller guarantees t is complete, e.g. after a synchronize); or a raw hipStream_t value."""
        import torch
        p = self.params
        px = p.input_pixel_size or p.input_components
        es = 2 if p.data_precision == 12 else 1
        assert t.is_cuda and t.dim() == 4 and t.element_size() == es and t.stride(3) == 1 and t.stride(2) == t.shape[3], "layout"
        assert tuple(t.shape[1:]) == (p.image_height, p.image_width, px), "tensor %s does not match the encoder (%d x %d x %d)" % (
            tuple(t.shape), p.image_height, p.image_width, px)
        assert 1 <= t.shape[0] <= self.max_batch
        if stream is None:
            # the torch stream current on t's device; torch's default stream has handle 0, which the C ABI reads as "the
            # encoder's own stream" -- 1 asks for that stream too, but ordered behind everything queued on the null stream
            # so far, i.e. behind whatever produced `t` there
            stream = torch.cuda.current_stream(t.device).cuda_stream or 1
        elif stream == "own":
            stream = None
        self.encode_device_ptr(t.data_ptr(), t.stride(1) * es, t.stride(0) * es, t.shape[0], stream)

    # component planes in (jpeg_write_raw_data / tj3CompressFromYUVPlanes8): no colour conversion
    @staticmethod
    def _plane_args(ptrs, pitches, strides, widths, heights):
        k = len(ptrs)
        pad = lambda v, z: list(v) + [z] * (4 - k)
        return ((C.c_void_p * 4)(*pad(ptrs, None)), (C.c_size_t * 4)(*pad(pitches, 0)), (C.c_size_t * 4)(*pad(strides, 0)),
                (C.c_int * 4)(*pad(widths, 0)), (C.c_int * 4)(*pad(heights, 0)))

    def encode_planes_host(self, planes):
        """planes: one array per component, [n, h_c, w_c] or [h_c, w_c] (uint8; uint16 for 12-bit).  Returns list of bytes."""
        dt = np.uint16 if self.params.data_precision == 12 else np.uint8
        arrs = [np.ascontiguousarray(a if a.ndim == 3 else a[None], dtype=dt) for a in planes]
        n = arrs[0].shape[0]
        args = self._plane_args([a.ctypes.data for a in arrs], [a.strides[1] for a in arrs], [a.strides[0] for a in arrs],
                                [a.shape[2] for a in arrs], [a.shape[1] for a in arrs])
        _chk(lib().mjh_encode_planes_host(self._h, *args, n))
        return [self.get_jpeg(i) for i in range(n)]

    def encode_planes_tensors(self, planes, stream=None):
        """planes: one CUDA tensor [n, h_c, w_c] per component (contiguous rows).  Asynchronous."""
        for t in planes:
            assert t.is_cuda and t.dim() == 3 and t.stride(2) == 1
        es = planes[0].element_size()
        n = planes[0].shape[0]
        args = self._plane_args([t.data_ptr() for t in planes], [t.stride(1) * es for t in planes],
                                [t.stride(0) * es for t in planes], [t.shape[2] for t in planes], [t.shape[1] for t in planes])
        _chk(lib().mjh_encode_planes_device(self._h, *args, n, stream))

    # quantized coefficients in (jpeg_write_coefficients / jpegtran): entropy-coding passes only
    def encode_coefficients_host(self, coefs):
        """coefs: one int16 array per component, [n, hib, wib(+pad), 64] or [hib, wib(+pad), 64], natural order."""
        arrs = [np.ascontiguousarray(a if a.ndim == 4 else a[None], dtype=np.int16) for a in coefs]
        n, k = arrs[0].shape[0], len(arrs)
        pad = lambda v, z: list(v) + [z] * (4 - k)
        _chk(lib().mjh_encode_coefficients_host(self._h, (C.c_void_p * 4)(*pad([a.ctypes.data for a in arrs], None)),
                                                (C.c_size_t * 4)(*pad([a.shape[2] for a in arrs], 0)),
                                                (C.c_size_t * 4)(*pad([a.strides[0] for a in arrs], 0)), n))
        return [self.get_jpeg(i) for i in range(n)]

    def encode_coefficients_tensors(self, coefs, stream=None):
        """coefs: one int16 CUDA tensor [n, hib, wib(+pad), 64] per component, contiguous.  Asynchronous."""
        for t in coefs:
            assert t.is_cuda and t.is_contiguous() and t.dim() == 4 and t.element_size() == 2
        n, k = coefs[0].shape[0], len(coefs)
        pad = lambda v, z: list(v) + [z] * (4 - k)
        _chk(lib().mjh_encode_coefficients_device(self._h, (C.c_void_p * 4)(*pad([t.data_ptr() for t in coefs], None)),
                                                  (C.c_size_t * 4)(*pad([t.shape[2] for t in coefs], 0)),
                                                  (C.c_size_t * 4)(*pad([t.stride(0) * 2 for t in coefs], 0)), n, stream))

    def set_inflight(self, batches):
        """device-resident batches in flight inside the encoder: 2 (default) or 1 (mjh_set_inflight)"""
        _chk(lib().mjh_set_inflight(self._h, batches))

    def sync(self):
        _chk(lib().mjh_encoder_sync(self._h))

    def get_jpeg(self, i):
        n = C.c_size_t()
        _chk(lib().mjh_get_jpeg_size(self._h, i, C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        _chk(lib().mjh_get_jpeg(self._h, i, buf.ctypes.data, n.value, C.byref(n)))
        return buf.tobytes()

    def jpeg_size(self, i):
        n = C.c_size_t()
        _chk(lib().mjh_get_jpeg_size(self._h, i, C.byref(n)))
        return n.value

    # -- introspection ------------------------------------------------------------------------
    def set_debug_taps(self, on=True):
        _chk(lib().mjh_set_debug_taps(self._h, int(on)))

    def set_profiling(self, on=True, focus=None):
        """0 off, 1 every kernel, 2 only the interval `focus` (a name out of kernel_times(); None = keep the current choice)"""
        if focus is not None:
            _chk(lib().mjh_set_profiling_focus(self._h, focus.encode()))
        _chk(lib().mjh_set_profiling(self._h, int(on)))

    def geometry(self, c):
        v = [C.c_int() for _ in range(4)]
        _chk(lib().mjh_component_geometry(self._h, c, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)  # wib, hib, pw, ph

    def read_tap(self, what, image=0, comp=0):
        wib, hib, pw, ph = self.geometry(comp if what not in (TAP_HUFF_BITS, TAP_HUFF_VALS, TAP_PROG_SCAN_US) else 0)
        if what == TAP_PROG_SCAN_US:
            out = np.zeros((2, 72), np.uint32)
        elif what == TAP_PLANE:
            out = np.empty((ph, pw), np.uint8)
        elif what == TAP_HUFF_BITS:
            out = np.empty((4, 17), np.uint8)
        elif what == TAP_HUFF_VALS:
            out = np.empty((4, 256), np.uint8)
        else:
            out = np.empty((64, wib * hib), np.int16)
        n = C.c_size_t()
        _chk(lib().mjh_read_tap(self._h, what, image, comp, out.ctypes.data, out.nbytes, C.byref(n)))
        return out

    def kernel_times(self):
        names = C.POINTER(C.c_char_p)()
        ms = C.POINTER(C.c_float)()
        cnt = C.c_int()
        _chk(lib().mjh_get_kernel_times(self._h, C.byref(names), C.byref(ms), C.byref(cnt)))
        return [(names[i].decode(), float(ms[i])) for i in range(cnt.value)]
