"""GPU (-m gpu): the HIP path, called through the C ABI (ctypes -> libmozjpeg_hip.so), against the
CPU oracle on the same inputs -- stage by stage and byte for byte -- against the committed goldens
produced by the real reference, and at BASELINE.json's full sizes."""
import os

import numpy as np
import pytest

import mozjpeg_amd as M
import oracle_lib as O
from cases import CASES, CASES12, images12
from gpu_stage_check import check_case

pytestmark = pytest.mark.gpu

GPU_CASES = [(c, kw) for c, kw, on_gpu in CASES if on_gpu]


@pytest.mark.parametrize("cname,kw", GPU_CASES)
def test_every_stage_matches_oracle(cname, kw, fixture_images):
    """planes, raw DCT, pre-trellis and post-trellis coefficients, Huffman tables, final bytes"""
    for iname, img in fixture_images.items():
        assert check_case(img, kw, verbose=False), (iname, cname)


@pytest.mark.parametrize("cname,kw", GPU_CASES)
def test_bytes_match_reference_goldens(cname, kw, goldens, fixture_images):
    for iname, img in fixture_images.items():
        h, w = img.shape[:2]
        enc = M.Encoder(M.make_params(w, h, **kw))
        data = enc.encode_host(img)[0]
        enc.close()
        g = goldens["%s/%s" % (iname, cname)]
        assert (len(data), O.md5(data)) == (g["bytes"], g["md5"]), (iname, cname)


@pytest.mark.parametrize("cname,kw", [(c, kw) for c, kw, _ in CASES12])
def test_12bit_stages_and_goldens(cname, kw, goldens):
    for iname, img in images12().items():
        assert check_case(img, kw, verbose=False), (iname, cname)
        h, w = img.shape[:2]
        enc = M.Encoder(M.make_params(w, h, **kw))
        data = enc.encode_host(img)[0]
        enc.close()
        g = goldens["%s/%s" % (iname, cname)]
        assert (len(data), O.md5(data)) == (g["bytes"], g["md5"]), (iname, cname)


def test_12bit_trellis_is_refused():
    """the reference aborts for 12-bit + trellis ("Bogus buffer control mode"): no behaviour to match"""
    with pytest.raises(M.MjhError) as ei:
        M.Encoder(M.make_params(64, 64, baseline=True, precision=12))
    assert ei.value.code == M.EUNSUPPORTED


@pytest.mark.parametrize("w,h,kw", [(1920, 1080, dict(baseline=True)),              # BASELINE config 2
                                    (3840, 2160, dict(baseline=True)),              # the metric's workload
                                    (1920, 1080, dict(revert=True)),
                                    (2048, 2048, dict(baseline=True, quality=90, sample=(1, 1))),
                                    (3840, 2160, dict(quality=85)),                 # BASELINE config 3: progressive + scan search
                                    (2048, 2048, dict(precision=12, baseline=True, notrellis=True, quality=90,
                                                      sample=(1, 1), restart=1))])   # BASELINE config 5 at reduced size
def test_full_size_frames_bit_exact(w, h, kw):
    img = O.synthetic_frame12(w, h, 1234) if kw.get("precision") == 12 else O.synthetic_frame(w, h, 1234)
    enc = M.Encoder(M.make_params(w, h, **kw))
    data = enc.encode_host(img)[0]
    enc.close()
    ref = O.encode(O.make_params(w, h, **kw), img)
    assert data == ref


def test_batch_positions_are_independent_and_deterministic():
    w, h = 640, 360
    frames = np.stack([O.synthetic_frame(w, h, 50 + i) for i in range(5)])
    enc = M.Encoder(M.make_params(w, h, baseline=True), max_batch=5)
    a = enc.encode_host(frames)
    b = enc.encode_host(frames[::-1].copy())
    assert a == b[::-1]
    single = M.Encoder(M.make_params(w, h, baseline=True), max_batch=1)
    for i in range(5):
        assert single.encode_host(frames[i])[0] == a[i]
    po = O.make_params(w, h, baseline=True)
    for i in range(5):
        assert a[i] == O.encode(po, frames[i])


def test_device_resident_input_via_torch_tensor():
    import torch
    w, h = 800, 600
    frames = np.stack([O.synthetic_frame(w, h, 9 + i) for i in range(3)])
    t = torch.from_numpy(frames).cuda()
    enc = M.Encoder(M.make_params(w, h, baseline=True), max_batch=3)
    enc.encode_tensor(t)
    enc.sync()
    po = O.make_params(w, h, baseline=True)
    for i in range(3):
        assert enc.get_jpeg(i) == O.encode(po, frames[i])


@pytest.mark.parametrize("progressive", [False, True])
def test_trellis_q_opt_with_16bit_tables_writes_the_final_precision(progressive):
    """trellis_q_opt replaces the estimated entries by values <= 254 (jcmaster.c:1014-1030) and the DQT is written at the
    precision of what is left (jcmarker.c:189-254): (a) a 16-bit entry the estimate replaces -> the table ends up 8-bit, the
    file shrinks and a sequential frame is SOF0 again; (b) a 16-bit entry no block ever quantizes to non-zero stays -> the
    table stays 16-bit (SOF1).  Byte for byte against the oracle (itself pinned to the reference by the q3 / q20 goldens)."""
    from cases import images
    img = images()["testorig"]                      # (a photograph: its low frequencies do get quantized to non-zero)
    h, w = img.shape[:2]
    kw = dict(quality=75, trellis_q_opt=True)       # (tables of quality 75 fit 8 bits; one entry is raised below)
    kw.update(dict(fastcrush=True) if progressive else dict(baseline=True))
    for nat_index, value, want_16 in ((1, 260, False), (63, 900, True)):
        po, pg = O.make_params(w, h, **kw), M.make_params(w, h, **kw)
        po.qtbl[0][nat_index] = value
        pg.quantval[0][nat_index] = value
        ref = O.encode(po, img)
        enc = M.Encoder(pg, max_batch=3)
        got = enc.encode_host(np.stack([img, img[::-1].copy(), img]))
        assert got[0] == ref and got[2] == ref
        i = ref.find(b"\xff\xdb")
        dqt_len = int.from_bytes(ref[i + 2:i + 4], "big")
        assert dqt_len == (2 + 129 + 65 if want_16 else 2 + 65 + 65), dqt_len
        if not progressive:
            assert (b"\xff\xc1" in ref[:i + dqt_len + 12]) == want_16
        enc.close()


@pytest.mark.parametrize("kw", [dict(baseline=True), dict(fastcrush=True), dict(baseline=True, gray=True),
                                dict(baseline=True, restart=2)])
def test_trellis_q_opt_encoder_can_be_reused(kw):
    """trellis_q_opt re-estimates the per-image quantization tables during an encode (jcmaster.c:1014-1030); the next call of
    the same encoder must start from the parameters' tables again (a new jpeg_start_compress in the reference): the
    second and third encode of one encoder, with different images and batch sizes, against the oracle"""
    from cases import images
    imgs = images()
    base = imgs["testorig"]
    h, w = base.shape[:2]
    kw = dict(kw, quality=75, trellis_q_opt=True)
    po, pg = O.make_params(w, h, **kw), M.make_params(w, h, **kw)
    a, b, c = base, base[::-1].copy(), np.roll(base, 37, axis=1).copy()
    want = {id(x): O.encode(po, x) for x in (a, b, c)}
    enc = M.Encoder(pg, max_batch=3)
    for batch in ([a], [b, a], [c, b, a], [a]):
        got = enc.encode_host(np.stack(batch))
        for g, x in zip(got, batch):
            assert g == want[id(x)]
    enc.close()


def test_tensor_encode_is_ordered_behind_the_default_stream_producer():
    """encode_tensor(stream=None) straight behind a producer on torch's default (null) stream, no synchronize in between:
    the encode has to see the finished pixels (the null stream has handle 0, which the ABI reads as "own stream"; the
    binding passes hipStreamLegacy instead)"""
    import torch
    w, h = 1920, 1080
    kw = dict(quality=75, baseline=True)
    frames = np.stack([O.synthetic_frame(w, h, 40 + i) for i in range(4)])
    want = [O.encode(O.make_params(w, h, **kw), f) for f in frames]
    src = torch.from_numpy(frames).cuda()
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=4)
    t = torch.zeros_like(src)
    a = torch.randn(4096, 4096, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        for _ in range(20):
            a = (a @ a).clamp_(-1, 1)          # ~tens of ms of work queued on the default stream ...
        t.copy_(src)                           # ... then the pixels, still queued behind it
        enc.encode_tensor(t)                   # no synchronize: must be ordered behind the copy
        enc.sync()
        assert [enc.get_jpeg(i) for i in range(4)] == want
        t.zero_()
        torch.cuda.synchronize()


@pytest.mark.gpu
def test_jpeg_structure_properties():
    """size-independent properties: SOI/EOI, marker walk, no unstuffed 0xFF in the scan"""
    w, h = 1024, 768
    img = O.synthetic_frame(w, h, 3)
    enc = M.Encoder(M.make_params(w, h, baseline=True))
    d = enc.encode_host(img)[0]
    assert d[:2] == b"\xff\xd8" and d[-2:] == b"\xff\xd9"
    pos, seen = 2, []
    while True:
        assert d[pos] == 0xFF
        m = d[pos + 1]
        seen.append(m)
        ln = (d[pos + 2] << 8) | d[pos + 3]
        pos += 2 + ln
        if m == 0xDA:
            break
    assert seen == [0xE0, 0xDB, 0xC0, 0xC4, 0xDA]     # APP0, one DQT, SOF0, one DHT, SOS (jcmarker.c:189,293)
    scan = d[pos:-2]
    i = scan.find(b"\xff")
    while i >= 0:
        assert scan[i + 1] == 0, "unstuffed 0xFF inside entropy-coded data"
        i = scan.find(b"\xff", i + 2)


def test_extreme_inputs():
    for img in (np.zeros((40, 40, 3), np.uint8), np.full((40, 40, 3), 255, np.uint8),
                np.random.default_rng(0).integers(0, 256, (64, 64, 3), dtype=np.uint8)):
        for kw in (dict(baseline=True), dict(baseline=True, quality=100), dict(baseline=True, quality=1)):
            h, w = img.shape[:2]
            enc = M.Encoder(M.make_params(w, h, **kw))
            assert enc.encode_host(img)[0] == O.encode(O.make_params(w, h, **kw), img), kw


# ---- component planes in: jpeg_write_raw_data / tj3CompressFromYUVPlanes8 -------------------
def _plane_goldens():
    import json
    from cases import HERE
    return json.load(open(os.path.join(HERE, "goldens_planes.json")))


@pytest.mark.gpu
def test_plane_input_matches_reference_goldens_and_oracle():
    from cases import PLANE_CASES
    g = _plane_goldens()
    for cname, w, h, kw in PLANE_CASES:
        po = O.make_params(w, h, **kw)
        planes = O.synthetic_planes(po, 7)
        enc = M.Encoder(M.make_params(w, h, **kw), max_batch=1)
        data = enc.encode_planes_host(planes)[0]
        enc.close()
        assert (len(data), O.md5(data)) == (g[cname]["bytes"], g[cname]["md5"]), cname
        assert data == O.encode_planes(po, planes), cname


@pytest.mark.gpu
def test_plane_input_device_batch_and_full_size_planes():
    """device-resident planes, a batch of 3, 1080p 4:2:0 trellis; also planes already padded to whole blocks"""
    import torch
    w, h, kw = 1920, 1080, dict(baseline=True)
    po = O.make_params(w, h, **kw)
    sets = [O.synthetic_planes(po, 20 + i) for i in range(3)]
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=3)
    ts = [torch.from_numpy(np.stack([s[c] for s in sets])).cuda() for c in range(3)]
    enc.encode_planes_tensors(ts)
    enc.sync()
    for i in range(3):
        assert enc.get_jpeg(i) == O.encode_planes(po, sets[i]), i
    # same image with the planes padded by replication to width_in_blocks*8 x height_in_blocks*8 (what a direct
    # jpeg_write_raw_data caller supplies): identical file
    padded = []
    for c in range(3):
        wib, hib, pw, ph = enc.geometry(c)
        a = sets[0][c]
        padded.append(np.pad(a, ((0, max(0, ph - a.shape[0])), (0, max(0, pw - a.shape[1]))), mode="edge"))
    assert enc.encode_planes_host(padded)[0] == O.encode_planes(po, sets[0])
    enc.close()


@pytest.mark.gpu
def test_plane_input_of_pixel_path_planes_gives_pixel_path_file():
    img = O.synthetic_frame(250, 187, 3)
    kw = dict(quality=85)
    enc = M.Encoder(M.make_params(250, 187, **kw), max_batch=1)
    enc.set_debug_taps(True)
    ref = enc.encode_host(img)[0]
    planes = [enc.read_tap(M.TAP_PLANE, 0, c) for c in range(3)]
    assert enc.encode_planes_host(planes)[0] == ref
    enc.close()


# ---- quantized coefficients in: jpeg_write_coefficients / jpegtran -------------------------
@pytest.mark.gpu
def test_coefficient_input_matches_jpegtran_goldens_and_oracle(fixture_images):
    import json
    from cases import HERE, TRANSCODE_CASES
    g = json.load(open(os.path.join(HERE, "goldens_transcode.json")))
    for cname, iname, src_kw, _switches, kw in TRANSCODE_CASES:
        img = fixture_images[iname]
        h, w = img.shape[:2]
        ps = O.make_params(w, h, **src_kw)
        _src, taps = O.encode(ps, img, want_taps=True)
        coefs = O.real_coefficients(ps, taps)
        pt = O.transcode_params(ps, **kw)
        mp = M.make_params(w, h, notrellis=True, gray=(ps.num_components == 1), grayin=(ps.num_components == 1),
                           sample=(ps.h_samp[0], ps.v_samp[0]), **kw)
        for t in range(4):
            for i in range(64):
                mp.quantval[t][i] = ps.qtbl[t][i]
        enc = M.Encoder(mp, max_batch=1)
        data = enc.encode_coefficients_host(coefs)[0]
        enc.close()
        assert (len(data), O.md5(data)) == (g[cname]["bytes"], g[cname]["md5"]), cname
        assert data == O.encode_coefficients(pt, coefs), cname


@pytest.mark.gpu
def test_coefficient_input_device_batch_1080p_and_trellis_is_refused():
    import torch
    w, h = 1920, 1080
    ps = O.make_params(w, h, baseline=True, notrellis=True)
    sets = []
    for i in range(2):
        _d, taps = O.encode(ps, O.synthetic_frame(w, h, 40 + i), want_taps=True)
        sets.append(O.real_coefficients(ps, taps))
    for kw in (dict(), dict(revert=True)):
        pt = O.transcode_params(ps, **kw)
        mp = M.make_params(w, h, notrellis=True, **kw)
        for t in range(4):
            for i in range(64):
                mp.quantval[t][i] = ps.qtbl[t][i]
        enc = M.Encoder(mp, max_batch=2)
        ts = [torch.from_numpy(np.stack([s[c] for s in sets])).cuda() for c in range(3)]
        enc.encode_coefficients_tensors(ts)
        enc.sync()
        for i in range(2):
            assert enc.get_jpeg(i) == O.encode_coefficients(pt, sets[i]), (kw, i)
        enc.close()
    enc = M.Encoder(M.make_params(w, h, baseline=True), max_batch=1)   # trellis on
    with pytest.raises(M.MjhError):
        enc.encode_coefficients_host(sets[0])
    enc.close()


@pytest.mark.gpu
def test_coefficient_input_out_of_range_is_an_error():
    """jpegtran on untrusted files: an AC coefficient beyond MAX_COEF_BITS or a DC difference beyond MAX_COEF_BITS + 1 has
    no Huffman symbol; the reference raises JERR_BAD_DCT_COEF (jchuff.c:489,596,624), the GPU path reports it as well"""
    w, h = 64, 48
    mp = M.make_params(w, h, baseline=True, notrellis=True)
    good = [np.zeros((6, 8, 64), np.int16), np.zeros((3, 4, 64), np.int16), np.zeros((3, 4, 64), np.int16)]
    good[0][2, 3, 5] = 1023
    good[0][0, 0, 0] = 1023
    good[0][0, 1, 0] = -1024            # DC difference 2047: 11 bits, the largest that exists
    enc = M.Encoder(mp)
    assert len(enc.encode_coefficients_host(good)[0]) > 100
    for k, v in ((5, 1024), (63, -2000)):
        bad = [a.copy() for a in good]
        bad[1][1, 2, k] = v
        with pytest.raises(M.MjhError) as ei:
            enc.encode_coefficients_host(bad)
        assert "JERR_BAD_DCT_COEF" in str(ei.value)
    bad = [a.copy() for a in good]
    bad[0][0, 1, 0] = -1100              # DC difference 2123: 12 bits
    with pytest.raises(M.MjhError) as ei:
        enc.encode_coefficients_host(bad)
    assert "JERR_BAD_DCT_COEF" in str(ei.value)
    assert len(enc.encode_coefficients_host(good)[0]) > 100   # the encoder is usable afterwards
    enc.close()


@pytest.mark.gpu
def test_scan_beyond_the_32bit_offset_range_is_an_error_not_a_wrapped_file():
    """bit offsets inside one scan are 32-bit; a scan of more than 2^32 bits (a > 512 MB JPEG) must be reported"""
    w = h = 16384
    rng = np.random.default_rng(3)
    tile = rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)
    img = np.tile(tile, (h // 512, w // 512, 1))          # incompressible at q100 4:4:4: several bits per coefficient
    enc = M.Encoder(M.make_params(w, h, quality=100, baseline=True, notrellis=True, sample=(1, 1)), max_batch=1)
    with pytest.raises(M.MjhError) as ei:
        enc.encode_host(img)
    assert "32-bit" in str(ei.value)
    enc.close()


@pytest.mark.gpu
def test_two_host_threads_with_their_own_encoders():
    """SURVEY 8b threading contract: distinct compress objects may run concurrently (one encoder per thread here;
    ctypes drops the GIL during the calls, mjh_last_error is thread-local)"""
    import threading
    jobs = [(O.synthetic_frame(640, 480, 11), dict(baseline=True)), (O.synthetic_frame(517, 389, 12), dict(quality=85)),
            (O.synthetic_frame(640, 480, 13), dict(revert=True, sample=(2, 1))), (O.synthetic_frame(333, 222, 14), dict(fastcrush=True))]
    results = [None] * len(jobs)

    def work(i):
        img, kw = jobs[i]
        h, w = img.shape[:2]
        enc = M.Encoder(M.make_params(w, h, **kw), max_batch=1)
        outs = [enc.encode_host(img)[0] for _ in range(5)]
        enc.close()
        results[i] = outs

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for (img, kw), outs in zip(jobs, results):
        h, w = img.shape[:2]
        ref = O.encode(O.make_params(w, h, **kw), img)
        assert outs is not None and all(o == ref for o in outs), kw


@pytest.mark.gpu
def test_progressive_eob_run_longer_than_32767_blocks():
    """a flat image with one busy corner: EOB runs of more than 0x7FFF blocks force an emission inside the run
    (jcphuff.c:719) -- the parallel AC-first encode hands such scans to the sequential walk"""
    w = h = 2048                                   # 65 536 luma blocks
    img = np.full((h, w, 3), 97, np.uint8)
    img[-64:, -64:] = np.random.default_rng(5).integers(0, 256, (64, 64, 3), dtype=np.uint8)
    for kw in (dict(fastcrush=True, notrellis=True), dict(fastcrush=True, notrellis=True, sample=(1, 1))):
        enc = M.Encoder(M.make_params(w, h, **kw))
        out = enc.encode_host(img)[0]
        enc.close()
        assert out == O.encode(O.make_params(w, h, **kw), img), kw


@pytest.mark.parametrize("quality,sample", [(75, (2, 2)), (88, (2, 2)), (92, (1, 1)), (97, (2, 1))])
def test_every_first_tier_capacity_of_the_ac_trellis_gives_the_same_file(quality, sample):
    """MJH_TRELLIS_VARIANT pins the queue capacity of the AC trellis' first tier (16 / 20 / 24 / 32 / 48 records, the last two
    only in the tile-sorted kernel): whatever it is -- nearly every block deferred at 16 records and q97, none at 48 and q75 --
    the file is the oracle's; so is the adaptive choice, from its starting point and after it has seen a batch.  Sequential
    and progressive mode (the progressive trellis passes run the same kernel)."""
    w, h = 600, 424
    rng = np.random.default_rng(quality)
    img = O.synthetic_frame(w, h, 60 + quality)
    img[100:300, 200:500] = rng.integers(0, 256, (200, 300, 3), dtype=np.uint8)      # a busy region: blocks with 40+ records
    for kw in (dict(quality=quality, baseline=True, sample=sample), dict(quality=quality, fastcrush=True, sample=sample)):
        want = O.encode(O.make_params(w, h, **kw), img)
        for variant in ("0", "1", "2", "3", "4", None):
            if variant is None:
                os.environ.pop("MJH_TRELLIS_VARIANT", None)
            else:
                os.environ["MJH_TRELLIS_VARIANT"] = variant
            try:
                enc = M.Encoder(M.make_params(w, h, **kw), max_batch=3)
            finally:
                os.environ.pop("MJH_TRELLIS_VARIANT", None)
            frames = np.stack([img, img[::-1].copy(), img])
            for rnd in range(3 if variant is None else 1):                             # adaptive: the choice moves between batches
                got = enc.encode_host(frames)
                assert got[0] == want and got[2] == want, (kw, variant, rnd)
            enc.close()


@pytest.mark.parametrize("w,h", [(65500, 9), (9, 65500), (65500, 1), (1, 65500)])
def test_frames_at_the_largest_dimension_jpeg_allows(w, h):
    """JPEG_MAX_DIMENSION (jmorecfg.h:215) is 65500: one-MCU-high frames of that width and one-MCU-wide frames of that height
    (8188 blocks in a row / 4094 iMCU rows of one MCU; plane heights of 65504 rows along grid.y) through the sequential
    trellis path, cjpeg's default progressive mode with its scan search, the plain libjpeg mode, restart markers every MCU row
    in 4:4:4 and, for the flat shapes, the arithmetic coder -- byte for byte what the oracle writes."""
    img = O.synthetic_frame(max(w, 64), max(h, 64), 5)[:h, :w].copy()
    kws = [dict(quality=75, baseline=True), dict(quality=85), dict(revert=True), dict(quality=75, baseline=True, sample=(1, 1), restart=1)]
    if min(w, h) == 1:
        kws.append(dict(arithmetic=True, quality=75))
    for kw in kws:
        want = O.encode(O.make_params(w, h, **kw), img)
        enc = M.Encoder(M.make_params(w, h, **kw), max_batch=2)
        got = enc.encode_host(np.stack([img, img[::-1, ::-1].copy()]))
        enc.close()
        assert got[0] == want, (w, h, kw)


@pytest.mark.parametrize("quality,sample", [(85, (2, 2)), (75, (1, 1)), (95, (2, 1)), (40, (2, 2))])
def test_skiplow_walks_of_the_scan_search_give_the_same_file(quality, sample):
    """The statistics and emit kernels of the first-pass AC scans (k_pp_stats<true, 1>, k_pp_emit) take the non-zeros below the band
    out of a block's mask up front instead of visiting and dropping them (pp_band_nonzeros<true>), and skip bursts no lane
    needs.  The oracle's files for the scan search (cjpeg's default), the fixed nine-scan script, a notrellis search (dense
    coefficient planes), grey and subsampled frames, frames whose upper bands are empty (flat) or full (noise), and more
    than one chunk of 2048 blocks per component.  (Round 4 wrote these walks as an opt-in variant; round 5 timed them on the
    chip -- C3 24.79 -> 23.82 ms per 32 frames, profiles/r05a_optin_kernels_ab.md -- and made them the only form.)"""
    w, h = 616, 440            # 77 x 55 = 4235 luma blocks at 1x1 sampling: three chunks, the last one ragged
    rng = np.random.default_rng(1000 + quality)
    img = O.synthetic_frame(w, h, 7 + quality)
    img[40:200, 300:560] = rng.integers(0, 256, (160, 260, 3), dtype=np.uint8)     # noise: every band of these blocks is busy
    img[260:420, 20:280] = 131                                                      # flat: nothing but DC
    flat = np.full_like(img, 77)
    frames = np.stack([img, flat, img[::-1, ::-1].copy()])
    for kw in (dict(quality=quality, sample=sample), dict(quality=quality, sample=sample, fastcrush=True), dict(quality=quality, gray=True),
               dict(quality=quality, sample=sample, notrellis=True), dict(quality=quality, sample=sample, dc_scan_opt=2, trellis_eob_opt=True)):
        want = [O.encode(O.make_params(w, h, **kw), f) for f in frames]
        enc = M.Encoder(M.make_params(w, h, **kw), max_batch=3)
        for rnd in range(2):
            got = enc.encode_host(frames)
            assert [bytes(g) for g in got] == want, (kw, rnd)
        enc.close()


def _front_image(w, h, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "noise":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img = O.synthetic_frame(max(w, 8), max(h, 8), seed)[:h, :w].copy()
    for _ in range(6):       # saturated patches: the deringing walk, also across block and strip borders
        y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
        img[y:y + int(rng.integers(1, 20)), x:x + int(rng.integers(1, 40))] = 255
    return img


@pytest.mark.parametrize("w,h,kw", [
    (8, 8, dict(baseline=True)), (16, 16, dict(baseline=True)), (24, 9, dict()), (512, 33, dict(baseline=True)),
    (520, 40, dict(baseline=True, quality=90)),              # 32.5 MCUs: an odd number of luma block columns
    (1024, 16, dict(baseline=True)), (1032, 24, dict(fastcrush=True, sample=(2, 1))), (1040, 57, dict(baseline=True, restart=1)),
    (2056, 17, dict(baseline=True, notrellis=True)), (2304, 72, dict(quality=60)), (1048, 136, dict(baseline=True, quality=40, trellis_loops=2)),
    (640, 200, dict(revert=True)), (1104, 50, dict(arithmetic=True, baseline=True, sample=(2, 1))), (776, 95, dict(baseline=True, trellis_q_opt=True)),
])
def test_vector_colour_kernel_at_its_edges(w, h, kw):
    """k_color_vec (packed RGB / BGR rows of a multiple of 8 bytes, 2:1 horizontal chroma): widths of one to a few hundred 8-pixel
    runs, heights around the MCU rows, saturated patches and noise, RGB and BGR byte order -- bytes against the oracle, and stage by
    stage (plane tap included).  (Round 5 also built these sizes for a one-kernel front end, pixel rows -> coefficients through an
    LDS tile; it tied with the two kernels on the metric and lost on 1080p frames -- profiles/r05q_front_end_fusion_ab.md -- and
    is not in the tree.)"""
    for kind in ("photo", "noise"):
        img = _front_image(w, h, 100 + w + h, kind)
        want = O.encode(O.make_params(w, h, **kw), img)
        enc = M.Encoder(M.make_params(w, h, **kw), max_batch=3)
        got = enc.encode_host(np.stack([img, img[::-1].copy(), img]))
        enc.close()
        assert got[0] == want and got[2] == want, (w, h, kw, kind)
        assert got[1] == O.encode(O.make_params(w, h, **kw), img[::-1].copy()), (w, h, kw, kind)
        p = M.make_params(w, h, **kw)                     # BGR rows of the same picture
        p.rgb_offset[0], p.rgb_offset[1], p.rgb_offset[2] = 2, 1, 0
        enc = M.Encoder(p, max_batch=1)
        assert enc.encode_host(np.ascontiguousarray(img[:, :, ::-1])[None])[0] == want, (w, h, kw, kind, "bgr")
        enc.close()
    assert check_case(img, kw, verbose=False), (w, h, kw)


def test_quantization_steps_of_8192_and_more_follow_the_reference():
    """The reference's 8-bit FDCT manager takes `quantval << 3` as a UINT16 (compute_reciprocal, jcdctmgr.c:182, :278-282): steps of
    8192 and more divide by (8 q) mod 65536 -- quality 1 has ten such steps in table 0 (goldens `q1_wrapped_divisors*`) -- and a step
    of exactly 8192 / 16384 / 24576 makes it divide by zero: refused for pixel input, with the reason"""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    for q in (8191, 8193, 9000, 20000, 32767):              # around and beyond the wrap: bytes equal the oracle's (= the reference's rule)
        po, pg = O.make_params(64, 48, quality=1, notrellis=True), M.make_params(64, 48, quality=1, notrellis=True)
        for t in range(2):
            for k in (40, 50, 63):
                po.qtbl[t][k] = q
                pg.quantval[t][k] = q
        enc = M.Encoder(pg)
        assert enc.encode_host(img)[0] == O.encode(po, img), q
        enc.close()
    pg = M.make_params(64, 48, quality=1, notrellis=True)
    pg.quantval[0][63] = 8192
    enc = M.Encoder(pg)
    with pytest.raises(Exception, match="divides by zero"):
        enc.encode_host(img)
    enc.close()
