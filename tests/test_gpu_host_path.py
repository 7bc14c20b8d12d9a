"""GPU (-m gpu): the host entry of the batch encoder -- pixels in host memory in, JPEG files in host memory out
(mjh_encode_host / mjh_collect / mjh_get_jpeg): asynchronous, double-buffered, pinned or pageable sources, odd row
pitches, sequential and progressive modes -- byte for byte against the CPU oracle."""
import os

import numpy as np
import pytest

import mozjpeg_amd as M
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _ref(w, h, kw, img):
    return O.encode(O.make_params(w, h, **kw), img)


@pytest.mark.parametrize("kw", [dict(baseline=True), dict(fastcrush=True), dict(revert=True)])
@pytest.mark.parametrize("pinned", [False, True])
def test_pipelined_batches_collect_previous_while_next_runs(kw, pinned):
    w, h, nb, B = 227, 149, 5, 3
    frames = np.stack([O.synthetic_frame(w, h, 300 + i) for i in range(nb * B)])
    src = frames
    if pinned:
        src = M.pinned_empty(frames.shape)
        src[...] = frames
    refs = [_ref(w, h, kw, f) for f in frames]
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=B)
    got = []
    enc.submit_host(src[0:B])
    for k in range(1, nb):
        enc.submit_host(src[k * B:(k + 1) * B])          # batch k is queued ...
        got += enc.collect(age=1)                        # ... before batch k-1 is picked up
    got += enc.collect(age=0)
    assert got == refs
    # the copying accessors read the same arena
    assert [enc.get_jpeg(i) for i in range(B)] == refs[-B:]
    enc.close()


def test_zero_copy_views_stay_valid_for_one_more_call():
    w, h = 320, 200
    frames = np.stack([O.synthetic_frame(w, h, 400 + i) for i in range(3)])
    refs = [_ref(w, h, dict(baseline=True), f) for f in frames]
    enc = M.Encoder(M.make_params(w, h, baseline=True), max_batch=1)
    enc.submit_host(frames[0:1])
    v0 = enc.collect(age=0, copy=False)
    enc.submit_host(frames[1:2])                          # one more call: batch 0's arena is still intact
    assert bytes(v0[0]) == refs[0]
    assert enc.collect(age=1) == [refs[0]]
    assert enc.collect(age=0) == [refs[1]]
    enc.close()


def test_row_pitch_and_image_stride_of_the_caller():
    """rows / images embedded in a larger host array (pitch > row bytes), pageable and pinned"""
    w, h = 121, 75
    frames = np.stack([O.synthetic_frame(w, h, 500 + i) for i in range(2)])
    refs = [_ref(w, h, dict(baseline=True), f) for f in frames]
    for pinned in (False, True):
        big = (M.pinned_empty if pinned else np.zeros)((2, h + 3, w + 5, 3), np.uint8)
        big[:, :h, :w, :] = frames
        view = big[:, :h, :w, :]
        enc = M.Encoder(M.make_params(w, h, baseline=True), max_batch=2)
        enc.submit_host(view)
        assert enc.collect() == refs
        enc.close()


def test_host_staging_buffer_of_the_encoder():
    """mjh_host_staging: the caller writes rows into the encoder's own pinned buffer (what the libjpeg drop-in does)"""
    import ctypes as C
    w, h = 200, 120
    img = O.synthetic_frame(w, h, 7)
    enc = M.Encoder(M.make_params(w, h, baseline=True), max_batch=1)
    for _ in range(3):
        buf, n = C.c_void_p(), C.c_size_t()
        assert M.lib().mjh_host_staging(enc._h, C.byref(buf), C.byref(n)) == 0
        assert n.value >= img.nbytes
        C.memmove(buf.value, img.ctypes.data, img.nbytes)
        assert M.lib().mjh_encode_host(enc._h, buf, w * 3, img.nbytes, 1) == 0
        assert enc.collect() == [_ref(w, h, dict(baseline=True), img)]
    enc.close()


def test_twelve_bit_and_progressive_search_through_the_host_entry():
    w, h = 96, 64
    img12 = O.synthetic_frame12(w, h, 9)
    kw = dict(precision=12, baseline=True, notrellis=True, quality=90, sample=(1, 1))
    enc = M.Encoder(M.make_params(w, h, **kw))
    assert enc.encode_host(img12) == [_ref(w, h, kw, img12)]
    enc.close()
    img = O.synthetic_frame(w, h, 9)
    enc = M.Encoder(M.make_params(w, h, quality=85), max_batch=2)      # progressive + scan search
    frames = np.stack([img, img[::-1].copy()])
    enc.submit_host(frames)
    enc.submit_host(frames[::-1].copy())
    a = enc.collect(age=1)
    b = enc.collect(age=0)
    assert a == [_ref(w, h, dict(quality=85), f) for f in frames] and b == a[::-1]
    enc.close()


def test_collect_errors():
    w, h = 64, 48
    enc = M.Encoder(M.make_params(w, h, baseline=True))
    with pytest.raises(M.MjhError):
        enc.collect()                       # nothing encoded yet
    enc.submit_host(O.synthetic_frame(w, h, 1))
    with pytest.raises(M.MjhError):
        enc.collect(age=1)                  # no batch before the first
    assert len(enc.collect()) == 1
    enc.close()


def test_device_entry_right_behind_a_host_batch():
    """(a) mjh_encode_device straight behind mjh_encode_host on the same encoder, nothing collected in between: the host
    batch's files are still being packed out of the single output buffers, the device batch has to wait for them and both
    come out right, for several batch sizes of one encoder."""
    import torch
    w, h = 640, 360
    kw = dict(quality=75, baseline=True)
    B = 30
    frames = np.stack([O.synthetic_frame(w, h, 700 + i) for i in range(B)])
    refs = [_ref(w, h, kw, f) for f in frames]
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=B)
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    for n in (B, B - 7, 16, 15, 1):
        enc.submit_host(frames[::-1][:n].copy())      # host batch (the frames reversed) ...
        enc.encode_tensor(d[:n], stream="own")        # ... and a device batch right behind it
        enc.sync()
        assert [enc.get_jpeg(i) for i in range(n)] == refs[:n]
        enc.submit_host(frames[:n])                   # and a host batch behind a device batch
        assert enc.collect(age=0) == refs[:n]
    # every kernel of the schedule is reported once per call
    enc.set_profiling(1)
    enc.encode_tensor(d, stream="own")
    names = [k for k, _ in enc.kernel_times()]
    assert "trellis_ac" in names and "dct_quant" in names and len(names) == len(set(names))
    enc.close()


def test_stage_commit_and_gather_of_member_encoders():
    """What the libjpeg shim does with concurrent clients: every client stages its image in the pinned buffer of an encoder
    of its own, sends the finished rows on their way while it is still writing (mjh_stage_commit), and one batch encoder
    gathers the members' images on the device and encodes them together (mjh_encode_gather)."""
    import ctypes as C
    w, h = 512, 300
    for kw in (dict(quality=75, baseline=True), dict(quality=80, fastcrush=True)):
        p = M.make_params(w, h, **kw)
        frames = [O.synthetic_frame(w, h, 900 + i) for i in range(5)]
        refs = [_ref(w, h, kw, f) for f in frames]
        members = [M.Encoder(p, max_batch=1) for _ in range(5)]
        batch = M.Encoder(p, max_batch=8)
        L = M.lib()
        for rnd in range(3):                                   # the members' double buffers flip with every gather
            order = list(range(5)) if rnd != 1 else [3, 1, 4]  # any subset, any order
            for i in order:
                buf, n = C.c_void_p(), C.c_size_t()
                assert L.mjh_host_staging(members[i]._h, C.byref(buf), C.byref(n)) == 0
                dst = np.frombuffer((C.c_uint8 * (w * h * 3)).from_address(buf.value), np.uint8).reshape(h, w, 3)
                rows = [0, 77, 200, h] if i % 2 else [0, h]    # some members commit in chunks, some not at all
                for a, b in zip(rows[:-1], rows[1:]):
                    dst[a:b] = frames[i][a:b]
                    if b < h:
                        assert L.mjh_stage_commit(members[i]._h, b * w * 3) == 0
            arr = (C.c_void_p * len(order))(*[members[i]._h for i in order])
            assert L.mjh_encode_gather(batch._h, arr, len(order)) == 0, L.mjh_last_error()
            assert batch.collect(age=0) == [refs[i] for i in order]
        # a member is still a normal encoder afterwards
        assert members[2].encode_host(frames[2]) == [refs[2]]
        for m in members:
            m.close()
        batch.close()
