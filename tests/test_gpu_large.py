"""GPU (-m gpu): BASELINE.json's configurations at their FULL sizes against the real reference (oracle/_ref/refenc, the
reference compiled from its own sources; the C restatement where that binary is absent), the one-process multi-device
pool, and the two-rank sharded run with the HIP encoder on device rank % device_count."""
import os
import socket
import sys

import numpy as np
import pytest

import mozjpeg_amd as M
import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _reference(img, kw):
    if O.have_ref():
        return O.ref_encode(img, **kw)[0], "reference"
    h, w = img.shape[:2]
    return O.encode(O.make_params(w, h, **kw), img), "port"


@pytest.mark.parametrize("name,w,h,kw", [
    ("C2", 1920, 1080, dict(quality=75, baseline=True)),
    ("metric", 3840, 2160, dict(quality=75, baseline=True)),
    ("C3", 3840, 2160, dict(quality=85, sample=(2, 2))),
    ("C5 (12-bit, the reference aborts on 12-bit + trellis: -notrellis)", 8192, 8192,
     dict(precision=12, baseline=True, notrellis=True, quality=90, sample=(1, 1), restart=1)),
    ("C5 8-bit twin with trellis", 8192, 8192, dict(baseline=True, quality=90, sample=(1, 1), restart=1)),
])
def test_baseline_configurations_at_full_size_match_the_reference(name, w, h, kw):
    twelve = kw.get("precision") == 12
    frames = np.stack([(O.synthetic_frame12 if twelve else O.synthetic_frame)(w, h, 4321 + i) for i in range(1 if w > 4000 else 2)])
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=len(frames))
    got = enc.encode_host(frames)
    enc.close()
    for i, f in enumerate(frames):
        want, kind = _reference(f, kw)
        assert got[i] == want, "%s frame %d differs from the %s (%d vs %d bytes)" % (name, i, kind, len(got[i]), len(want))


@pytest.mark.parametrize("kw", [dict(quality=100, sample=(1, 1)), dict(quality=97), dict(quality=100, sample=(1, 1), fastcrush=True)])
def test_dense_progressive_scans_take_the_direct_bit_writer(kw):
    """Noise at very high quality: the first-pass AC scans of a 2048-block chunk exceed the 16 KB LDS window of k_pp_emit
    (more than 64 bits per block on average), so the chunk is written straight into the stream; several chunks per scan,
    so the chunk offsets from the per-chunk symbol counts (k_pp_chunk_bits) are exercised across chunk borders too."""
    w, h = 768, 512                    # 6144 luma blocks = 3 chunks
    rng = np.random.RandomState(99)
    frames = np.stack([rng.randint(0, 256, (h, w, 3)).astype(np.uint8), O.synthetic_frame(w, h, 5)])
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=len(frames))
    got = enc.encode_host(frames)
    enc.close()
    for i, f in enumerate(frames):
        want, kind = _reference(f, kw)
        assert got[i] == want, "frame %d differs from the %s (%d vs %d bytes)" % (i, kind, len(got[i]), len(want))
    assert len(got[0]) * 8 > 150 * (w // 8) * (h // 8)      # the noise frame is as dense as the test needs


def test_progressive_scans_of_more_than_256_chunks():
    """A 36-Mpixel 4:4:4 frame: 565 504 blocks per component = 277 chunks of 2048 per scan, so k_pp_chunk_bits walks its
    chunk offsets in more than one round of 256 and the per-pair arrays are indexed far beyond the 4K sizes"""
    w = h = 6016
    tile = O.synthetic_frame(512, 512, 77)
    frame = np.ascontiguousarray(np.tile(tile, (12, 12, 1))[:h, :w])
    frame[::7, ::5, 1] ^= 0x55                      # (no two tiles alike)
    kw = dict(quality=60, sample=(1, 1))
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=1)
    got = enc.encode_host(frame[None])[0]
    enc.close()
    want, kind = _reference(frame, kw)
    assert got == want, "differs from the %s (%d vs %d bytes)" % (kind, len(got), len(want))


@pytest.mark.parametrize("devices", [None, [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("kw", [dict(baseline=True), dict(quality=85)])
def test_one_process_pool_deals_images_over_devices(devices, kw):
    """mjh_pool_*: one encoder + host thread per entry of `devices` (the same GPU may appear more than once, which is how
    a 1-GPU box exercises the N-device path), several steps per device, files back in image order"""
    w, h, n = 227, 149, 11
    frames = np.stack([O.synthetic_frame(w, h, 700 + i) for i in range(n)])
    pool = M.Pool(M.make_params(w, h, **kw), max_batch_per_device=2, devices=devices)
    assert pool.device_count == (len(devices) if devices else M.lib().mjh_device_count())
    got = pool.encode_host(frames)
    again = pool.encode_host(frames[:3])
    padded = np.zeros((n, h + 3, w + 5, 3), np.uint8)       # caller's row pitch / image stride
    padded[:, :h, :w] = frames
    jp, sz = M.C.POINTER(M.C.c_void_p)(), M.C.POINTER(M.C.c_size_t)()
    rc = M.lib().mjh_pool_encode_host(pool._h, padded.ctypes.data, padded.strides[1], padded.strides[0], n, M.C.byref(jp), M.C.byref(sz))
    assert rc == M.OK
    strided = [M.C.string_at(jp[i], sz[i]) for i in range(n)]
    pool.close()
    po = O.make_params(w, h, **kw)
    want = [O.encode(po, f) for f in frames]
    assert got == want
    assert again == want[:3]
    assert strided == want


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    import mozjpeg_amd as M2
    import oracle_lib as O2
    from mozjpeg_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # no collective on the data path: gloo only carries the bookkeeping
    ndev = torch.cuda.device_count()
    dev = rank % ndev
    n_images, w, h = 9, 320, 200
    mine = shard.shard_indices(n_images, rank, world)
    frames = np.stack([O2.synthetic_frame(w, h, 100 + i) for i in mine])
    enc = M2.Encoder(M2.make_params(w, h, baseline=True), max_batch=len(mine), device=dev)
    files = enc.encode_host(frames)
    enc.close()
    md5s = {i: O2.md5(f) for i, f in zip(mine, files)}
    gathered = [None] * world
    dist.all_gather_object(gathered, (dev, md5s))
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_run_with_the_hip_encoder():
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    merged = {}
    for _dev, d in gathered:
        assert not (set(d) & set(merged))
        merged.update(d)
    assert sorted(merged) == list(range(9))
    po = O.make_params(320, 200, baseline=True)
    for i in range(9):
        assert merged[i] == O.md5(O.encode(po, O.synthetic_frame(320, 200, 100 + i)))


@pytest.mark.parametrize("name,w,h,kw", [
    ("1080p sequential + the coder's trellis", 1920, 1080, dict(arithmetic=True, quality=75, baseline=True)),
    ("1080p 4:4:4 q92 sequential, restart per row, no trellis", 1920, 1080, dict(arithmetic=True, quality=92, baseline=True, sample=(1, 1), notrellis=True, restart=1)),
    ("1080p progressive with scan search (cjpeg -arithmetic)", 1920, 1080, dict(arithmetic=True, quality=85)),
    ("odd size, 4:2:2, fixed script", 1283, 727, dict(arithmetic=True, quality=60, fastcrush=True, sample=(2, 1))),
])
def test_arithmetic_coding_at_larger_sizes_matches_the_reference(name, w, h, kw):
    """cjpeg -arithmetic.  More than 64 blocks per scan row, several load batches per chain, interleaved MCUs
    with dummy blocks (odd sizes), the trellis pass' rate refresh over many iMCU rows, a batch of two different frames."""
    frames = np.stack([O.synthetic_frame(w, h, 77 + i) for i in range(2)])
    enc = M.Encoder(M.make_params(w, h, **kw), max_batch=2)
    got = enc.encode_host(frames)
    enc.close()
    for i, f in enumerate(frames):
        want, kind = _reference(f, kw)
        assert got[i] == want, "%s: frame %d differs from the %s (%d vs %d bytes)" % (name, i, kind, len(got[i]), len(want))
