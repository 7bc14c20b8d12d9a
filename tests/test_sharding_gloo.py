"""CPU, world_size 2 over gloo: the N>1 path of bench.py / mozjpeg_amd.shard -- independent images
are dealt round-robin to ranks, nothing is exchanged on the data path, and the reported time is the
max over ranks.  The per-rank "encoder" here is the CPU oracle (allowed in tests)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


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
    import oracle_lib as O
    from mozjpeg_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_images = 5
    mine = shard.shard_indices(n_images, rank, world)
    md5s = {}
    for i in mine:
        img = O.synthetic_frame(64, 48, 100 + i)
        md5s[i] = O.md5(O.encode(O.make_params(64, 48, baseline=True), img))
    elapsed = 1.0 + rank          # pretend rank 1 was slower
    worst = shard.max_over_ranks(elapsed, dist, torch.device("cpu"))
    gathered = [None] * world
    dist.all_gather_object(gathered, md5s)
    if rank == 0:
        q.put((worst, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_covers_every_image_once_and_times_are_max():
    sys.path.insert(0, HERE)
    import oracle_lib as O
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    worst, gathered = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert worst == 2.0
    merged = {}
    for d in gathered:
        assert not (set(d) & set(merged)), "an image was encoded by two ranks"
        merged.update(d)
    assert sorted(merged) == list(range(5))
    for i in range(5):
        img = O.synthetic_frame(64, 48, 100 + i)
        assert merged[i] == O.md5(O.encode(O.make_params(64, 48, baseline=True), img))


def test_shard_indices_partition():
    sys.path.insert(0, os.path.dirname(HERE))
    from mozjpeg_amd import shard
    for n in (0, 1, 7, 8, 1024):
        for world in (1, 2, 3, 8):
            allidx = sorted(i for r in range(world) for i in shard.shard_indices(n, r, world))
            assert allidx == list(range(n))
            sizes = [len(shard.shard_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _worker8(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    from mozjpeg_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_indices(1024, rank, world)          # BASELINE config 4: 1024 frames over the ranks
    # what bench.py --config c4 does with its share: encode calls of at most 256 frames (128 = one call at N = 8), seeds 1234 + frame index
    calls = [(s, min(256, len(mine) - s)) for s in range(0, len(mine), 256)]
    seeds = [1234 + i for i in mine]
    worst = shard.max_over_ranks(0.5 + 0.01 * rank, dist, torch.device("cpu"))
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, calls, seeds))
    if rank == 0:
        q.put((worst, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_shard_1024_frames_exactly_once():
    """BASELINE config 4 as the driver launches it at N = 8 (one process per GPU, no collective on the data path): every one of
    the 1024 frames belongs to exactly one rank, every rank gets 128 = one encode call, the reported time is the slowest rank's"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    worst, gathered = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(worst - 0.57) < 1e-9
    seen = []
    for mine, calls, seeds in gathered:
        assert len(mine) == 128 and calls == [(0, 128)]
        assert seeds == [1234 + i for i in mine]
        seen.extend(mine)
    assert sorted(seen) == list(range(1024))
