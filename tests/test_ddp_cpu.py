"""world_size-2 gloo tests of the data-parallel bucket logic (the N>1 path of bench.py) on CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from klab_multimodalmodel_amd.ddp import SegmentReducer
        segs = [("main", 0, 1000), ("main", 1000, 500), ("swin", 0, 300)]
        flats = {"main": torch.arange(1500, dtype=torch.float32) * (rank + 1), "swin": torch.full((300,), float(rank))}
        before = {k: v.clone() for k, v in flats.items()}
        red = SegmentReducer(segs, None, max_bucket_elems=256)  # forces 4 / 2 / 2 buckets
        assert [len(red.buckets(i)) for i in range(3)] == [4, 2, 2]
        assert sum(n for _, _, n in red.buckets(0)) == 1000
        red.reduce_segment(0, flats)
        # segment 1 and the other model's buffer are untouched until their own reduce
        assert torch.equal(flats["main"][1000:], before["main"][1000:]) and torch.equal(flats["swin"], before["swin"])
        red.reduce_segment(1, flats)
        red.reduce_segment(2, flats)
        red.finish()
        base = torch.arange(1500, dtype=torch.float32)
        ok = torch.allclose(flats["main"], base * (sum(range(1, world + 1)) / world)) and \
            torch.allclose(flats["swin"], torch.full((300,), (world - 1) / 2.0))
        # a model with no swin gradients (frozen): the reducer skips the missing buffer
        red.reduce_segment(2, {"main": flats["main"], "swin": None})
        red.finish()
        # per-segment join (DDP overlap_optimizer): on a backend without a comm stream it degrades to the full join
        f2 = {"main": torch.arange(1500, dtype=torch.float32) * (rank + 1), "swin": None}
        red.reduce_segment(0, f2)
        red.reduce_segment(1, f2)
        red.finish_segment(0)
        ok = ok and torch.allclose(f2["main"], base * (sum(range(1, world + 1)) / world)) and red.world_active()
        red.finish_segment(1)
        red.finish()
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_segment_reducer_averages_in_place_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_reducer_is_a_noop():
    from klab_multimodalmodel_amd.ddp import SegmentReducer
    red = SegmentReducer([("main", 0, 10)], None)
    f = {"main": torch.ones(10)}
    red.reduce_segment(0, f)
    red.finish()
    assert torch.equal(f["main"], torch.ones(10))
