"""world_size-2 gloo tests of the data-parallel bucket logic (the N>1 path of bench.py) on CPU."""
import os
import socket

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
        # KLAB_DDP_WIRE_DTYPE=bf16: gradients cross the wire in bf16 (stated departure): the mean within bf16 rounding, half the bytes
        os.environ["KLAB_DDP_WIRE_DTYPE"] = "bf16"
        try:
            red16 = SegmentReducer(segs, None, max_bucket_elems=256)
            assert red16.wire_dtype == torch.bfloat16
            f3 = {"main": (torch.arange(1500, dtype=torch.float32) * 0.37 + 1.0) * (rank + 1), "swin": None}
            red16.reduce_segment(0, f3)
            red16.reduce_segment(1, f3)
            red16.finish()
            want = (torch.arange(1500, dtype=torch.float32) * 0.37 + 1.0) * (sum(range(1, world + 1)) / world)
            ok = ok and float(((f3["main"] - want).abs() / want).max()) < 1.2e-2 and red16.stats()["bytes"] == 1500 * 2
        finally:
            del os.environ["KLAB_DDP_WIRE_DTYPE"]
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


class _FakeEngine:
    """what DistributedDataParallel / SegmentReducer read from the native engine: backward segments, per-layer buckets in the
    order backward finishes them, and the per-bucket wait (a no-op on CPU: gloo collectives are issued host-side)."""

    def __init__(self):
        # main flat buffer: [small 40 | shared 100 | dec layer0 60 | dec layer1 60 | cross kv 40] = segment 0 (300),
        #                   [small 20 | enc layer0 50 | enc layer1 50] = segment 1 (120);  swin: [small 30 | blk0 40 | blk1 40]
        self.segments = [("main", 0, 300), ("main", 300, 120), ("swin", 0, 110)]
        self.buckets = [[(200, 60), (140, 60)], [(370, 50), (320, 50)], [(70, 40), (30, 40)]]
        self.waits = []

    def bucket_wait(self, seg, i, stream):
        self.waits.append((seg, i))
        return True


def _wrapper_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import types
        from klab_multimodalmodel_amd.ddp import DistributedDataParallel

        class Fake(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.w = torch.nn.Parameter(torch.full((4,), float(rank + 1)))
                self.args = types.SimpleNamespace(image_model_train=True)
                self._engine = _FakeEngine()
                self._flat = {"main": torch.zeros(420), "swin": torch.zeros(110)}
                self._direct_grads = False
                self._segment_hook = None
                self._pending_reduce = None

        ok = True
        for overlap in (False, True):
            m = Fake()
            ddp = DistributedDataParallel(m, device_ids=None, overlap_optimizer=overlap, min_bucket_elems=50)
            ok = ok and bool(torch.equal(m.w.data, torch.full((4,), 1.0)))  # parameters broadcast from rank 0 (TORCH/ddp:864-867)
            ok = ok and m._direct_grads and m._segment_hook is not None
            # "backward": the engine fills a segment, then calls the hook -- exactly _LossFn.backward's order
            m._flat["main"].copy_(torch.arange(420, dtype=torch.float32) * (rank + 1))
            m._flat["swin"].fill_(float(rank))
            for seg in range(3):
                m._segment_hook(seg)
            if overlap:
                ok = ok and m._pending_reduce is ddp.reducer  # left to FusedAdam.step / the next forward
                ddp.join()
                ok = ok and m._pending_reduce is None
            mean_scale = sum(range(1, world + 1)) / world
            ok = ok and bool(torch.allclose(m._flat["main"], torch.arange(420, dtype=torch.float32) * mean_scale))
            ok = ok and bool(torch.allclose(m._flat["swin"], torch.full((110,), (world - 1) / 2.0)))
            plan = ddp.reducer.last_plan
            # every element of every segment is reduced exactly once ...
            cover = {"main": torch.zeros(420), "swin": torch.zeros(110)}
            for seg, model, off, n, wait in plan:
                cover[model][off:off + n] += 1
            ok = ok and bool((cover["main"] == 1).all()) and bool((cover["swin"] == 1).all())
            # ... layer buckets first, in ready order (last layer first), each behind its own event; the remainder after them
            seg0 = [(off, n, wait) for seg, _m, off, n, wait in plan if seg == 0]
            ok = ok and seg0[:2] == [(200, 60, 0), (140, 60, 1)] and all(w is None for _o, _n, w in seg0[2:])
            ok = ok and sorted((o, n) for o, n, _w in seg0[2:]) == [(0, 140), (260, 40)]
            # min_bucket_elems merges adjacent small buckets: with 100 the two 60-element layers travel as one message that waits
            # for the later event
            red2 = type(ddp.reducer)(m._engine.segments, None, engine=m._engine, min_bucket_elems=100)
            ok = ok and red2.plan(0)[0] == ("main", 140, 120, 1)
            # short_tail (what the wrapper uses): the segment's last layer bucket is never merged -- it is the only exposed message
            red3 = type(ddp.reducer)(m._engine.segments, None, engine=m._engine, min_bucket_elems=100, short_tail=True)
            ok = ok and red3.plan(0)[:2] == [("main", 200, 60, 0), ("main", 140, 60, 1)]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_ddp_wrapper_reduces_layer_buckets_then_remainder_world2():
    """DistributedDataParallel._on_segment with a fake module under gloo world-2: broadcast at construction, every gradient
    element averaged exactly once, layer buckets issued in ready order ahead of the segment remainder, overlap_optimizer's
    pending-join hand-off (ref/train.py:26,62; TORCH/ddp:1229-1250)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wrapper_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
