"""Data parallelism for the native MyModel: the reference's only parallel strategy
(`DDP(model, device_ids=[device_id])`, ref/train.py:26), re-done for MI355X.

The engine writes gradients into flat fp32 buffers whose layout follows the backward order
(segment 0 = LM head + decoder + tied embedding, 1 = encoder, 2 = Swin).  When a segment's kernels
have been enqueued, its slice of the flat buffer is all-reduced IN PLACE (no bucket copies) by RCCL
on a side HIP stream, overlapping the next segment's backward; the last reduce is joined before
`optimizer.step()`.  One process per GPU; `backend="nccl"` is RCCL over xGMI on ROCm.
Semantics kept from torch DDP: parameters broadcast from rank 0 at construction, gradients averaged
over ranks, every micro-step reduces (the reference never uses no_sync).
"""
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import nn


class SegmentReducer:
    """All-reduce (mean) of slices of flat gradient buffers, asynchronously, in the order the backward finishes them.

    A segment is reduced as (1) its layer buckets -- contiguous ranges the engine finishes one after the other, each behind its
    own engine-owned event, merged until a message is at least `min_bucket_elems` long (xGMI: per-peer slices of a direct
    reduce-scatter must stay bandwidth-bound) -- and (2) the remainder of the segment (norm weights, the tied
    embedding, the cross-attention k|v block), which is final when the segment's backward has been enqueued.  The host issues
    all of a segment's collectives right after enqueueing it; on the GPU each one starts when ITS event fires, i.e. while
    the later layers of the same segment are still running (torch DDP's bucket-ready overlap, TORCH/ddp:1229-1250).

    Device-agnostic so that the bucket logic is testable with gloo on CPU; on GPU the collectives run on `comm_stream`."""

    def __init__(self, segments: List[Tuple[str, int, int]], process_group=None, max_bucket_elems: int = 64 << 20,
                 engine=None, min_bucket_elems: int = 6 << 20, short_tail: bool = False):
        self.segments = segments
        # short_tail: the LAST layer bucket of a segment always travels alone.  Only the all-reduce issued after the segment's last
        # layer is exposed (everything earlier runs underneath the layers still in backward), so that message should be as small
        # as the layer granularity allows instead of whatever the min_bucket merge left over.
        self.short_tail = bool(short_tail)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.max_bucket = max_bucket_elems
        self.min_bucket = min_bucket_elems
        self.engine = engine  # provides .buckets[seg] = [(off, len)] in ready order and .bucket_wait(seg, i, stream)
        self.comm_stream: Optional[torch.cuda.Stream] = None
        self._works = []
        self._seg_events = {}  # segment -> event recorded on the comm stream behind that segment's last all-reduce
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self._avg_op = dist.ReduceOp.AVG if backend == "nccl" else None
        self._calls = 0
        self._bytes = 0
        # KLAB_DDP_WIRE_DTYPE=bf16 (or wire_dtype=torch.bfloat16): gradients cross the wire in bf16 -- half the bytes on the xGMI
        # links and half the time RCCL's kernels share the chip with the backward -- and are widened back to fp32 on arrival.
        # A STATED DEPARTURE from the reference (torch DDP averages in fp32, TORCH/ddp:1229-1250): each value is rounded to 8
        # significant bits once before the sum and the sum itself is kept in bf16 by RCCL.  Off by default.
        wd = os.environ.get("KLAB_DDP_WIRE_DTYPE", "fp32").lower()
        self.wire_dtype = torch.bfloat16 if wd in ("bf16", "bfloat16") else None
        self.last_plan = []  # [(segment, model, offset, length, waits_on_bucket | None)] of the last reduce_segment calls (tests)

    def reset_stats(self):
        self._calls = self._bytes = 0

    def stats(self):
        return {"calls": self._calls, "bytes": self._bytes}

    def world_active(self):
        return self.world > 1 or (dist.is_initialized() and os.environ.get("KLAB_DDP_FORCE_COLLECTIVE") == "1")

    def _cut(self, model, off, ln, wait):
        out = []
        while ln > 0:
            n = min(ln, self.max_bucket)
            out.append((model, off, n, wait))
            off += n
            ln -= n
        return out

    def plan(self, seg: int):
        """[(model, offset, length, wait)] messages of a segment in issue order.  wait = index of the engine bucket whose event
        the message waits for, or None = wait for the whole segment (the caller's stream)."""
        model, off, ln = self.segments[seg]
        layer = list(self.engine.buckets[seg]) if self.engine is not None and seg < len(self.engine.buckets) else []
        msgs, covered = [], []
        i = 0
        while i < len(layer):  # merge consecutive ready buckets (they are adjacent, descending in memory) up to min_bucket
            lo, hi, j = layer[i][0], layer[i][0] + layer[i][1], i
            while (hi - lo < self.min_bucket and j + 1 < len(layer) and layer[j + 1][0] + layer[j + 1][1] == lo
                   and not (self.short_tail and j + 1 == len(layer) - 1 and len(layer) > 1)):
                j += 1
                lo = layer[j][0]
            if not (off <= lo and hi <= off + ln):
                raise ValueError("engine bucket outside its segment")
            msgs += self._cut(model, lo, hi - lo, j)
            covered.append((lo, hi))
            i = j + 1
        covered.sort()
        cur = off
        rest = []
        for lo, hi in covered:
            if lo > cur:
                rest += self._cut(model, cur, lo - cur, None)
            cur = max(cur, hi)
        if off + ln > cur:
            rest += self._cut(model, cur, off + ln - cur, None)
        return msgs + rest

    def buckets(self, seg: int):
        """(model, offset, length) messages of a segment (compatibility with the round-1 interface)"""
        return [(m, o, n) for m, o, n, _w in self.plan(seg)]

    def reduce_segment(self, seg: int, flats, layer_events: bool = True):
        """layer_events=False: the caller modified the gradients on the current stream after the backward (accumulation steps
        add the running sums there): every message waits for the stream instead of the engine's per-layer events"""
        # KLAB_DDP_FORCE_COLLECTIVE=1: issue the collectives even in a one-rank group (test hook: exercises the comm-stream /
        # RCCL path on a single GPU; the mean over one rank is the identity)
        if not self.world_active():
            return
        joined = False  # comm stream already behind the whole segment
        for model, off, n, wait in self.plan(seg):
            flat = flats.get(model)
            if flat is None or n == 0:
                continue
            t = flat[off:off + n]
            self._calls += 1
            self._bytes += n * t.element_size()
            self.last_plan.append((seg, model, off, n, wait))
            if t.is_cuda:
                if self.comm_stream is None:
                    self.comm_stream = torch.cuda.Stream(device=t.device)
                if (wait is None or joined or not layer_events or self.engine is None
                        or not self.engine.bucket_wait(seg, wait, self.comm_stream)):
                    if not joined:
                        self.comm_stream.wait_stream(torch.cuda.current_stream(t.device))
                        joined = True
                with torch.cuda.stream(self.comm_stream):
                    if self.wire_dtype is not None:
                        self._bytes -= n * (t.element_size() - 2)
                        w16 = t.to(self.wire_dtype)  # (allocated and freed on the comm stream)
                        if self._avg_op is not None:
                            dist.all_reduce(w16, op=self._avg_op, group=self.pg)
                            t.copy_(w16)
                        else:
                            dist.all_reduce(w16, op=dist.ReduceOp.SUM, group=self.pg)
                            t.copy_(w16)
                            t.div_(self.world)
                    elif self._avg_op is not None:
                        dist.all_reduce(t, op=self._avg_op, group=self.pg)
                    else:
                        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
                        t.div_(self.world)
            elif self.wire_dtype is not None:
                self._bytes -= n * (t.element_size() - 2)
                w16 = t.to(self.wire_dtype)
                dist.all_reduce(w16, op=dist.ReduceOp.SUM, group=self.pg)
                t.copy_(w16)
                t.div_(self.world)
            else:
                w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                self._works.append((w, t))
        if self.comm_stream is not None:
            ev = self._seg_events.get(seg)
            if ev is None:
                ev = self._seg_events[seg] = torch.cuda.Event()
            ev.record(self.comm_stream)

    def finish_segment(self, seg: int, device=None):
        """partial join: the compute stream may consume segment `seg`'s averaged gradients (later segments may still be in
        flight).  CPU / gloo groups have no per-segment handle: they join everything."""
        if self._works:
            self.finish(device)
            return
        ev = self._seg_events.get(seg)
        if ev is not None and self.comm_stream is not None:
            torch.cuda.current_stream(device).wait_event(ev)

    def finish(self, device=None):
        """join: after this the compute stream may consume the averaged gradients."""
        for w, t in self._works:
            w.wait()
            t.div_(self.world)
        self._works = []
        if self.comm_stream is not None:
            torch.cuda.current_stream(device).wait_stream(self.comm_stream)


class DistributedDataParallel(nn.Module):
    """Drop-in for `torch.nn.parallel.DistributedDataParallel(model, device_ids=[...])` around the
    native MyModel (`.module`, `forward(*args)`), with segment-overlapped gradient reduction."""

    def __init__(self, module, device_ids=None, process_group=None, broadcast_parameters=True, max_bucket_elems=64 << 20,
                 overlap_optimizer=False, min_bucket_elems=6 << 20):
        """overlap_optimizer=True: backward returns without joining the last all-reduces; `optim.FusedAdam.step()` then
        updates segment 0 (decoder + embedding) while segment 1 (encoder) is still reducing.  Only for loops whose next
        consumer of the gradients is FusedAdam (the reference's loop, ref/train.py:62-69); anything else that reads
        `.grad` first (clipping, logging) must call `ddp.join()` before.  The next forward joins in any case."""
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.device_ids = device_ids
        eng = module._engine
        nseg = 3 if module.args.image_model_train else 2
        self.reducer = SegmentReducer(eng.segments[:nseg], process_group, max_bucket_elems, engine=eng, min_bucket_elems=min_bucket_elems,
                                      short_tail=True)
        self._nseg = nseg
        if hasattr(eng, "set_bucket_events"):  # per-layer ready events: only worth recording when collectives will be issued
            eng.set_bucket_events(dist.is_initialized())
        module._direct_grads = True
        module._segment_hook = self._on_segment
        module._reducer = self.reducer
        self.overlap_optimizer = bool(overlap_optimizer)
        module._pending_reduce = None
        if broadcast_parameters and dist.is_initialized() and dist.get_world_size(process_group) > 1:
            with torch.no_grad():  # torch DDP's _sync_module_states (TORCH/ddp:864-867)
                for p in module.parameters():
                    dist.broadcast(p.data, src=0, group=process_group)

    def _on_segment(self, seg, fresh=True):
        if seg == 0:
            self.reducer.last_plan = []
        flats = {"main": self.module._flat.get("main"), "swin": self.module._flat.get("swin")}
        self.reducer.reduce_segment(seg, flats, layer_events=fresh)
        if seg == self._nseg - 1:
            if self.overlap_optimizer and self.reducer.world_active():
                self.module._pending_reduce = self.reducer  # joined per segment by FusedAdam.step / fully by the next forward
            else:
                self.reducer.finish()

    def join(self):
        """wait (stream-wise) for every gradient all-reduce of the last backward"""
        if self.module._pending_reduce is not None:
            self.module._pending_reduce = None
            self.reducer.finish()

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
