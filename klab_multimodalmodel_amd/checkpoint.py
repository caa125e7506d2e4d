"""Checkpoint I/O with true resume.

The reference saves weights only -- `MyModel.save` writes {'transformer': sd[, 'image_model': sd]} (ref/models/model.py:30-42),
called from rank 0 at the end of an epoch (ref/train.py:88-104) -- and cannot resume: optimizer moments, scheduler and step
counters are lost.  This module keeps that file schema (a file written here loads with the reference's `MyModel.load`) and
adds what a resume needs, without stalling the training loop:

  * snapshot: every tensor is copied device -> pinned host memory on a side stream (non-blocking); the training stream only
    waits for that copy to be ENQUEUED, never for the disk;
  * write: a background thread waits for the copy event, `torch.save`s to `<name>.tmp` and renames (a crash never leaves a
    half-written checkpoint under the final name);
  * sharded optimizer state: under data parallelism every rank holds identical Adam moments, so rank r writes only slice r of
    the flat exp_avg / exp_avg_sq buffers (`<name>.opt<r>of<w>`), 1/world of the bytes per rank; rank 0 writes the model.
    With an optimizer other than `optim.FusedAdam` on its flat buffers the whole `state_dict()` goes into rank 0's file.

    ck = AsyncCheckpointer(args.result_dir)
    ck.save(model, optimizer, scheduler, step=global_step, name="epoch_3.pth", rank=rank, world=world)   # returns at once
    ...
    step = load_checkpoint(os.path.join(args.result_dir, "epoch_3.pth"), model, optimizer, scheduler)     # after one forward
"""
import os
import threading

import torch


def _to_host(obj, stream):
    """deep copy of a (nested) state dict with tensors in pinned host memory, copies enqueued on `stream`"""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            host = torch.empty(obj.shape, dtype=obj.dtype, device="cpu").pin_memory()
            with torch.cuda.stream(stream):
                host.copy_(obj, non_blocking=True)
            return host
        return obj.detach().clone()
    if isinstance(obj, dict):
        return {k: _to_host(v, stream) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_host(v, stream) for v in obj)
    return obj


def _core(model):
    return model.module if hasattr(model, "module") else model


def _flat_state(optimizer):
    """(m, v, steps) when `optimizer` is a FusedAdam running on its flat buffers, else None"""
    if getattr(optimizer, "_fallback", True) is None and getattr(optimizer, "_m", None) is not None:
        return optimizer._m, optimizer._v, optimizer._steps
    return None


class AsyncCheckpointer:
    def __init__(self, result_dir):
        self.dir = result_dir
        os.makedirs(result_dir, exist_ok=True)
        self._thread = None
        self._error = None
        self._stream = None

    def wait(self):
        """block until the last save is on disk; re-raises its error, if any"""
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self._error is not None:
            e, self._error = self._error, None
            raise e

    def save(self, model, optimizer=None, scheduler=None, step=0, name="checkpoint.pth", rank=0, world=1, extra=None):
        self.wait()  # one save in flight: the pinned snapshot of the previous one is released first
        core = _core(model)
        dev = next(core.transformer.parameters()).device
        if hasattr(core, "_join_weights"):
            core._join_weights()  # an update running on the optimizer's own stream (FusedAdam overlap_next_forward)
        if optimizer is not None and hasattr(optimizer, "join"):
            optimizer.join()
        stream = None
        if dev.type == "cuda":
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=dev)
            stream = self._stream
            stream.wait_stream(torch.cuda.current_stream(dev))  # the snapshot sees every update enqueued so far
        payload, shard = None, None
        flat = _flat_state(optimizer) if optimizer is not None else None
        if rank == 0:
            payload = {"transformer": _to_host(core.transformer.state_dict(), stream)}  # ref/models/model.py:32
            if getattr(core.args, "image_model_train", False):
                payload["image_model"] = _to_host(core.image_model.state_dict(), stream)  # :33-34
            payload["step"] = int(step)
            payload["world"] = int(world)
            payload["rng"] = {"cpu": torch.get_rng_state(), "cuda": torch.cuda.get_rng_state(dev) if dev.type == "cuda" else None}
            if scheduler is not None:
                payload["scheduler"] = scheduler.state_dict()
            if extra is not None:
                payload["extra"] = extra
            if optimizer is not None:
                if flat is None:
                    payload["optimizer"] = _to_host(optimizer.state_dict(), stream)
                else:
                    payload["optimizer_flat"] = {"numel": flat[0].numel(), "steps": int(flat[2]), "shards": int(world),
                                                 "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in optimizer.param_groups]}
        if flat is not None:
            n = flat[0].numel()
            lo, hi = n * rank // world, n * (rank + 1) // world
            shard = {"lo": lo, "hi": hi, "exp_avg": _to_host(flat[0][lo:hi], stream), "exp_avg_sq": _to_host(flat[1][lo:hi], stream)}
        ev = None
        if stream is not None:
            ev = torch.cuda.Event()
            ev.record(stream)
            torch.cuda.current_stream(dev).wait_stream(stream)  # later updates must not overtake the snapshot's reads
        path = os.path.join(self.dir, name)

        def write():
            try:
                if ev is not None:
                    ev.synchronize()
                for obj, p in ((payload, path), (shard, f"{path}.opt{rank}of{world}")):
                    if obj is not None:
                        torch.save(obj, p + ".tmp")
                        os.replace(p + ".tmp", p)
            except BaseException as e:  # surfaced by wait()
                self._error = e

        self._thread = threading.Thread(target=write, name="klab-checkpoint", daemon=False)
        self._thread.start()
        return path


def load_checkpoint(path, model, optimizer=None, scheduler=None, restore_rng=False, map_location="cpu"):
    """inverse of AsyncCheckpointer.save; returns the saved step.  For the FusedAdam flat state the model must have run one
    forward (its flat buffers exist then); every rank reads all shards (the moments are replicated under data parallelism)."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    core = _core(model)
    core.transformer.load_state_dict(ck["transformer"])          # ref/models/model.py:38
    if "image_model" in ck and getattr(core.args, "image_model_train", False):
        core.image_model.load_state_dict(ck["image_model"])      # :39-40
    if optimizer is not None:
        if "optimizer" in ck:
            optimizer.load_state_dict(ck["optimizer"])
        elif "optimizer_flat" in ck:
            meta = ck["optimizer_flat"]
            m = torch.empty(meta["numel"], dtype=torch.float32)
            v = torch.empty(meta["numel"], dtype=torch.float32)
            for r in range(meta["shards"]):
                sh = torch.load(f"{path}.opt{r}of{meta['shards']}", map_location="cpu", weights_only=False)
                m[sh["lo"]:sh["hi"]] = sh["exp_avg"]
                v[sh["lo"]:sh["hi"]] = sh["exp_avg_sq"]
            flat = core._flat.get("main") if hasattr(core, "_flat") else None
            if flat is None or flat.numel() != meta["numel"]:
                raise RuntimeError("flat optimizer state: run one forward before load_checkpoint (the engine lays the buffers out then)")
            core._grad_targets()
            optimizer._m, optimizer._v = m.to(flat.device), v.to(flat.device)
            optimizer._steps = meta["steps"]
            optimizer._fallback = None
            import weakref
            optimizer._owner = weakref.ref(core)
            for g, saved in zip(optimizer.param_groups, meta["param_groups"]):
                g.update(saved)
    if scheduler is not None and "scheduler" in ck:
        scheduler.load_state_dict(ck["scheduler"])
    if restore_rng and "rng" in ck:
        torch.set_rng_state(ck["rng"]["cpu"])
        if ck["rng"]["cuda"] is not None and torch.cuda.is_available():
            torch.cuda.set_rng_state(ck["rng"]["cuda"])
    return ck.get("step", 0)
