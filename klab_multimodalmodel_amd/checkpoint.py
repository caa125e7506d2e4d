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
        self._hold = None

    def wait(self):
        """block until the last save is on disk; re-raises its error, if any"""
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        self._hold = None  # pinned snapshot buffers are released here, on the caller's thread (not by the writer thread)
        if self._error is not None:
            e, self._error = self._error, None
            raise e

    def save(self, model, optimizer=None, scheduler=None, step=0, name="checkpoint.pth", rank=0, world=1, extra=None):
        self.wait()  # one save in flight: the pinned snapshot of the previous one is released first
        core = _core(model)
        dev = next(core.transformer.parameters()).device
        # 1. everything that creates tensors on the CURRENT stream first (FusedAdam.state_dict() clones its moments, other
        #    optimizers may build theirs lazily) ...
        flat = _flat_state(optimizer) if optimizer is not None else None
        if hasattr(core, "_join_pending_update"):
            core._join_pending_update()  # an in-backward optimizer update on its own stream still writes the weights
        t_sd = core.transformer.state_dict() if rank == 0 else None
        i_sd = core.image_model.state_dict() if rank == 0 and getattr(core.args, "image_model_train", False) else None
        o_sd = optimizer.state_dict() if rank == 0 and optimizer is not None and flat is None else None
        eng = getattr(core, "_engine", None)
        rng_dev = eng.rng_view if rank == 0 and eng is not None and getattr(eng, "shape", None) is not None else None
        # 2. ... then the side stream waits for the current one: the snapshot sees every update and every temporary above
        stream = None
        if dev.type == "cuda":
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=dev)
            stream = self._stream
            stream.wait_stream(torch.cuda.current_stream(dev))
        payload, shard = None, None
        if rank == 0:
            payload = {"transformer": _to_host(t_sd, stream)}  # ref/models/model.py:32
            if i_sd is not None:
                payload["image_model"] = _to_host(i_sd, stream)  # :33-34
            payload["step"] = int(step)
            payload["world"] = int(world)
            payload["rng"] = {"cpu": torch.get_rng_state(), "cuda": torch.cuda.get_rng_state(dev) if dev.type == "cuda" else None}
            # the engine's device-side dropout RNG {step seed, base, forwards since seeding} + the model's base seed: a resumed
            # run continues the mask stream instead of replaying the masks of steps 1..k
            payload["engine_rng"] = {"seed_base": int(getattr(core, "_seed_base", 0)),
                                     "state": _to_host(rng_dev, stream) if rng_dev is not None else None}
            if scheduler is not None:
                payload["scheduler"] = scheduler.state_dict()
            if extra is not None:
                payload["extra"] = extra
            if optimizer is not None:
                if flat is None:
                    payload["optimizer"] = _to_host(o_sd, stream)
                else:
                    payload["optimizer_flat"] = {"numel": flat[0].numel(), "steps": int(flat[2]), "shards": int(world),
                                                 "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in optimizer.param_groups]}
        if flat is not None:
            n = flat[0].numel()
            lo, hi = n * rank // world, n * (rank + 1) // world
            shard = {"lo": lo, "hi": hi, "exp_avg": _to_host(flat[0][lo:hi], stream), "exp_avg_sq": _to_host(flat[1][lo:hi], stream)}
        keep = (t_sd, i_sd, o_sd, rng_dev)  # device temporaries stay referenced until the copy event has completed (write())
        ev = None
        if stream is not None:
            ev = torch.cuda.Event()
            ev.record(stream)
            torch.cuda.current_stream(dev).wait_stream(stream)  # later updates must not overtake the snapshot's reads
        path = os.path.join(self.dir, name)

        def write():
            nonlocal keep
            try:
                if ev is not None:
                    ev.synchronize()
                keep = None  # the device-side sources may go now
                for obj, p in ((payload, path), (shard, f"{path}.opt{rank}of{world}")):
                    if obj is not None:
                        torch.save(obj, p + ".tmp")
                        os.replace(p + ".tmp", p)
            except BaseException as e:  # surfaced by wait()
                self._error = e

        self._hold = (payload, shard)
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
    er = ck.get("engine_rng")
    if er is not None and hasattr(core, "_seed_base"):  # continue the dropout mask stream (always: it is part of the training state)
        core._seed_base = int(er["seed_base"])
        if er.get("state") is not None:
            st = [int(x) & 0xFFFFFFFF for x in er["state"].tolist()]
            eng = core._engine
            if getattr(eng, "shape", None) is not None and core._bound_key is not None:
                eng.set_rng(st[1], st[2])
            else:
                core._pending_rng = (st[1], st[2])
    if restore_rng and "rng" in ck:
        torch.set_rng_state(ck["rng"]["cpu"])
        if ck["rng"]["cuda"] is not None and torch.cuda.is_available():
            torch.cuda.set_rng_state(ck["rng"]["cuda"])
    return ck.get("step", 0)
