#!/usr/bin/env python3
"""Caption-training throughput of the native Swin-V2 -> T5 path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload caption|cfg3|spanmask|cfg5]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...; without a launcher in the
     environment `python bench.py --gpus N` starts those N ranks itself as a child process, or fails -- it never measures one rank)

One step = what ref/train.py:58-67 does per iteration: loss = model(images, src, tgt); loss.backward();
optimizer.step(); optimizer.zero_grad()  -- forward + backward of MyModel (libklab_mm.so engine), the
gradient all-reduce over RCCL when N > 1 (klab DistributedDataParallel, overlapped per backward
segment) and the reference's Adam update over transformer.parameters().
Default workload = BASELINE.json configs[1], resolved per SURVEY §8d to a reference-runnable width-matched
pair: Swin-V2 C=64 (2,2,6,2)/(2,4,8,16) 224x224 w7 frozen + T5-small, bf16 operands with fp32
accumulation, batch 64 per GPU, Ls=9, Lt=64, T5 dropout 0.1 ON, synthetic inputs already resident
in HBM, random-init weights (no network).  The other workloads are the per-GPU slices of configs[2..4]
(parity-test cases and secondary data points, not the headline line).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

if "--graph" not in sys.argv:  # hipGraph replay is the exception: 16.6 ms/step with 8 queues against 6.9 with the default 4
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # see klab_multimodalmodel_amd/__init__.py: streams must not share HW queues
import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0           # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_FP8_TFLOPS = 5000.0            # dense fp8 MFMA peak, same table

# per-GPU workloads = BASELINE.json configs as SURVEY §8(d) resolves them (fwd+bwd algorithmic GFLOP/sample from that table)
WORKLOADS = {
    "caption": dict(name="BASELINE configs[1]: Swin-V2(C=64,(2,2,6,2),224,w7) frozen + T5-small", gflop=27.26, B=64, Ls=9, Lt=64,
                    swin=dict(image_size=224, embed_dim=64, depths=(2, 2, 6, 2), num_heads=(2, 4, 8, 16), window_size=7),
                    t5=dict(), train_swin=False, span=False),
    "cfg3": dict(name="BASELINE configs[2] per-GPU slice: Swin-V2(C=96,(2,2,18,2),224,w7) UNFROZEN + T5-base", gflop=137.27, B=32, Ls=9,
                 Lt=64, swin=dict(image_size=224, embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24), window_size=7),
                 t5=dict(d_model=768, d_ff=3072, num_heads=12, num_layers=12, num_decoder_layers=12), train_swin=True, span=False),
    "spanmask": dict(name="BASELINE configs[3] per-GPU slice: RedCaps span-mask pretraining (<extra_id_k> sentinels), "
                          "Swin-V2(C=96,(2,2,18,2),224,w7) UNFROZEN + T5-base", gflop=118.90, B=32, Ls=32, Lt=16,
                     swin=dict(image_size=224, embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24), window_size=7),
                     t5=dict(d_model=768, d_ff=3072, num_heads=12, num_layers=12, num_decoder_layers=12), train_swin=True, span=True),
    "cfg5": dict(name="BASELINE configs[4] per-GPU slice: Swin-V2(C=128,(2,2,18,2),384,w24,pretrained (12,12,12,6)) UNFROZEN + T5-large",
                 gflop=817.26, B=32, Ls=9, Lt=64,
                 swin=dict(image_size=384, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32), window_size=24,
                           pretrained_window_sizes=(12, 12, 12, 6)),
                 t5=dict(d_model=1024, d_ff=4096, num_heads=16, num_layers=24, num_decoder_layers=24), train_swin=True, span=False),
}
GFLOP_PER_SAMPLE = {"cfg2": WORKLOADS["caption"]["gflop"]}


def synth_batch(B, Ls, Lt, H, vocab, device, seed=1234):
    """SURVEY §8(d) synthetic inputs: N(0,1) pixels, uniform ids in [2, 32000), </s> last."""
    g = torch.Generator().manual_seed(seed)
    pix = torch.randn(B, 3, H, H, generator=g)
    src = torch.randint(2, 32000, (B, Ls), generator=g)
    tgt = torch.randint(2, 32000, (B, Lt), generator=g)
    src[:, -1] = 1
    tgt[:, -1] = 1
    return pix.to(device), src.to(device), tgt.to(device)


def synth_spanmask_batch(B, Ls, Lt, H, vocab, device, seed=1234, n_mask=4):
    """SURVEY §8(d), span-mask configuration: the token ids the reference's RedCaps path produces after tokenisation
    (ref/modules/loader.py:56-72, ref/train.py:56-57) -- `<extra_id_k>` = id 32099 - k (HF/t5tok:99-110).
    src: random words with sentinels k = 0..n_mask-1 planted in increasing order, </s>, pad tail;
    tgt: <extra_id_0> w <extra_id_1> w ... <extra_id_n_mask> </s>, padded with 0 to Lt (pads ARE scored: SURVEY §0.4);
    25 % of the rows are shorter (more padding on both sides)."""
    g = torch.Generator().manual_seed(seed)
    pix = torch.randn(B, 3, H, H, generator=g)
    src = torch.zeros(B, Ls, dtype=torch.int64)
    tgt = torch.zeros(B, Lt, dtype=torch.int64)
    for b in range(B):
        short = (b % 4) == 3
        ls = Ls - (Ls // 4 if short else 0)
        words = torch.randint(2, 32000, (ls - 1,), generator=g)
        pos = torch.sort(torch.randperm(ls - 1, generator=g)[:n_mask]).values
        for k, p in enumerate(pos.tolist()):
            words[p] = 32099 - k
        src[b, :ls - 1] = words
        src[b, ls - 1] = 1
        row = []
        for k in range(n_mask):
            row.append(32099 - k)
            nw = 1 if short else int(torch.randint(1, 3, (1,), generator=g))  # a masked word is 1-2 sentencepiece tokens
            row.extend(torch.randint(2, 32000, (nw,), generator=g).tolist())
        row.append(32099 - n_mask)
        row.append(1)
        row = row[:Lt]
        tgt[b, :len(row)] = torch.tensor(row)
    return pix.to(device), src.to(device), tgt.to(device)


def workload_configs(name):
    from klab_multimodalmodel_amd.engine import SwinConfig, T5Config
    w = WORKLOADS[name]
    return SwinConfig(**w["swin"]), T5Config(**w["t5"])


def cfg2_configs():
    return workload_configs("caption")


def usable_cores():
    """host cores this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU box
    exposes far more logical CPUs than its share; oversubscribing OpenMP threads would only time spinning)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("KLAB_CPU_BASELINE_THREADS", "16"))))


def cpu_baseline(budget_s=20.0):
    """oracle (CPU restatement of the reference path, oracle/swin_t5_oracle.py) timed on the host cores at
    BASELINE.json configs[0]: B=2, fp32, T5 dropout on, Adam step included."""
    from oracle import swin_t5_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    sc = O.SwinCfg(image_size=224, embed_dim=64, depths=(2, 2, 6, 2), num_heads=(2, 4, 8, 16), window_size=7)
    tc = O.T5Cfg()
    swin_sd = O.hf_like_init_swin(sc, g)
    lang_sd = O.hf_like_init_t5(tc, g, encoder_only=True)
    main_sd = {k: v.requires_grad_(True) for k, v in O.hf_like_init_t5(tc, g).items()}
    opt = torch.optim.Adam(list(main_sd.values()), lr=1e-3)
    B = 2
    pix, src, tgt = synth_batch(B, 9, 64, 224, 32128, "cpu")
    times = []
    t_all = time.perf_counter()
    it = 0
    while True:
        t0 = time.perf_counter()
        loss = O.mymodel_forward(swin_sd, lang_sd, main_sd, sc, tc, tc, pix, src, tgt, training=True)
        loss.backward()
        opt.step()
        opt.zero_grad()
        dt = time.perf_counter() - t0
        it += 1
        if it > 2:
            times.append(dt)
        if (time.perf_counter() - t_all > budget_s and len(times) >= 3) or len(times) >= 60:  # ~15 s of CPU work on 16 cores (bounded by budget_s)
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(B / med, 3), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle fwd+bwd+Adam, configs[0] (B=2, fp32, Ls=9, Lt=64, 224px), median of {len(times)} steps after 2 warm-up"}


def _traffic(key):
    """HBM bytes per launch from the PMC passes kept under profiles/ (collected per MI355X_MICROARCH.md's rocprofv3 recipe)"""
    tp = os.path.join(ROOT, "profiles", "kernel_traffic.json")
    try:
        return json.load(open(tp)).get(key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """`python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE in the environment): start the N ranks ourselves, the way
    ref/run_scripts/caption/train_with_swin.sh:1 starts the reference (torchrun, one process per GPU), as a CHILD process --
    this process has not touched the GPU and never will -- relay rank 0's single JSON line and exit with the child's code.
    A run that cannot start its ranks fails; it never falls back to one rank."""
    import subprocess
    avail = torch.cuda.device_count()  # (does not initialise the GPU on this image)
    if "KLAB_BENCH_DEVICE" not in os.environ and avail < n:
        print(f"[bench] --gpus {n} but only {avail} GPU(s) visible: refusing to measure fewer ranks than asked for", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes on this driver
    print("[bench] no launcher in the environment: starting", n, "ranks:", " ".join(cmd), file=sys.stderr, flush=True)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        print(f"[bench] ranks exited with {r.returncode}, {len(lines)} JSON line(s) on stdout", file=sys.stderr)
        sys.stderr.write(r.stdout[-2000:])
        return r.returncode or 3
    d = json.loads(lines[0])
    if d.get("n_gpus") != n:
        print(f"[bench] asked for {n} ranks, the line reports n_gpus={d.get('n_gpus')}", file=sys.stderr)
        return 4
    print(lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="caption",
                    help="caption = BASELINE configs[1] (the headline metric); cfg3 / spanmask / cfg5 = per-GPU slices of configs[2..4]")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's, configs[1]: 64)")
    ap.add_argument("--dtype", choices=["bf16", "fp8"], default="bf16", help="fp8: per-tensor-scaled fp8 MFMA GEMMs (configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--optimizer", choices=["klab", "torch"], default="klab",
                    help="klab: klab_multimodalmodel_amd.optim.FusedAdam (same update rule as torch.optim.Adam, one kernel); torch: torch.optim.Adam(fused=True)")
    ap.add_argument("--step-in-backward", action="store_true",
                    help="klab optimizer: start the Adam update of backward segment 0 (LM head / embedding / decoder) while the encoder's "
                         "backward is still running (measured: no gain, 6.67 vs 6.63 ms/step -- the co-running kernel slows the chain by what it hides)")
    ap.add_argument("--graph", action="store_true", help="replay the engine's launch sequences as hipGraphs (measured: no gain "
                    "while the step is GPU-bound; kept for when it becomes launch-bound)")
    a = ap.parse_args()
    wl = WORKLOADS[a.workload]
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:  # before ANY GPU call of this process
        sys.exit(self_launch(a.gpus, sys.argv[1:]))
    # stdout carries exactly ONE JSON line: everything else that writes to file descriptor 1 while the bench runs (RCCL's
    # version banner, library notices) is sent to stderr; the descriptor is restored for the final line
    sys.stdout.flush()
    _stdout_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or os.environ.get("KLAB_BENCH_FORCE_DIST") in ("1", "2")  # the latter: rehearse the N>1 code path with one rank
    if a.gpus != world and (dist_on or a.gpus > 1):  # never print n_gpus != --gpus
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # rehearsal knobs (never set by the driver): KLAB_BENCH_DEVICE pins every rank to one card and KLAB_BENCH_BACKEND=gloo moves the
    # collectives through the host, so that the N > 1 code path (ranks, barrier, reducer, rank-0 JSON) can run on a one-GPU box
    dev_index = int(os.environ.get("KLAB_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("KLAB_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from klab_multimodalmodel_amd.models.model import MyModel
    sw, t5 = workload_configs(a.workload)
    args = types.SimpleNamespace(result_dir="/tmp", language_model_name="-", image_model_name="-",
                                 image_model_train=wl["train_swin"], transformer_model_name="-")
    torch.manual_seed(0)
    model = MyModel(args, _configs=(sw, t5, t5), _seed=0, dtype=a.dtype).to(dev)
    ddp = None
    if dist_on and os.environ.get("KLAB_BENCH_FORCE_DIST") != "2":  # "2": process group without the wrapper (diagnostic)
        from klab_multimodalmodel_amd.ddp import DistributedDataParallel as DDP
        model = ddp = DDP(model, device_ids=[dev_index],
                          overlap_optimizer=(a.optimizer == "klab" and os.environ.get("KLAB_BENCH_OVERLAP_OPT", "1") == "1"))
        core = model.module
    else:
        core = model
        core._direct_grads = True  # grads land in the flat buffer (no autograd copies); same math
    core.use_graph = bool(a.graph)  # forward / backward launch sequences replayed as hipGraphs (same kernels, same math)
    lr = 1e-3 if a.workload == "caption" else 1e-4
    if a.optimizer == "klab":  # torch.optim.Adam's update rule in one kernel over the flat buffers (+ bf16 weight copies)
        from klab_multimodalmodel_amd.optim import FusedAdam
        optimizer = FusedAdam(core.transformer.parameters(), lr=lr, step_in_backward=a.step_in_backward)
    else:                      # ref/train.py:28 verbatim (torch's own fused multi-tensor kernel)
        optimizer = torch.optim.Adam(core.transformer.parameters(), lr=lr, fused=True)
    core.transformer.train()                                               # ref/train.py:52

    B, Ls, Lt = (a.batch or wl["B"]), wl["Ls"], wl["Lt"]
    synth = synth_spanmask_batch if wl["span"] else synth_batch
    pix, src, tgt = synth(B, Ls, Lt, sw.image_size, 32128, dev, seed=1234 + rank)
    images, se, te = {"pixel_values": pix}, {"input_ids": src}, {"input_ids": tgt}

    def step():
        loss = model(images, se, te)
        loss.backward()
        optimizer.step()
        optimizer.zero_grad()
        return loss

    for _ in range(a.warmup):
        step()
    eng = core._engine
    eng.probe_enable(True)
    if ddp is not None:
        ddp.reducer.reset_stats()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    host_ms = (time.perf_counter() - t0) / a.steps * 1e3  # host time to ENQUEUE a step (no device wait): < ms_per_step => GPU-bound
    torch.cuda.synchronize()
    t_sync = time.perf_counter()
    if dist_on:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on and rank == 0:
        print(f"[bench] closing barrier took {(time.perf_counter() - t_sync) * 1e3:.2f} ms", file=sys.stderr, flush=True)
    rank_ms = None
    if dist_on:
        # per-rank wall time of the K steps up to the rank's own synchronize (before the closing barrier): the spread shows stragglers
        own = torch.tensor([(t_sync - t0) / a.steps * 1e3], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(own) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, own)
        rank_ms = [round(float(x.item()), 3) for x in allr]
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    probes = [eng.probe_read(ch) for ch in (0, 1)]
    eng.probe_enable(False)
    lossv = float(loss.item())
    reducer_stats = ddp.reducer.stats() if ddp is not None else None  # (of the timed steps only: the probe steps below reduce too)
    # OUTSIDE the timed region: three more steps with HIP events around every klab_gemm launch -> the whole GEMM family's
    # sum(2 M N K) / sum(duration) (every Linear / dgrad / LM-head product of the step; the grouped weight gradients are channel 1)
    family = None
    if rank == 0:
        import ctypes as C
        from klab_multimodalmodel_amd import _lib as L
        lib = L.load()
        fam_steps = 3
        if lib.klab_gemm_probe_enable(1) == 0:
            for _ in range(fam_steps):
                step()
            torch.cuda.synchronize()
            n, tms, fl = C.c_int(), C.c_float(), C.c_double()
            if lib.klab_gemm_probe_read(C.byref(n), C.byref(tms), C.byref(fl)) == 0 and n.value > 0 and tms.value > 0:
                family = (n.value / fam_steps, tms.value / fam_steps, fl.value / fam_steps)
            lib.klab_gemm_probe_enable(0)
    elif dist_on:
        for _ in range(3):  # keep the ranks' collectives paired with rank 0's probe steps
            step()
        torch.cuda.synchronize()
    in_sync = None
    if dist_on:  # (outside the timed region) after K optimizer steps every replica must hold the same weights
        chk = torch.stack([p.detach().double().sum() for p in core.transformer.parameters()]).sum().view(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool(lo.item() == hi.item())

    if rank == 0:
        # fp8 mode runs its products on the NON-scaled fp8 MFMA (the block-scaled kernel, csrc/mmf8.hip, measured slower on these
        # shapes): that instruction issues at the bf16 rate, so fp8 lines are priced against the bf16 peak, not the 5 PF one
        peak = PEAK_BF16_TFLOPS
        ms = dt / a.steps * 1e3
        value = world * B * a.steps / dt
        out = {
            "metric": "caption-train samples/sec (224px img, 64-tok tgt)", "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{wl['name']}, fwd+bwd+Adam, T5 dropout 0.1 on, random-init weights",
                       "global_batch": world * B, "per_gpu_batch": B, "src_len": Ls, "tgt_len": Lt,
                       "parallelism": f"dp{world}", "hipgraph": bool(core.use_graph),
                       "optimizer": (("klab.optim.FusedAdam (Adam update of ref/train.py:28; segment 0 updated underneath the encoder backward, segment 1 at step())"
                                      if a.step_in_backward else "klab.optim.FusedAdam (Adam update of ref/train.py:28, one kernel)") if a.optimizer == "klab"
                                     else "torch.optim.Adam(fused=True)"), "fwd_bwd_gflop_per_sample": wl["gflop"],
                       "step_mfma_frac": round(B * wl["gflop"] / (ms * 1e-3) / 1e3 / peak, 4),
                       "host_enqueue_ms_per_step": round(host_ms, 3), "final_loss": round(lossv, 4)},
        }
        if ddp is not None:  # evidence that the N > 1 path really reduced over RCCL: ranks and bytes all-reduced per step
            st = reducer_stats
            out["config"]["rccl"] = {"ranks": dist.get_world_size(), "backend": dist.get_backend(),
                                     "allreduce_calls_per_step": round(st["calls"] / max(a.steps, 1), 2),
                                     "allreduce_bytes_per_step": int(st["bytes"] / max(a.steps, 1)),
                                     "replicas_in_sync_after_run": in_sync}
        if rank_ms:
            out["config"]["rank_ms_per_step"] = {"min": min(rank_ms), "max": max(rank_ms), "per_rank": rank_ms}
        # roofline of the kernels that hold the largest shares of GPU time (profiles/*_kernel_stats.csv): live HIP-event
        # durations around each launch on the stream it runs on; algorithmic FLOPs = 2*M*N*K per product
        rl = []
        names = ["klab_lmhead_areg_gemm / klab_lmhead_gemm<bf16> (LM-head logits GEMM [B*Lt, 32128] x d_model; A-stationary form at d_model 512)",
                 "mm8p_grouped_tn_kernel / gemm_glds_grouped_tn_kernel (one grouped launch = the weight-gradient GEMMs of 2-3 T5 layers on 256 x 256 tiles, or of one layer on 128 x 128; side stream)"]
        for ch, (n, tot_ms, fl) in enumerate(probes):
            if n <= 0 or tot_ms <= 0:
                continue
            ach = fl / (tot_ms * 1e-3) / 1e12
            rl.append({"kernel": names[ch], "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                       "frac": round(ach / peak, 4), "traffic": _traffic("lmhead" if ch == 0 else "grouped_wgrad"),
                       "avg_launch_ms": round(tot_ms / n, 4), "launches": n, "flops_per_launch": fl / n,
                       "gpu_time_share_per_step": round(tot_ms / a.steps / ms, 4)})
        if rl:
            rl.sort(key=lambda r: -r["gpu_time_share_per_step"])  # dominant (largest share of the step) first
            if family is not None:
                n_f, ms_f, fl_f = family
                ach = fl_f / (ms_f * 1e-3) / 1e12
                rl.append({"kernel": "klab_gemm family: every Linear / dgrad / LM-head launch of a step (all tile kernels; grouped weight gradients excluded), "
                                     "sum(2MNK) / sum(event duration), measured in 3 untimed steps after the timed region",
                           "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                           "launches_per_step": round(n_f, 1), "gpu_ms_per_step": round(ms_f, 3), "flops_per_step": fl_f,
                           "gpu_time_share_per_step": round(ms_f / ms, 4)})
            out["roofline"] = rl[0]
            out["roofline"]["whole_step_frac"] = out["config"]["step_mfma_frac"]
            if len(rl) > 1:
                out["roofline_secondary"] = rl[1:]
        print(f"[bench] gpu leg done: {value:.1f} samples/s, {ms:.2f} ms/step", file=sys.stderr, flush=True)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.dup2(_stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist_on:
        del model, core, optimizer, ddp, eng
        import gc
        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
