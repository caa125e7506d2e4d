"""bench.py's N > 1 code path on real hardware with what a one-GPU box allows: two ranks pinned to card 0, the collectives through
gloo (RCCL refuses two ranks on one device).  Everything but the transport is the production path: torch.distributed.run launch,
RANK / WORLD_SIZE handling, klab DistributedDataParallel with per-layer buckets and overlap_optimizer, FusedAdam, the barrier /
max-over-ranks timing and the single JSON line on rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_on_one_card_reduce_and_stay_in_sync():
    env = dict(os.environ, KLAB_BENCH_DEVICE="0", KLAB_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["config"]["global_batch"] == 128
    rc = d["config"]["rccl"]
    assert rc["ranks"] == 2 and rc["replicas_in_sync_after_run"] is True
    # every trainable T5 gradient crosses the wire once per step: 60.5 M fp32 values
    assert rc["allreduce_bytes_per_step"] == 242026496 and rc["allreduce_calls_per_step"] >= 4
    assert d["value"] > 0 and abs(d["value"] - 2 * 64 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-2 * d["value"]


def test_bench_gpus2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the shape of the driver's one-GPU command): bench.py starts
    the two ranks itself as a child torchrun (ref/run_scripts/caption/train_with_swin.sh:1) and relays ONE line with n_gpus == 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(KLAB_BENCH_DEVICE="0", KLAB_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl"]["ranks"] == 2 and d["config"]["rccl"]["replicas_in_sync_after_run"] is True
    assert len(d["config"]["rank_ms_per_step"]["per_rank"]) == 2 and "roofline" in d
