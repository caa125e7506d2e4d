"""Every A/B switch of round 3 selects another FORM of the same arithmetic: the loss after a few optimizer steps of the bench workload
(BASELINE configs[1], fixed seeds, dropout on) must not depend on it beyond bf16 rounding.  A switch that dropped a gradient, skipped a
clear or mis-ordered two streams moves the trajectory by far more than that.  One `bench.py` child process per setting (the switches
are read once per process)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = [
    {"KLAB_ZERO_ON_SIDE": "0"},       # gradient slices cleared on the main stream at the head of each segment
    {"KLAB_EARLY_SMALL": "0"},        # position biases / decoder embedding / label count where they are used
    {"KLAB_LMHEAD_AREG": "0"},        # LM-head logits on the 128 x 128 tiled kernel
    {"KLAB_T5_ATTN_FUSED": "0"},      # rms-norm + projection + attention as three launches
    {"KLAB_SWIN_FUSED_LIN_LN": "0", "KLAB_SWIN_FUSED_EMBED": "0"},  # frozen tower: GEMM + LayerNorm, im2col + GEMM + LayerNorm
    {"KLAB_WGRAD_GROUP_TILES": "0"},  # weight gradients: one 128-wide grouped launch per layer instead of 256-wide tiles over 2-3 layers
    {"KLAB_GEMM_P8": "0"},            # no 256 x 256 tiles (LM-head input gradient on 128 x 128 tiles, split K)
]


def _final_loss(extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])["config"]["final_loss"]


@pytest.fixture(scope="module")
def default_loss():
    return _final_loss({})


@pytest.mark.parametrize("sw", SWITCHES, ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_switch_selects_another_form_of_the_same_arithmetic(default_loss, sw):
    loss = _final_loss(sw)
    # four Adam steps at lr 1e-3 from the same initial weights: the loss has moved by ~1 from its initial value, bf16 forms agree to ~1e-3
    assert abs(loss - default_loss) <= 5e-3 * abs(default_loss), (sw, loss, default_loss)
