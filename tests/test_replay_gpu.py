"""Graph replay == eager launches, bit for bit, on the FULL SD2.1-base handle (tools/replay_guard.py): 200 replays per
attention mode with the compiled-in plans, every ring depth / split-K of the LDS-DMA ring kernels forced onto every layer
shape that admits it, and the 20-step device loop.  Runs in a child process: the forced plans are process-global debug state
(sd_tune_*, enabled by SD_TUNE=1) that must not leak into the other tests, and a hang stays bounded by the timeout."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_full_size_graph_replay_is_bit_identical_to_eager_for_every_ring_and_split():
    env = dict(os.environ, SD_TUNE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "replay_guard.py"), "200", "25"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=1500)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("REPLAY_GUARD ")]
    assert lines, f"replay_guard produced no report (rc {r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}"
    rep = json.loads(lines[-1][len("REPLAY_GUARD "):])
    assert rep["failures"] == [] and r.returncode == 0, rep["failures"]
    assert set(rep["default"]) == {"ORIGINAL", "SPLIT_EINSUM", "SPLIT_EINSUM_V2"} and not any(rep["default"].values())
    assert len(rep["forced"]) == 105 and rep["loop20_equal"]
