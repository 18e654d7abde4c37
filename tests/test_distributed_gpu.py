"""Multi-rank LM calibration with the GPU solver: two ranks, instances sharded by cost, one all-reduce of
31 doubles per iteration -- must take the same path as a single rank holding the whole surface."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(nproc, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "dist_gpu_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-3000:]
    return json.loads(lines[-1][7:])


def test_two_rank_calibration_equals_single_rank():
    one, two = _run(1, 29631), _run(2, 29632)
    assert two["world"] == 2 and two["shard"][0] == 0 and 0 < two["shard"][1] < 60
    assert one["iterations"] == two["iterations"] and one["pde_solves"] == two["pde_solves"] == 60 * 7 * one["iterations"] - 60
    for a, b in zip(one["errors"], two["errors"]):
        assert abs(a - b) <= 1e-6 * max(1.0, a)          # J^T J is summed in a different order across ranks
    for k in ("eta", "sigma", "rho", "v0"):
        assert abs(one[k] - two[k]) <= 1e-3 * max(1e-2, abs(one[k]))
    assert abs(one["kappa"] - two["kappa"]) <= 1e-2 * abs(one["kappa"])
    assert max(abs(x - y) for x, y in zip(one["prices"], two["prices"])) <= 1e-4
