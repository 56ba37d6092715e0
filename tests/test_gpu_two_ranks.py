"""GPU: BASELINE config 5's code path on ONE MI355X -- `bench.py --gpus 2 --share-gpu`: the launcher starts two fresh rank processes (before
anything touches the GPU: never re-exec a process that has), each solves its own shard of quadrotor N=50 on cuda:0 through the C ABI, timing is
reduced with max-over-ranks, the solutions are gathered once (gloo here, RCCL on a node with one GPU per rank: same helpers, sharding.py).
What is sharded: the loop of the reference's SQPOptimizationSolver.cpp:137-198 over independent instances.  No scaling number is claimed from this."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_share_one_gpu(built):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--no-extras", "--no-cpu-baseline", "--horizon", "50", "--batch", "256",
           "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]      # rank 0 prints the one line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    col = out["collective"]
    assert col["world"] == 2 and col["backend"] == "gloo" and col["share_gpu"] is True
    assert len(col["kernel_ms_per_rank"]) == 2 and all(k > 0 for k in col["kernel_ms_per_rank"])
    assert col["gather_bytes"] == 2 * 256 * 812 * 8        # world x batch x n doubles: the gathered solution block (its shape is asserted inside bench.py)
    assert out["solve_stats"]["solved_frac"] == 1.0
    assert out["config"]["batch_per_gpu"] == 256 and "horizon=50" in out["config"]["workload"]
    assert out["value"] > 0 and abs(out["value"] - 2 * 256 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]      # whole-job rate: units of all ranks over the max-over-ranks time
