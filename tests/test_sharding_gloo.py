"""CPU: the N > 1 path (static batch sharding, timing reduction, final gather) with world_size 2 over gloo.
The CPU oracle stands in for the per-rank GPU solve; on the GPU box bench.py runs the same helpers over RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from optimal_control_problem_amd import models, sharding
from tests.support import problems
rank, world, local, dist = sharding.init_distributed(2, backend="gloo")
assert world == 2
mdl, ls, _ = models.make_workload("double_integrator", 11, seed=99)      # 6 + 5 rows: uneven shards
a, b = sharding.shard_range(ls.batch, rank, world)
mine = models.LocalSystem(ls.n, ls.m, ls.Pp, ls.Pi, ls.Ap, ls.Ai, ls.P[a:b], ls.q[a:b], ls.A[a:b], ls.l[a:b], ls.u[a:b])
res = problems.oracle_solve(mine)
sharding.barrier(dist)
t = sharding.max_over_ranks(1.0 + rank, dist)
s = sharding.sum_over_ranks(float(b - a), dist)
x = sharding.gather_rows(res["x"], dist, dst=0)
rec = sharding.collective_record(dist, 10.0 + rank, ls.batch * ls.n * 8, 1.5, rank)
assert sharding.all_values(3.0 * rank, dist) == [0.0, 3.0]
assert rec == {"backend": "gloo", "world": 2, "kernel_ms_per_rank": [10.0, 11.0], "devices": [0, 1], "gather_bytes": ls.batch * ls.n * 8, "gather_ms": 1.5, "share_gpu": False}, rec
if rank == 0:
    import json
    print("COLLECTIVE " + json.dumps(rec))
    full = problems.oracle_solve(ls)
    assert t == 2.0 and s == ls.batch
    assert x.shape == full["x"].shape and np.array_equal(x, full["x"])
    print("GLOO_OK")
else:
    assert x is None
dist.barrier(); dist.destroy_process_group()
'''


def test_shard_range_partitions():
    from optimal_control_problem_amd.sharding import shard_range
    for total in (1, 7, 8, 8192, 65536):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo(built, tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill(); out, _ = p.communicate()
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), outs
    assert "GLOO_OK" in outs[0]
    assert 'COLLECTIVE {"backend": "gloo", "world": 2' in outs[0]        # the object bench.py adds to its line on every N > 1 run
    assert sharding_single_process_has_no_record()


def sharding_single_process_has_no_record():
    from optimal_control_problem_amd import sharding
    return sharding.collective_record(None, 1.0, 0, 0.0, 0) is None and sharding.all_values(2.5, None) == [2.5]


def test_rank_that_cannot_join_exits_with_a_reason(built, tmp_path):
    """a rank whose rendezvous fails (nobody listens on the port) ends with exit code 3 and one line on stderr instead of hanging"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "lonely.py"
    script.write_text("import sys\nsys.path.insert(0, %r)\nfrom optimal_control_problem_amd import sharding\nsharding.INIT_TIMEOUT_S = 5\n"
                      "sharding.init_distributed(2, backend='gloo')\nprint('joined')\n" % ROOT)
    env = dict(os.environ, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 3 and "could not join the gloo group" in r.stderr and "joined" not in r.stdout, (r.returncode, r.stderr[-400:])


def test_launch_ranks_ends_a_hung_job(built, tmp_path):
    """the launcher's overall timeout: ranks that never finish (and ignore SIGTERM) are killed, the exit code says so"""
    hung = tmp_path / "hung.py"
    hung.write_text("import signal, time\nsignal.signal(signal.SIGTERM, signal.SIG_IGN)\ntime.sleep(120)\n")
    drv = tmp_path / "driver.py"
    drv.write_text("import sys\nsys.path.insert(0, %r)\nfrom optimal_control_problem_amd import sharding\n"
                   "raise SystemExit(sharding.launch_ranks(2, [%r]))\n" % (ROOT, str(hung)))
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, str(drv)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60, env=dict(os.environ, MPCQP_LAUNCH_TIMEOUT_S="2"))
    assert r.returncode == 124 and "still running" in r.stdout and time.time() - t0 < 30, (r.returncode, r.stdout)


def test_launch_ranks_starts_one_process_per_rank(built, tmp_path):
    """bench.py's own launcher (plain `python bench.py --gpus N`): N child processes with the rendezvous variables set, rank 0's
    stdout passed through, a failing rank reported in the exit code"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    drv = tmp_path / "driver.py"
    drv.write_text("import sys\nsys.path.insert(0, %r)\nfrom optimal_control_problem_amd import sharding\n"
                   "raise SystemExit(sharding.launch_ranks(2, [%r]))\n" % (ROOT, str(script)))
    r = subprocess.run([sys.executable, str(drv)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0 and "GLOO_OK" in r.stdout, r.stdout
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(30)\n")
    drv.write_text("import sys\nsys.path.insert(0, %r)\nfrom optimal_control_problem_amd import sharding\n"
                   "raise SystemExit(sharding.launch_ranks(2, [%r]))\n" % (ROOT, str(bad)))
    r = subprocess.run([sys.executable, str(drv)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
    assert r.returncode == 7
