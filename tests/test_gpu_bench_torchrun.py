"""bench.py's multi-rank path, EXECUTED: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` as the driver
launches it, on the one GPU of the test box.  RCCL refuses two ranks on one device (tools/rccl_same_device_probe.py), so the
all-reduce goes through the library's caller-supplied transport entry point with a host-staged torch.distributed (gloo)
all-reduce (`--transport gloo --same-device`): everything else -- torchrun rendezvous on 127.0.0.1, the shard of every rank, the
replicated LM driver, barriers, max-over-ranks timing, rank 0's JSON line -- is the code the 8-GPU run executes."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + text[-2000:])


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_bench_two_ranks_through_torchrun_equals_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "3", "--warmup", "1", "--config", "C2", "--no-cpu-baseline", "--no-visual", "--no-front-end"]
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + common, cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(free_port()), "bench.py", "--gpus", "2", "--transport", "gloo",
                          "--same-device"] + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert two.returncode == 0, two.stderr[-3000:]
    a, b = last_json(one.stdout), last_json(two.stdout)
    assert b["n_gpus"] == 2 and b["steps"] == 3 and b["value"] > 0 and b["scaling"] == "strong"
    assert "2 rank(s)" in b["config"]["sharding"] and b["stage_ms"]["allreduce"] > 0
    assert a["config"]["n_factors"] == b["config"]["n_factors"]
    assert b["config"]["n_pairs_local"] < a["config"]["n_pairs_local"]           # rank 0 holds its shard's pairs only
    # the same LM steps on the same problem: accepted / evaluated counts and the cost after the last step
    # (`value` and config.*_in_timed_steps are the MEDIAN region's -- which region that is depends on the clock; the per-region
    # lists and the cost after the very last step do not)
    for k in ("evals_by_region", "accepted_by_region", "lm_runs_started_by_region"):
        assert a["timing"][k] == b["timing"][k], k
    assert a["timing"]["regions"] == 5 and len(a["timing"]["ms_per_step_by_region"]) == 5
    ca, cb = a["config"]["last_cost"], b["config"]["last_cost"]
    assert abs(ca - cb) <= 1e-9 * abs(ca), (ca, cb)


@pytest.mark.gpu
def test_bare_bench_with_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's one-GPU command with another N) must
    start the two ranks itself instead of exiting: same JSON line as the explicit torch.distributed.run form."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--steps", "2", "--warmup", "1", "--config", "C2", "--no-cpu-baseline", "--no-visual", "--no-front-end"]
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--transport", "gloo", "--same-device"] + common, cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    b = last_json(r.stdout)
    assert b["n_gpus"] == 2 and b["steps"] == 2 and b["value"] > 0
    assert "2 rank(s)" in b["config"]["sharding"] and b["stage_ms"]["allreduce"] > 0
