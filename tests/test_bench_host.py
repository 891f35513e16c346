"""Host-side pieces of bench.py that need no GPU."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bare_gpus_n_becomes_a_torchrun_job(monkeypatch):
    """`python bench.py --gpus 4` without WORLD_SIZE replaces itself by the documented launcher command (one rank per GPU,
    rendezvous on 127.0.0.1, the same bench arguments)."""
    b = _bench()
    seen = {}

    def fake_execve(path, argv, env):
        seen.update(path=path, argv=list(argv), env=dict(env))
        raise SystemExit(0)

    monkeypatch.setattr(os, "execve", fake_execve)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        b.main()
    except SystemExit:
        pass
    a = seen["argv"]
    assert seen["path"] == sys.executable and a[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in a and "--nproc-per-node=4" in a
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(a[a.index("--master-port") + 1]) < 65536
    i = a.index(os.path.join(ROOT, "bench.py"))
    assert a[i + 1:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"


def test_ldlt_chain_length():
    b = _bench()
    assert b.ldlt_chain(12000, 2597) == 73 + 42
    assert b.ldlt_chain(640, 600) == 10          # too short for the two-ended form
