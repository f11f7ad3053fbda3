"""bench.py host logic that needs no GPU: `--gpus N` without a launcher re-executes itself under torch.distributed.run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_n_builds_the_launcher_command(monkeypatch):
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    calls = []

    def fake_call(cmd, env=None):
        calls.append((cmd, env))
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "3"])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert len(calls) == 1
    cmd, env = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    i = cmd.index("--master-addr")
    assert cmd[i + 1] == "127.0.0.1" and cmd[i + 2] == "--master-port" and 0 < int(cmd[i + 3]) < 65536
    assert os.path.samefile(cmd[i + 4], os.path.join(ROOT, "bench.py"))
    assert cmd[i + 5:] == ["--gpus", "4", "--steps", "7", "--warmup", "3"]
    assert env.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
