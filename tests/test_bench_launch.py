"""bench.py --gpus N without a launcher on the command line (no GPU needed: the children are not started here).

The driver's contract launches N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P bench.py --gpus N ...`; a plain `python bench.py --gpus N` must end up as exactly that, started as CHILD processes
by a parent that has not touched the GPU (/root/reference/main.py:285-290 runs one process per GPU too)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _launch(monkeypatch, argv, env):
    import bench
    calls = []

    class Done:
        returncode = 7

    monkeypatch.setattr(subprocess, "run", lambda cmd, env=None: (calls.append((cmd, env)), Done())[1])
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    try:
        bench._self_launch()
    except SystemExit as e:
        return calls, e.code
    return calls, None


def test_a_plain_multi_gpu_invocation_starts_its_own_ranks(monkeypatch):
    calls, code = _launch(monkeypatch, ["--gpus", "4", "--config", "3", "--steps", "5"], {})
    assert code == 7 and len(calls) == 1                              # the children's exit code is the parent's
    cmd, env = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--config", "3", "--steps", "5"]      # the ranks get the caller's own arguments
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.parametrize("argv,env", [(["--gpus", "1"], {}), ([], {}), (["--gpus=8"], {"WORLD_SIZE": "8"}), (["--gpus", "2"], {"WORLD_SIZE": "2"})])
def test_single_gpu_runs_and_launched_ranks_go_straight_on(monkeypatch, argv, env):
    calls, code = _launch(monkeypatch, argv, env)
    assert calls == [] and code is None


def test_the_equals_form_of_the_flag_counts_too(monkeypatch):
    calls, code = _launch(monkeypatch, ["--gpus=2", "--steps", "3"], {})
    assert code == 7 and calls[0][0][calls[0][0].index("--nproc-per-node") + 1] == "2"
