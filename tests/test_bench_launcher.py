"""bench.py --gpus N without a launcher: N rank processes are started through torch.distributed.run BEFORE the parent touches
the GPU. Checked on CPU: the command line, the environment, and that the children really start as ranks
(they stop at "needs a GPU" here, which is the loud no-fallback failure, with a non-zero exit code)."""
import os
import subprocess
import sys

from conftest import ROOT


def test_launcher_command(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = bench.parse()
    assert bench.launch_ranks(args) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and "--nnodes=1" in cmd
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_parent_never_touches_the_gpu_and_children_are_ranks():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-ops", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240, env=env)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "bench.py needs a GPU (no CPU fallback)" in r.stdout      # from the rank processes
