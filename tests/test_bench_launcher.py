"""`python3 bench.py --gpus N` with WORLD_SIZE unset starts its own N ranks (the driver's SCALE command shape)."""
import argparse
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _args(**kw):
    base = dict(gpus=8, steps=5, warmup=2, n=512, order=4, cpu_n=256, no_cpu=False, no_secondary=False, cpu_worker=None)
    base.update(kw)
    return argparse.Namespace(**base)


def test_launcher_command_line():
    sys.path.insert(0, str(ROOT))
    import bench
    cmd = bench.launcher_command(_args(), port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(str(ROOT / "bench.py"))
    tail = cmd[i + 1:]
    assert tail[tail.index("--gpus") + 1] == "8" and tail[tail.index("--steps") + 1] == "5"
    assert tail[tail.index("--warmup") + 1] == "2" and tail[tail.index("--mesh") + 1] == "512"
    assert "--no-cpu" not in tail
    assert "--no-cpu" in bench.launcher_command(_args(no_cpu=True, gpus=2), port=1)
    # a free port is picked when none is given
    port = int(bench.launcher_command(_args(gpus=2))[9])
    assert 1024 < port < 65536


def test_launcher_starts_child_ranks_and_propagates_the_exit_code():
    """No GPU here: both ranks stop with 'needs an MI355X' -- which proves the parent started them as children
    (without importing torch itself), relayed their output and handed their failure on."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the ranks would run the real benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--mesh", "8", "--no-cpu", "--no-secondary"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "needs an MI355X" in p.stderr
