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


def test_a_rank_that_dies_makes_the_parent_exit_non_zero(tmp_path):
    """run_child: the job is a child process group; its failure is the parent's exit code, its output is relayed."""
    sys.path.insert(0, str(ROOT))
    import bench
    script = tmp_path / "job.py"
    script.write_text("import sys\nprint('rank 0 alive', flush=True)\nsys.exit(3)\n")
    rc = bench.run_child([sys.executable, str(script)], dict(os.environ), timeout_s=60)
    assert rc == 3


def test_a_job_that_hangs_is_killed_as_a_group_within_the_timeout(tmp_path):
    """A rank stuck in a collective whose peer is gone never returns: the parent kills the group it started (and only
    that group) and reports 124."""
    import time
    sys.path.insert(0, str(ROOT))
    import bench
    pidfile = tmp_path / "pids"
    script = tmp_path / "job.py"
    # a parent with a grandchild, both sleeping: the whole group must go
    script.write_text(
        "import os, subprocess, sys, time\n"
        f"p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])\n"
        f"open({str(pidfile)!r}, 'w').write(f'{{os.getpid()}} {{p.pid}}')\n"
        "time.sleep(600)\n")
    t0 = time.time()
    rc = bench.run_child([sys.executable, str(script)], dict(os.environ), timeout_s=3)
    assert rc == 124 and time.time() - t0 < 40
    time.sleep(0.5)
    for pid in map(int, pidfile.read_text().split()):
        alive = True
        try:
            os.kill(pid, 0)
            # a zombie still answers kill(0): read its state
            state = Path(f"/proc/{pid}/stat").read_text().split(")")[-1].split()[0]
            alive = state != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, f"process {pid} of the job's group survived"


def test_transport_decision_refuses_a_silent_fallback():
    """SCALE day: a job on the nccl backend whose RCCL communicator is missing on some rank must not time a host path
    (cutfemx_amd.dist.decide_transport; bench.py sets CFX_DIST_STRICT=1 unless CFX_REHEARSE=1)."""
    import pytest
    from cutfemx_amd.dist import TransportError, decide_transport
    assert decide_transport(True, 8, 8, strict=True) == "rccl"
    assert decide_transport(True, 8, 8, strict=False) == "rccl"
    assert decide_transport(False, 0, 2, strict=False) == "host-staged"      # a gloo rehearsal: host-staged by design
    assert decide_transport(False, 0, 2, strict=True) == "host-staged"
    # the library's communicator missing on a rank of an nccl job: the job's own nccl group carries the device buffers
    # (still RCCL over xGMI), strict or not ...
    assert decide_transport(True, 7, 8, strict=True, why="rank 3: ncclCommInitRank failed") == "rccl-torch"
    assert decide_transport(True, 0, 8, strict=False) == "rccl-torch"
    # ... and with that path switched off (CFX_DIST_TORCH_P2P=0) a strict job ends instead of timing a host path
    assert decide_transport(True, 7, 8, strict=False, why="rank 3: ncclCommInitRank failed", torch_p2p=False) == "host-staged"
    with pytest.raises(TransportError, match="7 of 8 ranks"):
        decide_transport(True, 7, 8, strict=True, why="rank 3: ncclCommInitRank failed", torch_p2p=False)


def test_scale_line_names_its_transport():
    sys.path.insert(0, str(ROOT))
    import bench

    class Comm:
        transport, rccl_ranks, fallback_reason = "rccl", 8, ""
    f = bench.scale_fields(Comm(), 8, [3.1, 3.0, 3.3, 3.2, 3.25, 3.1, 3.0, 2.9], 0.04)
    assert f["transport"] == "rccl" and f["rccl_ranks"] == 8 and f["world"] == 8
    assert f["slowest_rank_ms_per_step"] == 3.3 and len(f["per_rank_ms_per_step"]) == 8
    assert f["exchange_ms"] == 0.04 and f["imbalance"] > 1.0

    class Host:
        transport, rccl_ranks, fallback_reason = "host-staged", 0, "rehearsal"
    assert bench.scale_fields(Host(), 2, [1.0, 1.0], 0.5)["transport"] == "host-staged"


def _canned_record():
    """A full bench record of the shape `bench.main` builds (round 4's committed one), with the fields round 5 added."""
    import json
    rec = json.loads((ROOT / "profiles" / "r04_default_bench.json").read_text())
    rec["whole_step_roofline"].update(frac_survey_bytes=0.29, frac_moved_bytes=0.22)
    rec["detail_file"] = "gpurun_out/bench_detail.json"
    return rec


def test_the_contract_line_is_short_and_carries_roofline_and_cpu_baseline():
    """BENCH_r04.json.parsed was null: the one stdout line had grown to 20 kB.  The line the driver parses is now the
    contract fields + roofline + cpu_baseline + one-number summaries, under 4 kB whatever the legs report."""
    import json
    sys.path.insert(0, str(ROOT))
    import bench
    rec = _canned_record()
    line = bench.headline_line(rec)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT == 4096
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["value"] == rec["value"] and d["ms_per_step"] == rec["ms_per_step"]
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "DOF/s" and c["value"] > 0 and c["sample"]
    assert d["whole_step"] == {"frac_survey_bytes": 0.29, "frac_moved_bytes": 0.22}
    s = d["secondary"]
    assert s["config_p2_gyroid_256"]["sparsity"]["traffic_ratio"] > 1 and s["config_elasticity_share"]["matrix"]["frac"] > 0
    assert "NOT a measured" in s["projected_scaling"]["kind"]
    # legs that failed, exploded in size or are absent never break the line
    rec["config_p2_gyroid_256"] = {"error": "RuntimeError: " + "x" * 5000}
    rec["cpu_baseline"]["sample"] = "y" * 10000
    rec["config"]["workload"] = rec["config"]["workload"] * 3
    line = bench.headline_line(rec)
    assert len(line) < 4096 and json.loads(line)["roofline"]["frac"] > 0
    for k in ("config_128", "config_32", "moving_domain", "projected_scaling", "implicit_structured",
              "config_elasticity_share", "cpu_baseline_all_cores", "whole_step_roofline", "step_mode", "phases_ms"):
        rec.pop(k, None)
    d = json.loads(bench.headline_line(rec))
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0


def test_emit_prints_one_stdout_line_and_writes_the_detail_file(tmp_path, monkeypatch, capsys):
    import json
    sys.path.insert(0, str(ROOT))
    import bench
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    rec = _canned_record()
    bench.emit(rec)
    cap = capsys.readouterr()
    lines = cap.out.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["detail"] == "gpurun_out/bench_detail.json"
    full = json.loads((tmp_path / "gpurun_out" / "bench_detail.json").read_text())
    assert full["kernels"] and full["value"] == d["value"]
    assert cap.err.startswith("# bench_detail: ")
