"""CPU: `bench.py --gpus N` reaches N ranks from the plain command line (no launcher), agrees with a launcher when there is
one, and refuses what it cannot honour.  No GPU is touched: --launch-check stops after the process group is formed."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_plan_decisions():
    import bench
    assert bench.launch_plan(1, {}, [], 1) == ("run", 1)
    action, cmd = bench.launch_plan(4, {}, ["--gpus", "4", "--steps", "3"], 8)
    assert action == "spawn"
    assert "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert "torch.distributed.run" in cmd
    # under a launcher: the world is the launcher's, --gpus has to agree with it
    assert bench.launch_plan(8, {"WORLD_SIZE": "8"}, ["--gpus", "8"], 8) == ("run", 8)
    assert bench.launch_plan(1, {"WORLD_SIZE": "2"}, [], 2) == ("run", 2)          # --gpus not given: follow the launcher
    with pytest.raises(SystemExit):
        bench.launch_plan(2, {"WORLD_SIZE": "4"}, ["--gpus", "2"], 8)
    # fewer devices than ranks: refused loudly (no device sharing, no CPU fallback) ...
    with pytest.raises(SystemExit):
        bench.launch_plan(8, {}, ["--gpus", "8"], 1)
    with pytest.raises(SystemExit):
        bench.launch_plan(2, {"WORLD_SIZE": "2"}, ["--gpus", "2"], 1)
    # ... except for the explicit test switch that lets ranks share a device
    assert bench.launch_plan(2, {"MJH_BENCH_DIST_BACKEND": "gloo"}, ["--gpus", "2"], 1)[0] == "spawn"


def test_gpus_flag_starts_that_many_ranks_without_a_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["ranks"] == [0, 1]
