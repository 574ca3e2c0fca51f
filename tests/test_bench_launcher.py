"""bench.py as its own launcher (CPU side): `python bench.py --gpus N` without torchrun spawns torch.distributed.run as a child
BEFORE importing torch or touching HIP; here (no GPU) the ranks exit with "needs a GPU" and the launcher hands that on."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_command_is_the_drivers_torchrun_line():
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "7"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7"]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="the GPU form of this is tests/test_bench_gpu.py::test_bench_launches_its_own_ranks")
def test_self_launch_spawns_ranks_and_forwards_their_failure():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "dblp", "--steps", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and "must be launched with" not in r.stderr
    assert r.stdout.strip() == ""          # no line without a measurement
