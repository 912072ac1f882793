"""bench.py started as the driver starts it for N > 1 but WITHOUT a launcher: the command it re-executes itself
with (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) and the device-sharing fallback."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def test_self_launch_command_one_rank_per_gpu():
    cmd, env = bench.self_launch_command(["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, 8, port=29555)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert env["MASTER_ADDR"] == "127.0.0.1" and "IDTO_BENCH_VISIBLE_GPUS" not in env


def test_self_launch_shares_devices_when_fewer_are_visible():
    cmd, env = bench.self_launch_command(["--gpus", "2"], 2, 1, port=1)
    assert "--nproc-per-node=2" in cmd and env["IDTO_BENCH_VISIBLE_GPUS"] == "1"
    _, env0 = bench.self_launch_command(["--gpus", "2"], 2, 0, port=1)   # no device at all: still one "slot"
    assert env0["IDTO_BENCH_VISIBLE_GPUS"] == "1"


def test_free_port_is_chosen():
    cmd, _ = bench.self_launch_command(["--gpus", "2"], 2, 2)
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
