"""bench.py host logic that needs no GPU: the self-launch of N ranks (`python bench.py --gpus N` with WORLD_SIZE unset
must start N ranks itself, as a child process, before anything touches the GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_self_launch_builds_torchrun_child_command():
    cmd = _run(["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--print-launch"])
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(BENCH)
    tail = cmd[i + 1:]
    assert tail == ["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1"], tail  # the caller's arguments, unchanged


def test_no_relaunch_inside_a_rank_or_for_one_gpu():
    # a rank started by the driver's torch.distributed.run (WORLD_SIZE set) runs the benchmark itself
    cmd = _run(["--gpus", "8", "--print-launch"], {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3"})
    assert "torch.distributed.run" not in cmd and cmd[1] == BENCH
    cmd = _run(["--gpus", "1", "--print-launch"])
    assert "torch.distributed.run" not in cmd


def test_parent_of_a_self_launch_does_not_import_torch():
    # the launching parent must not initialise the GPU; the simplest guarantee is that it never imports torch at all
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--print-launch'];\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit:\n    pass\n"
            "assert 'torch' not in sys.modules, 'parent imported torch'\n" % BENCH)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
