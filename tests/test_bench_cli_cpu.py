"""bench.py must never report a line for a GPU count it did not run on (round-1 verdict: `--gpus 8` silently ran one rank
and printed n_gpus: 1).  No GPU needed: the refusal happens before any device is touched."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert "refusing" in r.stderr and r.stdout.strip() == ""


def test_self_spawn_command_line():
    """Without a launcher, --gpus N re-executes under torch.distributed.run with N ranks on 127.0.0.1."""
    sys.path.insert(0, ROOT)
    import bench
    captured = {}

    def fake_exec(path, argv, env):
        captured["argv"] = argv
        captured["env"] = env
        raise SystemExit(0)
    old, old_argv = os.execvpe, sys.argv
    env_backup = os.environ.pop("WORLD_SIZE", None)
    try:
        os.execvpe = fake_exec
        sys.argv = ["bench.py", "--gpus", "4", "--steps", "3"]
        try:
            bench.maybe_spawn(bench.parse_args(["--gpus", "4", "--steps", "3"]))
        except SystemExit:
            pass
    finally:
        os.execvpe, sys.argv = old, old_argv
        if env_backup is not None:
            os.environ["WORLD_SIZE"] = env_backup
    a = captured["argv"]
    assert "torch.distributed.run" in a and a[a.index("--nproc-per-node") + 1] == "4"
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and a[-4:] == ["--gpus", "4", "--steps", "3"]
    assert captured["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    # one GPU: no spawn
    assert bench.maybe_spawn(bench.parse_args(["--gpus", "1"])) is None
