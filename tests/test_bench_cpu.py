"""CPU: bench.py's launcher logic (no GPU here): `--gpus N` without a torchrun environment must start the ranks itself or
fail loudly -- never print a one-GPU line for an N-GPU request (round-1 defect)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=300)


def test_more_gpus_than_visible_fails_loudly():
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "1"])
    assert r.returncode != 0
    assert "--gpus 8 requested but only" in r.stderr
    assert "\"metric\"" not in r.stdout


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in r.stderr
