"""bench.py --gpus N started as one plain process is its own launcher (child processes, never exec).  Without a GPU
the ranks cannot run, which is exactly the failure path: ONE JSON line with an error entry, non-zero status."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_start_with_two_ranks_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.device_count() > 0:
        import pytest
        pytest.skip("GPU present: covered by tests/test_gpu_bench_multirank.py")
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--log2n", "10", "--comm", "torch", "--watchdog-s", "60"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode != 0
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] is None and "child exit codes" in d["error"]
    assert "no GPU visible" in out.stderr          # the children's own diagnostics reach stderr


def test_launcher_is_not_entered_under_an_external_launcher():
    """WORLD_SIZE in the environment (torch.distributed.run started us): main() must take the rank path - seen here by
    its assertion on the world size, not by a second generation of children"""
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "--gpus 2 but WORLD_SIZE=4" in out.stderr
