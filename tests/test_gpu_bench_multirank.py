"""bench.py with MORE THAN ONE RANK, before the driver runs it on eight GPUs.

A one-GPU box cannot host two RCCL ranks (RCCL refuses a device twice), so:
 * two ranks of bench.py itself, launched exactly as the driver launches them (torch.distributed.run), with
   --dist-backend gloo: both ranks share the GPU, the partial points travel through host memory - every line of
   the world > 1 path runs (cyclic shard inputs, the collective branch of run_steps with batch 3 and three
   launches in flight, the exchange inside the C library's callback transport, the cross-rank exponent-identity
   check, the sharded prover on two blocks with the block-local generator fold, max-over-ranks timing);
 * one rank with the real RCCL communicator (--force-collective): ncclAllGather on the MSM's stream and the
   sharded round context over RCCL.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def last_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip().startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


def test_two_ranks_of_bench_py_over_gloo():
    env = dict(os.environ, VMPC_P4_JUMP_MIN_LOG2="9", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "2", "--log2n", "12",
           "--dist-backend", "gloo", "--sharded-log2n", "12", "--watchdog-s", "600"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    d = last_json_line(out.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 7 and d["checked"] is True and "error" not in d
    assert d["config"]["total_terms"] == 2 << 12 and d["config"]["commitments_per_launch"] == 3
    assert d["config"]["launches_in_flight"] == 3 and "callback" in d["config"]["collective"]
    assert abs(d["value"] - 2 * (1 << 12) * 7 / (d["ms_per_step"] * 7e-3)) / d["value"] < 1e-6
    sp = d["ac20_n2^12_sharded"]
    assert "error" not in sp, sp
    assert sp["blocks"] == 2 and sp["ranks_agree"] is True and sp["verified"] is True
    assert sp["rounds_in"].startswith("libvmpc_hip")


def test_two_ranks_started_plain_no_launcher():
    """`python3 bench.py --gpus 2 ...` as ONE plain process - the way the N = 1 line is started - must start its ranks
    itself (child processes), relay rank 0's line as the last line of stdout and say what the communicator saw."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env.update(VMPC_P4_JUMP_MIN_LOG2="9", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--log2n", "12", "--dist-backend", "gloo", "--sharded-log2n", "11", "--config4-log2n", "13",
           "--watchdog-s", "600"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # ONE line on stdout: rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["checked"] is True and "error" not in d
    assert d["config"]["comm"] == {"kind": "callback", "world": 2, "communicators": 3, "source": "vmpc_comm_info"}
    assert d["config"]["launched_by"].startswith("bench.py itself")
    c4 = d["msm_config4_n2^13_over_2_gpus"]
    assert "error" not in c4, c4
    assert c4["checked"] is True and c4["total_terms"] == 1 << 13 and c4["terms_per_gpu"] == 1 << 12
    assert c4["scalar_mults_per_s_prepared"] > 0
    for key in ("ac20_n2^11_sharded", "ac20_n2^12_sharded_weak_scaling"):
        sp = d[key]
        assert "error" not in sp, (key, sp)
        assert sp["blocks"] == 2 and sp["ranks_agree"] is True and sp["verified"] is True


def test_plain_start_reports_a_failing_rank():
    """a rank that cannot start (here: more ranks than the nccl backend has GPUs for) ends the launcher with a
    non-zero status and ONE JSON line that carries an error entry - no hang, no traceback as the last line"""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    import torch
    n = torch.cuda.device_count() + 1
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
           "--log2n", "10", "--comm", "torch", "--no-sharded-prove", "--config4-log2n", "0", "--watchdog-s", "120"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode != 0
    d = last_json_line(out.stdout)
    assert d["value"] is None and "error" in d and d["n_gpus"] == n


def test_eight_ranks_of_bench_py_over_gloo():
    """the world size the driver's scaling run ends with: eight ranks (all on this one GPU), eight cyclic shards of
    one commitment, the sharded prover on eight blocks of 512 with a block-local fold of 2^5 / 8 = 4 strides"""
    env = dict(os.environ, VMPC_P4_JUMP_MIN_LOG2="8", HSA_ENABLE_IPC_MODE_LEGACY="0", VMPC_MSM_SLOTS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "1", "--log2n", "11",
           "--dist-backend", "gloo", "--sharded-log2n", "12", "--config4-log2n", "15", "--watchdog-s", "900"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    d = last_json_line(out.stdout)
    assert d["n_gpus"] == 8 and d["checked"] is True and "error" not in d
    assert d["config"]["total_terms"] == 8 << 11
    assert d["config"]["comm"]["world"] == 8 and d["config"]["comm"]["kind"] == "callback"
    sp = d["ac20_n2^12_sharded"]
    assert "error" not in sp, sp
    assert sp["blocks"] == 8 and sp["ranks_agree"] is True and sp["verified"] is True
    # BASELINE config 4 at a reduced size: one 2^15-term commitment as eight cyclic shards of 2^12
    c4 = d["msm_config4_n2^15_over_8_gpus"]
    assert "error" not in c4, c4
    assert c4["checked"] is True and c4["terms_per_gpu"] == 1 << 12 and c4["n_gpus"] == 8
    # weak scaling of the sharded prover: 2^12 generators per rank -> N = 2^15 on eight blocks
    sw = d["ac20_n2^15_sharded_weak_scaling"]
    assert "error" not in sw, sw
    assert sw["blocks"] == 8 and sw["ranks_agree"] is True and sw["verified"] is True


def test_one_rank_over_rccl_force_collective():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(free_port()))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "1", "--log2n", "14",
           "--force-collective", "--sharded-prove", "--sharded-log2n", "14", "--no-cpu-baseline", "--no-prove",
           "--watchdog-s", "600"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    d = last_json_line(out.stdout)
    assert d["n_gpus"] == 1 and d["checked"] is True and "error" not in d
    assert "rccl" in d["config"]["collective"], d["config"]
    assert d["config"]["comm"]["kind"] == "rccl" and d["config"]["comm"]["world"] == 1      # RCCL's own count
    # the slot-per-communicator pattern of the multi-GPU run under REAL RCCL: three communicators,
    # one per commitment slot = one per stream, and the pipelined step loop (three launches in flight, each a pass of
    # three commitments followed by ncclAllGather + ordered add on ITS stream) running through all of them
    assert d["config"]["comm"]["communicators"] == 3 and d["config"]["launches_in_flight"] == 3
    assert d["config"]["commitments_per_launch"] == 3 and "all_gather" in d["config"]["collective"]
    sp = d["ac20_n2^14_sharded"]
    assert "error" not in sp, sp
    assert sp["transport"] == "rccl" and sp["verified"] is True and sp["rounds_in"].startswith("libvmpc_hip")


def test_torch_distributed_fallback_path_one_rank():
    """--comm torch: the exchange through torch.distributed (the path taken when the native communicator cannot
    be created on some rank)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(free_port()))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--log2n", "13",
           "--force-collective", "--comm", "torch", "--sharded-prove", "--sharded-log2n", "12",
           "--no-cpu-baseline", "--no-prove", "--watchdog-s", "600"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    d = last_json_line(out.stdout)
    assert d["checked"] is True and "torch.distributed" in d["config"]["collective"]
    sp = d["ac20_n2^12_sharded"]
    assert "error" not in sp and sp["verified"] is True and sp["rounds_in"].startswith("python")
