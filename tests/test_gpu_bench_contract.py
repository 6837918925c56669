"""bench.py prints ONE JSON line with the fields the driver reads (small size, few steps)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                          "--log2n", "14", "--no-cpu-baseline", "--no-prove"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["value"] > 0 and abs(d["value"] - (1 << 14) * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    # the result of the timed steps is checked (exponent identity) and the line says how to read `value`
    assert d["checked"] is True and "error" not in d
    t = d["config"]["timing"]
    assert t["throughput_ms_per_commitment"] > 0 and t["latency_ms_one_commitment_alone"] > 0
    # the headline is the median of five back-to-back timed regions; min and max are reported beside it
    assert t["repeats"] == 5 and len(t["ms_per_step_of_each_repeat"]) == 5
    assert t["ms_per_step_min"] <= d["ms_per_step"] <= t["ms_per_step_max"]
    assert sorted(t["ms_per_step_of_each_repeat"])[2] == pytest.approx(d["ms_per_step"], rel=1e-3)
    assert d["config"]["variable_base_scalar_mults_per_s"] > 0 and "distinct" in d["config"]["scalar_vectors"]
    assert r["alu"]["frac"] > 0 and r["traffic_source"]


def test_phase_pipelined_bucket_stream_gives_the_same_commitments():
    """the experimental phase pipeline (VMPC_EXPERIMENTAL=1 VMPC_BUCKET_STREAM=1: every slot's bucket stage on one shared
    low-priority stream as a persistent launch, include/vmpc.h vmpc_ctx_set_bucket_stream) - measured slower and off by
    default, but its ordering by events must hold: the timed commitments are checked by the exponent identity"""
    env = dict(os.environ, VMPC_EXPERIMENTAL="1", VMPC_BUCKET_STREAM="1", VMPC_PIPE_BUCKET_WGS_PER_CU="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "9", "--warmup", "3", "--repeats", "2",
                          "--log2n", "15", "--no-cpu-baseline", "--no-prove"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert d["checked"] is True and "error" not in d and d["config"]["launches_in_flight"] == 3
    assert d["config"]["timing"]["repeats"] == 2 and len(d["config"]["timing"]["ms_per_step_of_each_repeat"]) == 2
