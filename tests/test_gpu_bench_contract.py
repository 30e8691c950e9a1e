"""bench.py prints ONE JSON line with the fields the driver reads (small batch so the test stays short)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract(dev):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["unit"] == "images/sec"
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["dtype"] == "bf16" and out["data"] == "synthetic"
    assert "workload" in out["config"] and "model" not in out["config"]
    assert out["value"] > 0 and abs(out["value"] - 8 * 1000.0 / out["ms_per_step"]) / out["value"] < 0.02
    ro = out["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in ro, k
    assert ro["bound"] in ("hbm", "mfma") and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3


def test_bench_stdout_is_one_json_line_on_the_data_parallel_path(dev):
    """YH_FORCE_DP=1: the RCCL communicator comes up (its version banner must not land on stdout) and the step runs with the bucket
    hooks; the line stays the only thing on stdout — what the driver parses for N > 1"""
    env = dict(os.environ, YH_FORCE_DP="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline",
                        "--no-roofline"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[:500]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0
