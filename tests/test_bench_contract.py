"""The driver's contract with bench.py: one JSON line on stdout with the agreed keys (GPU box only; a reduced
layer so the check takes seconds -- the numbers of record come from the default invocation)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--n", "256", "--c", "512", "--m", "512", "--cpu-sample", "32"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    for key in ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]:
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["value"] > 0 and out["ms_per_step"] > 0 and out["vs_baseline"] is None and "workload" in out["config"]
    roof = out["roofline"]
    for key in ["bound", "achieved", "peak", "unit", "frac", "traffic"]:
        assert key in roof, key
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cpu = out["cpu_baseline"]
    for key in ["value", "unit", "cores", "kind", "sample"]:
        assert key in cpu, key
    assert cpu["kind"] in ("port", "reference") and cpu["value"] > 0
    assert out["parity_sample"]["neurons_with_index_mismatch"] == 0
