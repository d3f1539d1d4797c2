"""The driver reads ONE JSON line from `python bench.py --gpus N --steps K --warmup W`: this test runs the
N = 1 command as the driver does (a subprocess, its own process and library load) with a short K and checks
the line's keys, types and internal consistency -- the contract of the task statement, section 4."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_conforming_json_line(gpu_device):
    env = dict(os.environ)
    env.pop("RAGFIN_LIB", None)          # the product library, as the driver runs it
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "5",
                        "--no-configs"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines            # exactly one line on stdout
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int),
                     ("warmup", int), ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str),
                     ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert key in d and isinstance(d[key], typ), key
    assert "vs_baseline" in d and d["vs_baseline"] is None        # BASELINE.md holds no published number for this metric
    assert d["n_gpus"] == 1 and d["steps"] >= 30 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["scaling"] in ("weak", "strong") and "workload" in d["config"] and "model" not in d["config"]
    assert d["data"].startswith("synthetic") and d["dtype"] == "f16"
    # value = units processed / time of the timed region
    assert abs(d["value"] - 64 / (d["ms_per_step"] * 1e-3)) / d["value"] < 0.01
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.3 < rf["frac"] < 1.0
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0
    # the oracle check of the same run
    assert d["recall_at_10"] == 1.0 and d["ids_ranks_exact"] is True and d["checked_queries"] == 64
