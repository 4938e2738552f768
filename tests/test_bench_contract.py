"""CPU: the bench line committed under profiles/ (produced by `python bench.py` on the MI355X box) carries every field
of the driver's contract, the roofline and cpu_baseline objects, and internally consistent numbers."""
import json
import os
import subprocess
import sys

from tests.conftest import ROOT


def _line():
    with open(os.path.join(ROOT, "profiles", "r1k_bench_line.json")) as f:
        return json.loads(f.read().strip().split("\n")[-1])


def test_bench_line_contract():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    frames = d["config"]["frames_per_step_per_gpu"] * d["steps"] * d["n_gpus"]
    assert abs(d["value"] - frames / (d["ms_per_step"] * d["steps"] / 1e3)) / d["value"] < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert abs(r["achieved"] - r["flop_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) / r["achieved"] < 1e-6
    assert r["traffic"] > r["hbm"]["algorithmic_bytes_per_launch"]  # measured L2-miss traffic, never below the algorithmic bytes
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["unit"] == d["unit"]
    lat = d["latency"]
    assert abs(lat["speedup_vs_cpu_1thread"] - lat["cpu_1thread_ms_per_frame"] / lat["single_frame_ms"]) < 1e-6
    assert lat["speedup_vs_cpu_1thread"] >= 50.0  # north-star latency target


def test_bench_cli_accepts_the_driver_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
