"""bench.py's output contract (one JSON line with the keys the driver reads) and
__graft_entry__.smoke(), exercised on the GPU tier with a short run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int,
            "ms_per_step": float, "higher_is_better": bool, "scaling": str, "dtype": str, "data": str,
            "config": dict, "roofline": dict, "cpu_baseline": dict}


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3",
                        "--prewarm", "100"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for key, typ in REQUIRED.items():
        assert key in out and isinstance(out[key], typ), key
    assert "vs_baseline" in out and out["vs_baseline"] is None  # BASELINE.md publishes no number
    assert out["n_gpus"] == 1 and out["steps"] == 12 and out["warmup"] == 3
    assert out["unit"] == "elements/s" and out["dtype"] == "u64" and out["higher_is_better"] is True
    assert "2^24" in out["config"]["workload"] and "model" not in out["config"]
    assert 1e9 < out["value"] < 1e12
    assert abs(out["value"] - (1 << 24) / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-6
    roof = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0.02 < roof["frac"] < 1.0
    assert roof["traffic"] is None or roof["traffic"] >= roof["algorithmic_bytes_per_launch"]
    cpu = out["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and cpu["value"] > 0


@pytest.mark.gpu
def test_smoke_entry_point():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_randomised_parity_stress_short():
    """tools/stress.py for 25 s: random prime / length / split / batch / direction / fused product
    against the oracle (a 7-minute run is kept in profiles/r01/stress_7min.txt)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress.py"), "25", "7"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
