"""The one JSON line bench.py prints (the driver's contract): checked on the line committed under profiles/ (the bench itself needs a GPU)
and on bench.py's argument defaults."""
import json
import os
import re

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_committed_bench_line_has_the_contract_fields():
    j = json.loads(open(os.path.join(ROOT, "profiles", "r02_bench.json")).read())
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in j, k
    assert "registered points/sec" in base["metric"] and "registered points/sec" in j["metric"] and j["unit"] == "points/s"
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert "workload" in j["config"] and not any(k in j["config"] for k in ("model", "seq_len", "global_batch"))
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["unit"] == "GB/s" and r["peak"] == 8000.0
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["unit"] == j["unit"]
    # whole-job value and the per-step time are one measurement
    n_raw = j["config"]["n_raw"]
    assert abs(j["value"] - n_raw / (j["ms_per_step"] * 1e-3)) < 0.02 * j["value"]


def test_bench_defaults_are_the_drivers_assumptions():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'"--gpus", type=int, default=1\b', src)
    assert re.search(r'"--steps", type=int, default=\d+', src) and re.search(r'"--warmup", type=int, default=\d+', src)
    assert "oracle" not in src.split("def main")[0]  # nothing of oracle/ is imported at module level: only the cpu_baseline leg loads it
