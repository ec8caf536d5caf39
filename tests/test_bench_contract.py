"""The JSON line bench.py prints (the copy kept under profiles/) carries every field the driver's contract names."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_default_run.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == base["metric"] and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and not base.get("published")            # no published number for this metric
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    # value = clips of all ranks / max-over-ranks time
    assert abs(d["value"] - d["n_gpus"] * d["config"]["global_batch"] / d["n_gpus"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
