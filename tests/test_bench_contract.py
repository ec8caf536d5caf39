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


def test_bench_line_self_verifying_distributed_fields():
    """round 5: the line of a run with a process group (`profiles/r05_bench_force_dist_one_rank.json`: `bench.py --gpus 1 --force-dist`,
    the code path of N = 2, 4, 8) carries what makes a multi-GPU run verify itself — how many ranks RCCL really reduced over, the
    library version, every rank's step time, what DistributedDataParallel did with the gradients, and the k-means row's exchange"""
    path = os.path.join(ROOT, "profiles", "r05_bench_force_dist_one_rank.json")
    d = json.load(open(path))
    assert "watchdog" not in d, "the secondary rows of this run were cut short by bench.py's watchdog: the line is a headline only"
    dist = d["distributed"]
    assert dist["initialised"] is True and dist["backend"] == "nccl"
    assert dist["rccl_ranks_seen"] == d["n_gpus"] == dist["world_size_env"] == len(dist["ms_per_step_per_rank"])
    assert dist["rccl_version"].count(".") == 2
    assert dist["ms_per_step_min"] <= dist["ms_per_step_max"] <= d["ms_per_step"] * 1.001
    ddp = dist["ddp"]
    assert ddp["buckets"] == len([b for b in ddp["bucket_sizes_bytes"].split(",") if b.strip()]) >= 1
    assert isinstance(ddp["gradient_as_bucket_view"], bool) and ddp["gradient_bytes"] == 34520896 * 4       # R3D-18: 34.52 M parameters
    sec = d["secondary"]
    ex = sec["exchange"]
    assert ex["kind"] in ("allreduce", "allgather", "oneshot") and ex["collectives_per_iteration"] == 1
    assert ex["payload_bytes_per_rank"] == (500 * 512 + 500 + 2) * 8                                      # fp64 [K D sums | K counts | n_changed]
    assert sec["weak_scaled"]["scaling"] == "weak" and sec["weak_scaled"]["n_gpus"] == d["n_gpus"]
    one = d.get("kmeans_oneshot_row_from_stderr")
    assert one and one["exchange"]["kind"] == "oneshot" and one["value"] > 0
