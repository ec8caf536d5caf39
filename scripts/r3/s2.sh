#!/bin/bash
# round-3 GPU session 2: sharded k-means halves, RCCL 1-rank groups, config1 gradient diagnostic, sharded k-means timing
mkdir -p gpurun_out/s2; O=gpurun_out/s2
timeout -k 10 400 python -m pytest tests/test_kmeans_gpu.py -x -q > $O/t_km.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_dist_gpu.py -x -q -k "kmeans or pipeline" > $O/t_dist.txt 2>&1
timeout -k 10 500 python bench.py --gpus 1 --steps 3 --warmup 1 --self-launch --force-dist --no-cpu-baseline --quick > $O/bench_dist.txt 2>&1
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --quick > $O/bench_plain.txt 2>&1
timeout -k 10 900 python scripts/r3/diag_cfg1.py 32 > $O/diag32.txt 2>&1
tail -n 3 $O/t_km.txt $O/t_dist.txt; tail -n 12 $O/diag32.txt
python - <<'PY'
import json
for f in ("bench_dist", "bench_plain"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/s2/{f}.txt") if l.startswith("{")][-1])
        s = d["secondary"]
        print(f, round(d["value"], 1), "km", s.get("value"), s.get("ms_per_iter"), s.get("error"))
    except Exception as e:
        print(f, "ERR", e)
PY
