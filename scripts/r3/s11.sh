#!/bin/bash
mkdir -p gpurun_out/s11; O=gpurun_out/s11
timeout -k 10 300 python -m pytest tests/test_kmeans_gpu.py -x -q > $O/t_km.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_dist_gpu.py -x -q > $O/t_dist.txt 2>&1
timeout -k 10 500 python bench.py --gpus 1 --steps 3 --warmup 1 --self-launch --force-dist --no-cpu-baseline --quick > $O/bench_dist.txt 2>&1
SLIC_KMEANS_COMM=torch timeout -k 10 500 python bench.py --gpus 1 --steps 3 --warmup 1 --self-launch --force-dist --no-cpu-baseline --quick > $O/bench_dist_torch.txt 2>&1
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --quick > $O/bench_plain.txt 2>&1
tail -n 3 $O/t_km.txt $O/t_dist.txt
python - <<'PY'
import json
for f in ("bench_dist", "bench_dist_torch", "bench_plain"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/s11/{f}.txt") if l.startswith("{")][-1])
        s = d["secondary"]
        print(f, round(d["value"], 1), "km", s.get("value"), s.get("ms_per_iter"), s.get("error"))
    except Exception as e:
        print(f, "ERR", e)
PY
