#!/bin/bash
mkdir -p gpurun_out/s12; O=gpurun_out/s12
timeout -k 10 300 python -m pytest tests/test_kmeans_gpu.py -x -q > $O/t_km.txt 2>&1
timeout -k 10 300 python -m pytest tests/test_retrieval_nce_gpu.py -x -q > $O/t_nce.txt 2>&1
timeout -k 10 300 python -m pytest tests/test_train_loop_gpu.py tests/test_dist_gpu.py -x -q -k "not ddp and not bench" > $O/t_loop.txt 2>&1
timeout -k 10 300 python scripts/bench_kmeans.py > $O/km.txt 2>&1
timeout -k 10 300 python scripts/fuzz_kernels.py > $O/fuzz.txt 2>&1
timeout -k 10 300 python scripts/bench_rows.py > $O/rows.txt 2>&1
tail -n 3 $O/t_km.txt $O/t_nce.txt $O/t_loop.txt; tail -n 8 $O/km.txt; tail -n 5 $O/fuzz.txt; tail -n 12 $O/rows.txt
