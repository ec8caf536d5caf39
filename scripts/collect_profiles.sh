#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash scripts/collect_profiles.sh'): default bench, rocprofv3 kernel stats, two PMC passes.
# Outputs land in gpurun_out/prof_r06/ (merged back); `python scripts/collect_profiles.py gpurun_out/prof_r06 profiles r05` files them.
set -u
R="$(pwd)"
S="$R/gpurun_out/prof_r06"
rm -rf "$S"; mkdir -p "$S"
cd /tmp && export TMPDIR=/tmp
timeout 600 python "$R/bench.py" > "$S/default.log" 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$S/stats" -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$S/stats.log" 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$S/pmc_fetch" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$S/pmc_fetch.log" 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$S/pmc_write" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$S/pmc_write.log" 2>&1 < /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$S/pmc_sq" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$S/pmc_sq.log" 2>&1 < /dev/null
# keep the merge-back small: per-kernel csvs only
find "$S" -name "*agent_info.csv" -delete
# the other rows: k-means Lloyd iterations, retrieval / NCE / NT-Xent, reference-shaped fit_cluster + FINCH
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$S/stats_kmeans" -- python "$R/scripts/bench_kmeans.py" > "$S/stats_kmeans.log" 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$S/stats_rows" -- python "$R/scripts/bench_rows.py" > "$S/stats_rows.log" 2>&1 < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$S/stats_fit" -- python "$R/scripts/bench_fit_cluster.py" > "$S/stats_fit.log" 2>&1 < /dev/null
timeout 300 python "$R/scripts/bench_rows.py" > "$S/rows.log" 2>&1 < /dev/null
timeout 300 python "$R/scripts/bench_topk_k.py" > "$S/topk_k.log" 2>&1 < /dev/null
timeout 300 python "$R/scripts/bench_conv.py" 32 > "$S/conv_shapes.log" 2>&1 < /dev/null
# round 5 rows: retrieval per kernel (collect vs streaming), the one-rank distributed path with the self-verifying fields and the one-shot
# exchange row (stderr), the B = 8 step's kernel statistics, the planar-layout diagnostic of the dominant kernel
(cd "$R" && timeout 300 bash scripts/r5/prof_topk.sh) > "$S/topk_kernels.log" 2>&1 < /dev/null
timeout 400 python "$R/bench.py" --gpus 1 --force-dist --quick --no-cpu-baseline --steps 5 --warmup 2 > "$S/force_dist.log" 2> "$S/force_dist.err" < /dev/null
grep '"row": "kmeans_oneshot"' "$S/force_dist.err" > "$S/force_dist_oneshot.json"
(cd "$R" && timeout 300 bash scripts/r5/prof_b8.sh) > "$S/b8_kernels.log" 2>&1 < /dev/null
# round 6 rows: the persistent kernel against the one-block kernel (per launch, per shape, whole steps), DistributedDataParallel at one rank
(cd "$R" && timeout 500 bash scripts/r6/ab_persist.sh) > "$S/persist_ab.log" 2>&1 < /dev/null
(cd "$R" && timeout 500 bash scripts/r6/ab_ddp.sh) > "$S/ddp_ab.log" 2>&1 < /dev/null
(cd "$R" && timeout 300 bash scripts/r6/ddp_sequence.sh) > "$S/ddp_seq.log" 2>&1 < /dev/null
rm -f "$R/gpurun_out/r5_b8_kernel_trace.csv"
find "$S" -name "*kernel_trace.csv" -path "*stats_*" -delete
find "$S" -name "*agent_info.csv" -delete
du -sh "$S"; ls "$S"
grep "^{" "$S/default.log" | cut -c1-200
