#!/bin/bash
mkdir -p gpurun_out/s10; O=gpurun_out/s10
timeout -k 10 300 python -m pytest tests/test_encoder_gpu.py -x -q -k "winograd" > $O/t_wino.txt 2>&1 || { tail -n 40 $O/t_wino.txt; exit 1; }
timeout -k 10 300 python scripts/bench_conv.py 32 "l" > $O/conv.txt 2>&1
for cfg in "auto" "1"; do SLIC_WGRAD_STREAM=$cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad_stream=$cfg', round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('final_loss'))"; done > $O/step_ab.txt 2>&1
tail -n 3 $O/t_wino.txt; cat $O/conv.txt $O/step_ab.txt
