#!/bin/bash
# round-3 GPU session 6: Winograd wgrad parity + speed, pipelined forward kernel
mkdir -p gpurun_out/s6; O=gpurun_out/s6
timeout -k 10 300 python -m pytest tests/test_encoder_gpu.py -x -q -k "winograd" > $O/t_wino.txt 2>&1 || { tail -n 40 $O/t_wino.txt; exit 1; }
timeout -k 10 300 python scripts/bench_conv.py 32 "l" > $O/conv.txt 2>&1
for wgs in 512 2048; do SLIC_WINO_WGRAD_WGS=$wgs timeout -k 10 200 python scripts/bench_conv.py 32 "c2" 2>&1 | tail -n 1; done > $O/conv_wgs.txt 2>&1
for cfg in "1 1" "1 0" ; do set -- $cfg; SLIC_WINO=$1 SLIC_WINO_WGRAD=$2 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wino$1 wgrad$2', round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('final_loss'))"; done > $O/step_ab.txt 2>&1
tail -n 3 $O/t_wino.txt; cat $O/conv.txt $O/conv_wgs.txt $O/step_ab.txt
