#!/bin/bash
# round-3 GPU session 1: host facts, new parity tests, wgrad kernel A/B, tail-split A/B
mkdir -p gpurun_out/s1; O=gpurun_out/s1
{ nproc; free -g; python -c "import psutil; print(psutil.virtual_memory())"; } > $O/host.txt 2>&1
timeout -k 10 400 python -m pytest tests/test_finch_gpu.py tests/test_conv_plan_host.py -x -q > $O/t_finch.txt 2>&1
timeout -k 10 400 python -m pytest tests/test_encoder_gpu.py -q -k "conv_ or dgrad_ or tripletnet" > $O/t_conv.txt 2>&1
SLIC_WGRAD_KERNEL=1 timeout -k 10 300 python -m pytest tests/test_encoder_gpu.py -q -k "conv_fwd_dgrad_wgrad" > $O/t_wg1.txt 2>&1
SLIC_WGRAD_KERNEL=2 timeout -k 10 300 python -m pytest tests/test_encoder_gpu.py -q -k "conv_fwd_dgrad_wgrad" > $O/t_wg2.txt 2>&1
timeout -k 10 400 python -m pytest tests/test_encoder_gpu.py -q -k "ragged or config0 or tiny_encoder_train or full_size" > $O/t_enc.txt 2>&1
( time timeout -k 10 400 python -m pytest tests/test_encoder_gpu.py -q -k "config1" ) > $O/t_cfg1.txt 2>&1
for k in 0 1 2; do echo "wgrad kernel $k"; SLIC_WGRAD_KERNEL=$k WGONLY=1 python scripts/bench_conv.py 32; echo; done > $O/wg_ab.txt 2>&1
{ echo "tail on"; python scripts/bench_conv.py 32; echo "tail off"; SLIC_CONV_TAIL=0 python scripts/bench_conv.py 32; } > $O/conv_ab.txt 2>&1
for k in 0 1 2; do SLIC_WGRAD_KERNEL=$k python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wg$k', round(d['value'],1), round(d['ms_per_step'],2))"; done > $O/step_ab.txt 2>&1
SLIC_CONV_TAIL=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tail0', round(d['value'],1), round(d['ms_per_step'],2))" >> $O/step_ab.txt 2>&1
tail -3 $O/t_*.txt; cat $O/wg_ab.txt $O/step_ab.txt
