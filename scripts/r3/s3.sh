#!/bin/bash
# round-3 GPU session 3: Winograd F(4,3) kernel — parity, then speed; config1 test with the branch-aware head gate
mkdir -p gpurun_out/s3; O=gpurun_out/s3
timeout -k 10 300 python -m pytest tests/test_encoder_gpu.py -x -q -k "winograd" > $O/t_wino.txt 2>&1 || { tail -n 40 $O/t_wino.txt; exit 1; }
timeout -k 10 300 python scripts/bench_conv.py 32 "l" > $O/conv.txt 2>&1
SLIC_WINO_STAGES=2 timeout -k 10 300 python scripts/bench_conv.py 32 "c2" > $O/conv_st2.txt 2>&1
for wv in 1 0; do SLIC_WINO=$wv timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wino$wv', round(d['value'],1), round(d['ms_per_step'],2), d['config'].get('final_loss'))"; done > $O/step_ab.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_encoder_gpu.py -q -k "config1 or config0 or ragged or tiny_encoder_train or full_size" > $O/t_enc.txt 2>&1
tail -n 3 $O/t_wino.txt; cat $O/conv.txt $O/conv_st2.txt $O/step_ab.txt; tail -n 15 $O/t_enc.txt
