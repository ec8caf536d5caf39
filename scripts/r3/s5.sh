#!/bin/bash
mkdir -p gpurun_out/s5; O=gpurun_out/s5
timeout -k 10 900 python -m pytest tests/test_encoder_gpu.py -x -q > $O/t_enc.txt 2>&1
tail -n 25 $O/t_enc.txt
