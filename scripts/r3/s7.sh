#!/bin/bash
mkdir -p gpurun_out/s7; O=gpurun_out/s7
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/t_all.txt 2>&1
tail -n 15 $O/t_all.txt
