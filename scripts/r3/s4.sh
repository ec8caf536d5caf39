#!/bin/bash
mkdir -p gpurun_out/s4; O=gpurun_out/s4
SLIC_WINO=1 timeout -k 10 500 python scripts/r3/diag_cfg1.py 4 > $O/diag4_wino1.txt 2>&1
SLIC_WINO=0 timeout -k 10 500 python scripts/r3/diag_cfg1.py 4 > $O/diag4_wino0.txt 2>&1
SLIC_WINO=1 timeout -k 10 500 python scripts/r3/diag_cfg1.py 4 > $O/diag4_wino1b.txt 2>&1
tail -n 22 $O/diag4_wino1.txt $O/diag4_wino0.txt; tail -n 12 $O/diag4_wino1b.txt
