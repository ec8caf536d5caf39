#!/bin/bash
# round 4, call 1: MFMA / VALU co-execution microbenchmark (table + PMC pass) and the baseline bench on the same box
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r4_micro; mkdir -p $O
# the binary is built in-tree (hipcc cross-compiles without a GPU) and is not part of the history
[ -x scripts/micro/mfma_valu_coexec ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/micro/mfma_valu_coexec scripts/micro/mfma_valu_coexec.hip || exit 1
./scripts/micro/mfma_valu_coexec > $O/table.txt 2>&1 || exit 1
cat $O/table.txt
export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc -- ./scripts/micro/mfma_valu_coexec pmc > $O/pmc_stdout.txt 2>&1 || echo "pmc pass failed"
find $O/pmc -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/pmc_counters.csv
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err || exit 1
tail -c 3000 $O/bench.json
