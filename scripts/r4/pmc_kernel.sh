#!/bin/bash
# round 4: memory-path counters of the conv kernels at one shape (GPU box):  bash scripts/r4/pmc_kernel.sh SHAPE [LIBNAME]
#   one rocprofv3 --pmc pass per counter group (kernel trace only), mean per launch and kernel -> gpurun_out/r4_pmc_<shape><lib>.txt
cd "$(dirname "$0")/../.."
R=$PWD; SH=${1:-l1}; LIB=${2:-}
O=$R/gpurun_out/r4_pmc_${SH}${LIB}; rm -rf $O; mkdir -p $O
[ -n "$LIB" ] && export SLIC_LIB_PATH=$R/video_similarity_search_amd/csrc/_exp/libslic_w2_$LIB.so
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_READ_sum TCC_READ_SECTORS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TA_BUFFER_READ_LDS_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -- python $R/scripts/bench_conv.py 32 $SH > $O/g$i.log 2>&1 < /dev/null
done
python - "$O" <<'PY' > $R/gpurun_out/r4_pmc_${SH}${LIB}.txt
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(dict))
for f in sorted(glob.glob(O + '/g*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'wino' not in k and 'conv_' not in k: continue
        d = agg[k][r['Counter_Name']]
        d[r['Dispatch_Id']] = d.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = list(agg[k][c].values())
        print(f"   {c:42s} mean {sum(v)/len(v):16.1f}  launches {len(v)}")
PY
find $O -name "*.csv" -delete; rm -rf $O
tail -80 $R/gpurun_out/r4_pmc_${SH}${LIB}.txt
