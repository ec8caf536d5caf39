#!/bin/bash
# GPU box: PMC passes (each its own run, --kernel-trace only) over the layer1 gather-GEMM alone (scripts/dbg_l1.py)
R="$(pwd)"; S="$R/gpurun_out/pmc_l1"; rm -rf "$S"; mkdir -p "$S"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$S/counters.txt" 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$S/p$i" -- python "$R/scripts/dbg_l1.py" > "$S/p$i.log" 2>&1 < /dev/null
done
find "$S" -name "*agent_info.csv" -delete; find "$S" -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$S/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv_gemm_dma_kernel<128" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        # rows come per dispatch (possibly per dimension instance): sum per dispatch = total / number of dispatches
        print(f"{k:36s} total/launch {sum(v)/6:16.0f}   ({len(v)} rows)")
PY
