#!/bin/bash
# FETCH_SIZE / duration of conv_wino2_kernel at the layer1 and layer3 shapes: the library in the tree against csrc/_exp/libslic_w2_ktouter.so
cd "$(dirname "$0")/../.."
R=$PWD; O=$R/gpurun_out/r4_fetch_ab; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for lib in new ktouter; do
  if [ $lib = new ]; then unset SLIC_LIB_PATH; else export SLIC_LIB_PATH=$R/video_similarity_search_amd/csrc/_exp/libslic_w2_ktouter.so; fi
  for sh in l1 c7; do
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${lib}_$sh -- python $R/scripts/bench_conv.py 32 $sh > $O/${lib}_$sh.log 2>&1
  done
done
python - <<'PY'
import csv,glob,os
O=os.environ.get('O') or '/root/repo/gpurun_out/r4_fetch_ab'
for d in sorted(glob.glob(O+'/*/')):
    f=glob.glob(d+'**/*counter_collection.csv',recursive=True)
    if not f: continue
    per={}
    for r in csv.DictReader(open(f[0])):
        if 'conv_wino2_kernel' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
            per.setdefault(r['Dispatch_Id'],0.0); per[r['Dispatch_Id']]+=float(r['Counter_Value'])
    v=list(per.values())
    if v: print(os.path.basename(d.rstrip('/')), 'FETCH_SIZE KB mean', sum(v)/len(v), 'launches', len(v))
PY
