#!/bin/bash
out=gpurun_out/sgpmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/p1 -- python3 tools/dbg/small_gemm_trace.py > $out/log1 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_REQ_sum --kernel-trace --output-format csv -d $out/p2 -- python3 tools/dbg/small_gemm_trace.py > $out/log2 2>&1
python - <<'PY'
import csv,glob,collections
for d in ('p1','p2'):
    fs=glob.glob(f'gpurun_out/sgpmc/{d}/**/*counter_collection.csv',recursive=True)
    if not fs: print(d,'no counters'); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if 'gemm' not in r['Kernel_Name']: continue
        key=(r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X'), r['Kernel_Name'][40:90])
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(d,k,{c:round(sum(x)/len(x)) for c,x in v.items()}, 'n',len(next(iter(v.values()))))
PY
rm -rf $out/p1 $out/p2
