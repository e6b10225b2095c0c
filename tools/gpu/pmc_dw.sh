#!/bin/bash
mkdir -p gpurun_out/pmcdw
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -o "Name:\s*[A-Za-z0-9_]*" | sort -u | grep -i "TCC_HIT\|TCC_MISS\|MemUnit\|TCP_TCC_READ\|TCC_REQ\|FETCH_SIZE\|WRITE_SIZE\|TCC_EA0_RDREQ\|TCC_READ\|L2CacheHit\|MemWrites32B\|TA_BUSY\|TCP_PENDING\|LDSBank\|OccupancyPercent\|TCC_EA0_WRREQ" | tr '\n' ' ' > gpurun_out/pmcdw/avail.txt
cat gpurun_out/pmcdw/avail.txt
for c in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "MemUnitBusy MemUnitStalled" "FETCH_SIZE WRITE_SIZE"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcdw/$n -- python3 tools/hbm_bench.py --batch 8 > /dev/null 2> gpurun_out/pmcdw/err_$n
  python - "$n" <<'PY'
import csv, glob, sys, collections
n = sys.argv[1]
f = glob.glob(f'gpurun_out/pmcdw/{n}/**/*counter_collection.csv', recursive=True)
if not f:
    print(n, 'no output'); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    if 'dw_stencil' in k or 'dw_bwd' in k or 'dw_gelu' in k or 'ln_fwd' in k:
        key = (k.split('(')[0][-50:], r['Grid_Size'])
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, d in sorted(acc.items()):
    print(key, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, len(next(iter(d.values()))))
PY
  rm -rf gpurun_out/pmcdw/$n
done > gpurun_out/pmcdw/summary.txt 2>&1
cat gpurun_out/pmcdw/summary.txt | head -80
