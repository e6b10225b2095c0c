#!/bin/bash
out=gpurun_out/bigpmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM SQ_ACCUM_PREV_HIRES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  d=$out/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/dbg/big_gemm_one.py 8192 8192 8192 ${HINT:-516} > $out/log 2>&1
  python - "$d" "$c" <<'PY'
import csv,glob,sys,collections
fs=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
if not fs: print(sys.argv[2],'-> no counters'); sys.exit()
agg=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'gemm_glds' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print({k:round(sum(v)/len(v)) for k,v in agg.items()})
PY
  rm -rf $d
done
