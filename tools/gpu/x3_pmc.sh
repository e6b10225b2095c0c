#!/bin/bash
# PMC passes (one per counter pair) over ONE lean split-bf16 GEMM shape: where a k-tile's time goes
out=gpurun_out/x3pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S=${SHAPE:-"4096 320 320"}
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "SQ_INST_LEVEL_LDS SQ_WAVES"; do
  d=$out/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 tools/dbg/x3_gemm_one.py $S > $out/log 2>&1
  python - "$d" "$c" <<'PY'
import csv,glob,sys,collections
fs=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
if not fs: print(sys.argv[2],'-> no counters'); sys.exit()
agg=collections.defaultdict(list); dur=[]
for r in csv.DictReader(open(fs[0])):
    if 'gemm_x3_lean' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
print({k:round(sum(v)/len(v)) for k,v in agg.items()})
PY
  rm -rf $d
done
