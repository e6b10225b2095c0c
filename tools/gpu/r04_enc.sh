#!/bin/bash
# kernel trace of the single-lane MiT-B5 forward at B = 4 (tools/dbg/enc_scaling.py one): per-kernel totals of the LAST replay
out=gpurun_out/${1:-r04enc}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ENC_ONLY_B=4 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python3 $GRAFT_REPO_ROOT/tools/dbg/enc_scaling.py > $GRAFT_REPO_ROOT/$out/run.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -v amdgpu.ids $out/run.txt
python - $out <<'PY'
import csv, glob, sys, re, collections
f = glob.glob(sys.argv[1] + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'\(.*', '', n)[:70]
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n))
rows.sort()
# last replay: the last 588-ish kernels; find by the patch-embed conv marker (nchw_to_nhwc_pad) occurrences
marks = [i for i, r in enumerate(rows) if r[2].startswith('nchw_to_nhwc_pad')]
lo = marks[-1]
it = rows[lo:]
span = (it[-1][1] - it[0][0]) / 1e3
busy = sum(e - s for s, e, n in it) / 1e3
print(f'last replay: {len(it)} kernels, span {span:.1f} us, kernel time {busy:.1f} us, gaps {span - busy:.1f} us')
fam = collections.Counter(); cnt = collections.Counter()
for s, e, n in it:
    fam[n] += e - s; cnt[n] += 1
for k, v in fam.most_common(40):
    print(f'  {v / 1e3:8.1f} us  {cnt[k]:4d} x {v / cnt[k] / 1e3:7.2f} us  {k}')
PY
rm -rf $out/trace
