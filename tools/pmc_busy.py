#!/usr/bin/env python3
"""Fold one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES` pass over a bench.py command into MFMA-busy per kernel
family and per GEMM kernel instance.
  busy/CU   = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)  -- of the CUs the launch occupied, while they were occupied
  busy/chip = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch duration x 2.4 GHz)  -- of the whole device for the launch's duration
(SQ_VALU_MFMA_BUSY_CYCLES counts pipe cycles: 16 per 16x16x32 bf16 instruction, MI355X_MICROARCH.md counter table.)  Durations are the
counter pass's own (serialised dispatches).
usage: pmc_busy.py <pmc_dir> <out.txt> [iterations in the pass; 0 = fold the last step only]"""
import collections
import csv
import glob
import re
import sys

from pmc_traffic import family

CLK_GHZ, SIMDS = 2.4, 1024


def main():
    d, out = sys.argv[1:3]
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        k = (r.get('Process_Id', ''), r['Dispatch_Id'])
        e = disp[k]
        e['name'] = r['Kernel_Name']
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        if 'Start_Timestamp' in r and r['Start_Timestamp']:
            e['dur'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    fam = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    inst = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    order = [disp[k] for k in sorted(disp, key=lambda k: (k[0], int(k[1])))]
    if iters == 0:   # the LAST step only: from the last EMA launch (the step's first kernel) to the end of the pass
        marks = [i for i, e in enumerate(order) if 'ema_kernel' in e['name']]
        order, iters = order[marks[-1]:], 1
    for e in order:
        n = e['name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
        short = re.sub(r'\(.*', '', n)[:70]
        for acc, key in ((fam, family(n)), (inst, short)):
            a = acc[key]
            a[0] += 1
            a[1] += e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
            a[2] += e.get('SQ_BUSY_CU_CYCLES', 0.0)
            a[3] += e.get('dur', 0)
    lines = ['rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES (own pass, program directly behind `--`) over',
             '`python3 bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-parity-mode`; the last step of the pass',
             'busy/CU = MFMA_BUSY / (4 x BUSY_CU_CYCLES); busy/chip = MFMA_BUSY / (1024 SIMDs x duration x 2.4 GHz)', '']

    def table(title, acc, top):
        lines.append(title)
        lines.append('%-72s %8s %12s %10s %10s' % ('', 'launches', 'ms / step', 'busy/CU', 'busy/chip'))
        for k, a in sorted(acc.items(), key=lambda kv: -kv[1][3])[:top]:
            if a[1] == 0 and acc is inst:
                continue
            cu = a[1] / (4 * a[2]) if a[2] else 0.0
            chip = a[1] / (SIMDS * a[3] * CLK_GHZ) if a[3] else 0.0
            lines.append('%-72s %8d %12.3f %10.3f %10.3f' % (k, a[0] // iters, a[3] / 1e6 / iters, cu, chip))
        lines.append('')

    table('per kernel family', fam, 20)
    table('per kernel instance with MFMA work (sorted by time)', inst, 40)
    tot = [sum(a[i] for a in fam.values()) for i in range(4)]
    lines.append('whole step: %d launches, %.2f ms of kernel time, MFMA busy/CU %.3f, busy/chip %.3f'
                 % (tot[0] // iters, tot[3] / 1e6 / iters, tot[1] / (4 * tot[2]) if tot[2] else 0.0,
                    tot[1] / (SIMDS * tot[3] * CLK_GHZ) if tot[3] else 0.0))
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
