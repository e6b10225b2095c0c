#!/usr/bin/env python3
"""Which GEMM kernel instances over-fetch?  Folds the FETCH_SIZE / WRITE_SIZE passes of one bench.py command per kernel INSTANCE
(template instantiation) and sets the measured bytes against the algorithmic bytes of the launches that ran on it: the bench's own
launch-ordered GEMM record of its last (eager) step (CMDA_BENCH_GEMM_LOG) is matched, dispatch by dispatch, with the 'gemm' dispatches
of the last step of each counter pass (the passes run the same deterministic launch sequence).
Corrections as in pmc_traffic.py: KiB units, FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section).
usage: pmc_gemm_instances.py <fetch_dir> <write_dir> <gemm_log.json> <out.txt>"""
import collections
import csv
import glob
import json
import re
import sys


def last_step_gemms(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    disp = {}
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = (r.get('Process_Id', ''), int(r['Dispatch_Id']))
        e = disp.setdefault(k, [r['Kernel_Name'], 0.0])
        e[1] += float(r['Counter_Value'])
    order = [disp[k] for k in sorted(disp)]
    marks = [i for i, e in enumerate(order) if 'ema_kernel' in e[0]]
    order = order[marks[-1]:] if marks else order
    return [(re.sub(r'\(.*', '', n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', ''))[:64], v)
            for n, v in order if 'gemm' in n]


def main():
    fetch_dir, write_dir, log, out = sys.argv[1:5]
    fe, wr = last_step_gemms(fetch_dir, 'FETCH_SIZE'), last_step_gemms(write_dir, 'WRITE_SIZE')
    rec = json.load(open(log))
    lines = ['rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 bench.py --steps 1 --warmup 1 --no-graph '
             '--no-cpu-baseline --no-parity-mode`, last step; per GEMM kernel instance: measured HBM-side bytes (FETCH x 2 + WRITE) against the',
             'algorithmic bytes (operand tensors once + output) of the launches that ran on it (bench.py CMDA_BENCH_GEMM_LOG, matched in launch order)', '']
    # a GROUPED record of the bench (one cmda_gemm_grouped flush = the deferred weight gradients of a backward phase) is several
    # dispatches: the grouped kernels of each tile kind + the problems the planner launches by themselves (weight-gradient form:
    # both operands K-strided).  Walk both sequences: a plain record takes one dispatch, a grouped record takes the following run of
    # grouped / weight-gradient-form dispatches; the match counts only if both sequences end together.
    def wgrad_form(n):
        return 'grouped' in n or 'gemm_wg_kernel' in n or re.search(r'gemm_glds_kernel<\d+, \d+, true, true', n) is not None
    same = len(fe) == len(wr) and all(a[0] == b[0] for a, b in zip(fe, wr))
    owner, i = [None] * len(fe), 0      # dispatch -> (record index, share of the record's bytes)
    ok = same
    for ri, r in enumerate(rec):
        if not ok or i >= len(fe):
            ok = False
            break
        if 'grouped' in [str(x) for x in r['key']]:
            j = i
            while j < len(fe) and wgrad_form(fe[j][0]):
                j += 1
            if j == i:
                ok = False
                break
            tot = sum(fe[k][1] for k in range(i, j)) or 1.0
            for k in range(i, j):   # the flush's algorithmic bytes split over its dispatches by their measured fetch share
                owner[k] = (ri, fe[k][1] / tot)
            i = j
        else:
            owner[i] = (ri, 1.0)
            i += 1
    matched = ok and i == len(fe)
    lines.append(f'dispatches: fetch pass {len(fe)}, write pass {len(wr)}, bench records {len(rec)} ({sum(1 for r in rec if "grouped" in [str(x) for x in r["key"]])} '
                 f'grouped flushes) -> {"matched" if matched else "NOT matched: instance totals only"}')
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, collections.Counter()])
    for i, (n, v) in enumerate(fe):
        a = acc[n]
        a[0] += 1
        a[1] += v * 1024 * 2
        if matched:
            ri, share = owner[i]
            a[3] += rec[ri]['bytes'] * share
            k = rec[ri]['key']
            a[4][' '.join(str(int(x)) if not isinstance(x, str) else x for x in k[:4])] += 1
    for n, v in wr:
        acc[n][2] += v * 1024
    tot_m = sum(a[1] + a[2] for a in acc.values())
    tot_a = sum(a[3] for a in acc.values())
    lines.append(f'family: measured {tot_m / 1e6:.1f} MB per step, algorithmic {tot_a / 1e6:.1f} MB per step' + (f', ratio {tot_m / tot_a:.2f}' if tot_a else ''))
    lines.append('')
    lines.append('%-66s %7s %12s %12s %12s %7s %12s   %s' % ('instance', 'calls', 'fetch MB', 'write MB', 'algorithmic', 'ratio', 'excess MB', 'most frequent M N K batch'))
    for n, a in sorted(acc.items(), key=lambda kv: -(kv[1][1] + kv[1][2] - kv[1][3])):
        meas = a[1] + a[2]
        shapes = '; '.join(f'{k} x{c}' for k, c in a[4].most_common(2))
        lines.append('%-66s %7d %12.1f %12.1f %12.1f %7s %12.1f   %s' % (n, a[0], a[1] / 1e6, a[2] / 1e6, a[3] / 1e6,
                                                                         '%.2f' % (meas / a[3]) if a[3] else '-', (meas - a[3]) / 1e6, shapes))
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:40]))


if __name__ == '__main__':
    main()
