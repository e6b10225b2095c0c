#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of one bench.py command into per-launch HBM-side traffic by
kernel family: profiles/pmc_traffic_<workload>.json (read by bench.py -> roofline.traffic) + a text table.
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of wide coalesced reads -> x2; WRITE_SIZE is exact for 16-byte streaming stores and float atomics.
usage: pmc_traffic.py <fetch_dir> <write_dir> <workload> <out_prefix> [algorithmic_mb_per_gemm_launch]"""
import collections
import csv
import glob
import json
import sys


def family(n):
    if 'gemm' in n:
        return 'gemm'
    if 'attn' in n:
        return 'attention'
    if 'ln_' in n:
        return 'layernorm'
    if 'dw_' in n:
        return 'depthwise'
    if 'bn_' in n:
        return 'batchnorm'
    if 'ce_' in n or 'pseudo' in n:
        return 'ce/pseudo-label'
    return 'other'


def load(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = family(r['Kernel_Name'])
        acc[k][0] += 1
        acc[k][1] += float(r['Counter_Value'])
    return acc


def main():
    fetch_dir, write_dir, workload, out = sys.argv[1:5]
    alg = float(sys.argv[5]) if len(sys.argv) > 5 else None
    fe, wr = load(fetch_dir, 'FETCH_SIZE'), load(write_dir, 'WRITE_SIZE')
    rows = {}
    for k in sorted(set(fe) | set(wr)):
        n = max(fe[k][0], wr[k][0], 1)
        rows[k] = dict(launches=n, fetch_mb_per_launch=round(fe[k][1] * 1024 * 2 / n / 1e6, 3),
                       write_mb_per_launch=round(wr[k][1] * 1024 / n / 1e6, 3))
    g = rows.get('gemm', dict(launches=0, fetch_mb_per_launch=0.0, write_mb_per_launch=0.0))
    js = dict(workload=workload, hbm_mb_per_launch=round(g['fetch_mb_per_launch'] + g['write_mb_per_launch'], 3),
              fetch_mb_per_launch=g['fetch_mb_per_launch'], write_mb_per_launch=g['write_mb_per_launch'], launches=g['launches'],
              algorithmic_mb_per_launch=alg, families=rows,
              source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 bench.py --steps 1 --warmup 1 '
                     '--no-cpu-baseline --no-graph`; FETCH_SIZE x2 (gfx950), KiB units (MI355X_MICROARCH.md)')
    with open(out + '.json', 'w') as f:
        json.dump(js, f, indent=1)
    with open(out + '.txt', 'w') as f:
        f.write(js['source'] + '\n\n%-18s %9s %16s %16s\n' % ('family', 'launches', 'fetch MB/launch', 'write MB/launch'))
        for k, r in rows.items():
            f.write('%-18s %9d %16.3f %16.3f\n' % (k, r['launches'], r['fetch_mb_per_launch'], r['write_mb_per_launch']))
        if alg:
            f.write(f'\nGEMM family: algorithmic bytes per launch {alg} MB vs measured {js["hbm_mb_per_launch"]} MB\n')
    print(open(out + '.txt').read())


if __name__ == '__main__':
    main()
