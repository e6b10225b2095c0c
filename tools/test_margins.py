"""Fold the margin logs of several test runs (CMDA_TEST_MARGINS=<file>, tests/conftest.py) into one table: per (test, check) the
worst value seen over all runs / boxes, its bound and the headroom = bound / worst ('le' checks) or (1 - bound) / (1 - worst)
('ge' checks on agreement fractions).  Flags every bound with less than --min headroom (default 2) -- VERDICT r03 #1b: no bound
within 2x of an observed value.  Exact checks (bound 0, value 0) and structural ones are skipped.

    python tools/test_margins.py gpurun_out/*/margins*.jsonl [--min 2] [--all]
"""
import argparse
import json
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('files', nargs='+')
    ap.add_argument('--min', type=float, default=2.0)
    ap.add_argument('--all', action='store_true', help='print every check, not only the tight ones')
    args = ap.parse_args()
    worst = {}
    runs = {}
    for fi, f in enumerate(args.files):
        for line in open(f):
            r = json.loads(line)
            key = (r['test'], r['name'], r['kind'])
            v, b = r['value'], r['bound']
            w = worst.get(key)
            if w is None or (v > w[0] if r['kind'] == 'le' else v < w[0]):
                worst[key] = (v, b)
            runs.setdefault(key, set()).add(fi)
    rows = []
    for (test, name, kind), (v, b) in worst.items():
        if kind == 'le':
            if b == 0 and v == 0:
                continue
            head = float('inf') if v == 0 else b / v
        else:
            head = float('inf') if v >= 1.0 else ((1.0 - b) / (1.0 - v) if b < 1.0 else 0.0)
        rows.append((head, test, name, kind, v, b, len(runs[(test, name, kind)])))
    rows.sort()
    tight = [r for r in rows if r[0] < args.min]
    print(f'{len(rows)} bounded checks over {len(args.files)} logs; {len(tight)} with headroom < {args.min}x')
    for head, test, name, kind, v, b, n in (rows if args.all else tight):
        print(f'{head:8.2f}x  {kind}  worst {v:.4g}  bound {b:.4g}  ({n} logs)  {test} :: {name}')
    return 1 if tight else 0


if __name__ == '__main__':
    sys.exit(main())
