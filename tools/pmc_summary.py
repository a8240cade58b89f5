#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters from a *_counter_collection.csv.
   python tools/pmc_summary.py <counter_collection.csv> [--md]"""
import csv
import re
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(path, newline='') as f:
        rd = csv.DictReader(f)
        for row in rd:
            name = re.sub(r'\(.*', '', row.get('Kernel_Name', row.get('Kernel Name', '?')))[:110]
            cn = row.get('Counter_Name'); cv = row.get('Counter_Value')
            if cn is None or cv is None:
                continue
            a = acc[name][cn]
            a[0] += float(cv); a[1] += 1
    md = '--md' in sys.argv
    if md:
        print('| kernel | counter | dispatches | mean per dispatch | total |\n|---|---|---|---|---|')
    for name in sorted(acc, key=lambda n: -max(v[0] for v in acc[n].values())):
        for cn, (s, n) in sorted(acc[name].items()):
            if md:
                print('| `%s` | %s | %d | %.4g | %.4g |' % (name, cn, n, s / max(n, 1), s))
            else:
                print('%-112s %-14s %6d %14.4g %14.4g' % (name, cn, n, s / max(n, 1), s))


if __name__ == '__main__':
    main()
