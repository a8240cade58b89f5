#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters from a *_counter_collection.csv.
   python tools/pmc_summary.py <counter_collection.csv> [--md]

HBM-side traffic summary that bench.py loads (no hand-copied constants in bench.py):
   python tools/pmc_summary.py --traffic <fetch.csv|fetch.md> <write.csv|write.md> --workload c2 --batch 8 --time 11000 \
          --tag r4 --out profiles/traffic.json
reads the FETCH_SIZE and WRITE_SIZE passes (rocprofv3 cannot collect both in one pass; the raw CSV or the committed --md table),
applies the gfx950 calibration of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts half of a wide streaming read: x 2; both
counters are in KiB-like units of 1024 B) and writes, per kernel and per training step, bytes = 2 x FETCH_SIZE + WRITE_SIZE.
The number of steps in the passes is the dispatch count of wn_adam_kernel (one per step)."""
import csv
import json
import re
import sys
from collections import defaultdict

UNIT = 1024.0        # FETCH_SIZE / WRITE_SIZE are reported in kilobytes


def read_counters(path):
    """{kernel: {counter: [sum, dispatches]}} from a rocprofv3 counter_collection.csv or from this tool's own --md table."""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    if path.endswith('.md'):
        for line in open(path):
            m = re.match(r'\|\s*`(.*)`\s*\|\s*(\w+)\s*\|\s*(\d+)\s*\|\s*([0-9.eE+-]+)\s*\|\s*([0-9.eE+-]+)\s*\|', line)
            if m:
                acc[m.group(1)][m.group(2)] = [float(m.group(5)), int(m.group(3))]
        return acc
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            name = re.sub(r'\(.*', '', row.get('Kernel_Name', row.get('Kernel Name', '?')))[:110]
            cn = row.get('Counter_Name'); cv = row.get('Counter_Value')
            if cn is None or cv is None:
                continue
            a = acc[name][cn]
            a[0] += float(cv); a[1] += 1
    return acc


def traffic_summary(fetch_path, write_path, gate_kernel='wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, 0, 1, 3>'):
    fe, wr = read_counters(fetch_path), read_counters(write_path)

    def steps_of(acc, counter):
        for k, v in acc.items():
            if k.startswith('wn_adam_kernel') and counter in v:
                return v[counter][1]
        raise SystemExit('no wn_adam_kernel dispatches in the %s pass' % counter)
    sf, sw = steps_of(fe, 'FETCH_SIZE'), steps_of(wr, 'WRITE_SIZE')
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        f = fe.get(k, {}).get('FETCH_SIZE', [0.0, 0]); w = wr.get(k, {}).get('WRITE_SIZE', [0.0, 0])
        kernels[k] = {'dispatches_per_step': (f[1] / sf) if f[1] else (w[1] / sw),
                      'fetch_x2_bytes_per_step': 2.0 * f[0] * UNIT / sf, 'write_bytes_per_step': w[0] * UNIT / sw}
        kernels[k]['bytes_per_step'] = kernels[k]['fetch_x2_bytes_per_step'] + kernels[k]['write_bytes_per_step']
    out = {'steps_in_fetch_pass': sf, 'steps_in_write_pass': sw,
           'fetch_x2_bytes_per_step': sum(v['fetch_x2_bytes_per_step'] for v in kernels.values()),
           'write_bytes_per_step': sum(v['write_bytes_per_step'] for v in kernels.values()),
           'formula': 'bytes = 2 x FETCH_SIZE + WRITE_SIZE (x 1024), separate rocprofv3 --pmc passes, gfx950 FETCH_SIZE calibration (wide streaming reads counted at half)',
           'kernels': kernels}
    out['bytes_per_step'] = out['fetch_x2_bytes_per_step'] + out['write_bytes_per_step']
    gk = [k for k in kernels if gate_kernel in k]
    if gk:
        g = kernels[gk[0]]
        out['gate_kernel'] = gk[0]
        out['gate_bytes_per_launch'] = g['bytes_per_step'] / max(g['dispatches_per_step'], 1e-9)
        out['gate_launches_per_step'] = g['dispatches_per_step']
    return out


def main():
    if '--traffic' in sys.argv:
        i = sys.argv.index('--traffic')
        fetch, write = sys.argv[i + 1], sys.argv[i + 2]

        def opt(name, default=None):
            return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default
        s = traffic_summary(fetch, write)
        s.update(workload=opt('--workload', 'c2'), batch=int(opt('--batch', 8)), time=int(opt('--time', 11000)), tag=opt('--tag', ''),
                 sources=[opt('--fetch-name', fetch), opt('--write-name', write)])
        text = json.dumps(s, indent=1, sort_keys=True)
        out = opt('--out')
        if out:
            open(out, 'w').write(text + '\n')
        print('%.3f GB/step = %.3f fetched (x2) + %.3f written; gate launch %.1f MB' % (s['bytes_per_step'] / 1e9, s['fetch_x2_bytes_per_step'] / 1e9,
              s['write_bytes_per_step'] / 1e9, s.get('gate_bytes_per_launch', 0) / 1e6))
        return
    path = sys.argv[1]
    acc = read_counters(path)
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
