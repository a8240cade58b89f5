#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: phases, per-queue busy time, idle gaps.
   python tools/timeline.py <kernel_trace.csv> [step_index_from_end=1]"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    m = re.match(r'void wn_gemm_lds_kernel<([^>]*)>', n)
    if m:
        a = [x.strip() for x in m.group(1).split(',')]
        epi = {'0': 'gate', '1': 'store(out/skip/head)', '2': 'f32(yhat/dc)', '3': 'dgate', '4': 'relumask', '5': 'dx'}.get(a[6], a[6])
        return 'gemm<%sx%s,%s>' % (a[0], a[1], epi)
    m = re.match(r'void (wn_wgrad_lds_kernel<[^>]*>)', n)
    if m:
        return m.group(1)
    return re.sub(r'\(.*', '', n)[:40]


def main():
    rows = []
    with open(sys.argv[1], newline='') as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']))
    rows.sort()
    packs = [i for i, r in enumerate(rows) if r[2].startswith('wn_pack_kernel')]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    i0, i1 = packs[-k - 1], packs[-k]
    step = rows[i0:i1]
    t0 = step[0][0]
    T = (rows[i1][0] - t0) / 1e3
    print('step of %d kernels, %.1f us between consecutive wn_pack_kernel starts' % (len(step), T))
    # phases
    def first(pred): return next((r for r in step if pred(r[2])), None)
    def last(pred): return next((r for r in reversed(step) if pred(r[2])), None)
    marks = [('pack start', step[0][0]),
             ('first gate', first(lambda n: ', 0, 1, ' in n and 'wn_gemm_lds' in n)[0]),
             ('loss done', last(lambda n: 'wn_mol_loss' in n or 'wn_gauss_loss' in n or 'wn_ce_loss' in n)[1]),
             ('last dx done', max(r[1] for r in step if 'wn_gemm_lds' in r[2] and ', 5, 1' in r[2])),
             ('last wgrad done', max(r[1] for r in step if 'wgrad' in r[2])),
             ('adam done', last(lambda n: 'wn_adam' in n)[1])]
    prev = t0
    for name, t in marks:
        print('  %-16s at %9.1f us  (+%8.1f)' % (name, (t - t0) / 1e3, (t - prev) / 1e3))
        prev = t
    # union busy + gaps
    ev = sorted((s, e) for s, e, _, _ in step)
    busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
    for s, e in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e)); cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print('GPU busy (union of kernel intervals) %.1f us = %.1f %% of the step; %d idle gaps, sum %.1f us' % (busy / 1e3, 100 * busy / 1e3 / T, len(gaps), sum(g for g, _ in gaps) / 1e3))
    for g, at in sorted(gaps, reverse=True)[:8]:
        before = max((r for r in step if r[1] <= at + 1), key=lambda r: r[1])
        print('   gap %7.1f us at %9.1f us after %s' % (g / 1e3, (at - t0) / 1e3, short(before[2])))
    # per queue
    q = defaultdict(list)
    for s, e, n, qi in step:
        q[qi].append((s, e, n))
    for qi, lst in sorted(q.items()):
        tot = sum(e - s for s, e, _ in lst)
        print('queue %s: %4d kernels, busy %9.1f us, span %9.1f .. %9.1f us' % (qi, len(lst), tot / 1e3, (lst[0][0] - t0) / 1e3, (max(e for _, e, _ in lst) - t0) / 1e3))
    # concurrency histogram: time with n kernels in flight
    pts = sorted([(s, 1) for s, e, _, _ in step] + [(e, -1) for s, e, _, _ in step])
    depth, lastt, hist = 0, pts[0][0], defaultdict(int)
    for t, d in pts:
        hist[depth] += t - lastt; lastt = t; depth += d
    print('kernels in flight: ' + '  '.join('%d: %.0f us' % (d, v / 1e3) for d, v in sorted(hist.items())))
    # excerpt: every kernel of a window of the forward chain and of the backward chain (start, end, queue): how the two part streams interleave
    if '--detail' in sys.argv:
        for label, w0 in (('forward', marks[1][1] + 800000), ('backward', marks[2][1] + 800000)):
            print('--- %s chain, 600 us window from %.1f us: start end dur queue kernel' % (label, (w0 - t0) / 1e3))
            for s, e, n, qi in step:
                if e > w0 and s < w0 + 600000:
                    print('   %9.1f %9.1f %7.1f  q%s  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, qi, short(n)))
    # per kernel kind
    agg = defaultdict(lambda: [0, 0])
    for s, e, n, _ in step:
        a = agg[short(n)]; a[0] += 1; a[1] += e - s
    for n, (c, tt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print('   %-34s x%3d  %8.1f us  (avg %6.1f)' % (n, c, tt / 1e3, tt / 1e3 / c))


if __name__ == '__main__':
    main()
