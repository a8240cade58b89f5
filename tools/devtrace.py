#!/usr/bin/env python3
"""Device timeline of one training step from the engine's own in-kernel stamps (WN_DEVTRACE=<file>, csrc/wn_common.h): every
tile-engine and grouped weight-gradient launch with {first workgroup's start, last workgroup's end} on the 100 MHz wall clock.  No
profiler attached, so the host's enqueue lead is the product's.
   python tools/devtrace.py <file> [--all]"""
import sys

EPI = {0: 'gate', 1: 'store(out/skip/head)', 2: 'f32(yhat/dc)', 3: 'dgate', 4: 'relumask', 5: 'dx', 101: 'wgrad<1>', 102: 'wgrad<2>', 103: 'wgrad<3>'}


def main():
    rows = []
    for line in open(sys.argv[1]):
        if line.startswith('#'):
            continue
        i, epi, st, n, s, e = line.split()
        s, e = int(s), int(e)
        if s == 2 ** 64 - 1 or e < s:
            continue
        rows.append((s, e, int(epi), st, int(n)))
    streams = {st: k for k, st in enumerate(sorted({r[3] for r in rows}, key=lambda st: min(r[0] for r in rows if r[3] == st)))}
    rows.sort()
    t0 = rows[0][0]
    us = lambda t: (t - t0) / 100.0
    print('%d stamped launches, %.1f us from the first start to the last end' % (len(rows), us(max(r[1] for r in rows))))
    for st, k in streams.items():
        mine = [r for r in rows if r[3] == st]
        print('stream %d (%s): %3d launches, busy %8.1f us, span %8.1f .. %8.1f us' % (k, st, len(mine), sum(r[1] - r[0] for r in mine) / 100.0, us(mine[0][0]), us(max(r[1] for r in mine))))
    # phases per stream: first / last launch of each kind
    for st, k in streams.items():
        for epi in sorted({r[2] for r in rows if r[3] == st}):
            mine = [r for r in rows if r[3] == st and r[2] == epi]
            d = [(r[1] - r[0]) / 100.0 for r in mine]
            print('   stream %d %-22s x%3d  first start %8.1f  last end %8.1f  avg %6.1f us  min %6.1f  max %6.1f' % (k, EPI.get(epi, epi), len(mine), us(mine[0][0]), us(max(r[1] for r in mine)), sum(d) / len(d), min(d), max(d)))
    # concurrency histogram over the stamped launches
    pts = sorted([(r[0], 1) for r in rows] + [(r[1], -1) for r in rows])
    depth, last, hist = 0, pts[0][0], {}
    for t, dd in pts:
        hist[depth] = hist.get(depth, 0) + t - last; last = t; depth += dd
    print('stamped launches in flight: ' + '  '.join('%d: %.0f us' % (k, v / 100.0) for k, v in sorted(hist.items())))
    if '--all' in sys.argv:
        for s, e, epi, st, n in rows:
            print('%9.1f %9.1f %7.1f  s%d  %-22s rows/groups %d' % (us(s), us(e), (e - s) / 100.0, streams[st], EPI.get(epi, epi), n))


if __name__ == '__main__':
    main()
