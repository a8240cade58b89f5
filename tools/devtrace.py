#!/usr/bin/env python3
"""Device timeline of one training step from the engine's own in-kernel stamps (wn_trace_arm / wn_trace_read, or WN_DEVTRACE=<file>;
csrc/wn_train.hip): every tile-engine and grouped weight-gradient launch with {first workgroup's start, last workgroup's end} on the
100 MHz wall clock.  No profiler attached, so the host's enqueue lead is the product's.
   python tools/devtrace.py <file> [--all]"""
import sys

EPI = {0: 'gate', 1: 'store(out/skip/head)', 2: 'f32(yhat/dc)', 3: 'dgate', 4: 'relumask', 5: 'dx', 101: 'wgrad<1>', 102: 'wgrad<2>', 103: 'wgrad<3>',
       201: 'upsample net fwd*', 202: 'input conv fwd*', 203: 'loss + d y_hat*', 204: 'column sums*', 205: 'upsample net bwd*', 206: 'clip + Adam + EMA*', 207: 'input conv bwd*'}
# kinds >= 200 (*): groups of kernels without in-kernel stamps, bracketed by one-thread stamp kernels on their stream (+- one dispatch)
NEVER = 2 ** 64 - 1


def parse_file(path):
    """-> [(kind, stream, start_tick, end_tick)] in enqueue order."""
    rows = []
    for line in open(path):
        if line.startswith('#'):
            continue
        _, epi, st, _, s, e = line.split()
        rows.append((int(epi), st, int(s), int(e)))
    return rows


def summarise(rows):
    """rows: [(kind, stream, start_tick, end_tick)].  Phases of the step in microseconds from the first stamped start:
    forward = up to the start of the first backward launch (head mask GEMM), backward chain = up to the end of the last d x,
    tail = up to the end of the last weight gradient; plus the time with n stamped launches in flight."""
    rows = [r for r in rows if r[2] != NEVER and r[3] >= r[2]]
    if not rows:
        return None
    t0 = min(r[2] for r in rows)
    us = lambda t: (t - t0) / 100.0
    first_bwd = min((r[2] for r in rows if r[0] in (3, 4, 5)), default=None)
    last_dx = max((r[3] for r in rows if r[0] == 5), default=None)
    last_wg = max((r[3] for r in rows if 100 <= r[0] < 200), default=None)
    last_any = max(r[3] for r in rows)
    pts = sorted([(r[2], 1) for r in rows] + [(r[3], -1) for r in rows])
    depth, last, hist = 0, pts[0][0], {}
    for t, d in pts:
        hist[depth] = hist.get(depth, 0) + t - last; last = t; depth += d
    streams = sorted({r[1] for r in rows}, key=lambda st: min(r[2] for r in rows if r[1] == st))
    per_stream = []
    for st in streams:
        mine = [r for r in rows if r[1] == st]
        per_stream.append({'launches': len(mine), 'busy_us': sum(r[3] - r[2] for r in mine) / 100.0, 'first_start_us': us(min(r[2] for r in mine)), 'last_end_us': us(max(r[3] for r in mine)),
                           'first_backward_start_us': us(min((r[2] for r in mine if r[0] in (3, 4, 5)), default=t0)) if any(r[0] in (3, 4, 5) for r in mine) else None})
    return {'launches': len(rows), 'span_us': us(max(r[3] for r in rows)),
            'forward_us': us(first_bwd) if first_bwd else None,
            'backward_chain_us': (us(last_dx) - us(first_bwd)) if first_bwd and last_dx else None,
            'weight_gradient_tail_us': (us(last_wg) - us(last_dx)) if last_wg and last_dx else None,
            'after_last_weight_gradient_us': (us(last_any) - us(last_wg)) if last_wg else None,      # side-stream groups + the optimiser (kinds >= 200)
            'in_flight_us': {str(k): v / 100.0 for k, v in sorted(hist.items())}, 'streams': per_stream}


def main():
    rows = parse_file(sys.argv[1])
    s = summarise(rows)
    print('%d stamped launches, %.1f us from the first start to the last end' % (s['launches'], s['span_us']))
    print('forward %.1f us | backward chain %.1f us | weight-gradient tail %.1f us' % (s['forward_us'] or 0, s['backward_chain_us'] or 0, s['weight_gradient_tail_us'] or 0))
    for k, st in enumerate(s['streams']):
        print('stream %d: %3d launches, busy %8.1f us, span %8.1f .. %8.1f us, first backward launch at %s' % (k, st['launches'], st['busy_us'], st['first_start_us'], st['last_end_us'],
              '%.1f us' % st['first_backward_start_us'] if st['first_backward_start_us'] is not None else '-'))
    good = [r for r in rows if r[2] != NEVER and r[3] >= r[2]]
    t0 = min(r[2] for r in good)
    us = lambda t: (t - t0) / 100.0
    order = {st: k for k, st in enumerate(sorted({r[1] for r in good}, key=lambda st: min(r[2] for r in good if r[1] == st)))}
    for st, k in order.items():
        for epi in sorted({r[0] for r in good if r[1] == st}):
            mine = [r for r in good if r[1] == st and r[0] == epi]
            d = [(r[3] - r[2]) / 100.0 for r in mine]
            print('   stream %d %-22s x%3d  first start %8.1f  last end %8.1f  avg %6.1f us  min %6.1f  max %6.1f' % (k, EPI.get(epi, epi), len(mine), us(min(r[2] for r in mine)), us(max(r[3] for r in mine)), sum(d) / len(d), min(d), max(d)))
    print('stamped launches in flight: ' + '  '.join('%s: %.0f us' % kv for kv in s['in_flight_us'].items()))
    if '--all' in sys.argv:
        for epi, st, b, e in sorted(good, key=lambda r: r[2]):
            print('%9.1f %9.1f %7.1f  s%d  %s' % (us(b), us(e), (e - b) / 100.0, order[st], EPI.get(epi, epi)))


if __name__ == '__main__':
    main()
