#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel stats table
(name, calls, total ms, avg us, % of GPU kernel time).   python tools/rocpd_stats.py results.db [--md]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(rocpd_kernel_dispatch)')]
    sym_cols = [r[1] for r in cur.execute('pragma table_info(rocpd_info_kernel_symbol)')]
    name_col = 'display_name' if 'display_name' in sym_cols else ('kernel_name' if 'kernel_name' in sym_cols else 'name')
    q = ('select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) '
         'from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.%s order by 3 desc' % (name_col, name_col))
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    md = '--md' in sys.argv
    if md:
        print('| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|')
    for name, n, t, mn, mx in rows:
        short = re.sub(r'\(.*', '', name)[:110]
        if md:
            print('| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |' % (short, n, t / 1e6, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
        else:
            print('%-112s %6d %10.3f ms %9.1f us  %5.1f%%' % (short, n, t / 1e6, t / n / 1e3, 100.0 * t / tot))
    print(('\ntotal kernel time %.3f ms over %d dispatches' % (tot / 1e6, sum(r[1] for r in rows))))


if __name__ == '__main__':
    main()
