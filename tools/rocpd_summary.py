#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.x) rocpd SQLite database as text.

    python tools/rocpd_summary.py stats <results.db>      # --kernel-trace --stats run: per-kernel time table
    python tools/rocpd_summary.py pmc   <results.db>      # --pmc run: per-kernel mean counter value per dispatch
    python tools/rocpd_summary.py detail <results.db> [n] # --kernel-trace run: time per (kernel, grid, workgroup size)

rocprofv3 on this image writes <pid>_results.db instead of CSV files; the tracked summaries under
profiles/ are produced with this script from the databases collected on the GPU box."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)
    return name.replace('void ', '')[:70]


def stats(db):
    con = sqlite3.connect(db)
    rows = con.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                       'from kernels group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows)
    print('%-72s %7s %12s %11s %11s %11s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
    for n, c, s, a, mn, mx in rows:
        print('%-72s %7d %12.1f %11.2f %11.2f %11.2f %6.2f' % (short(n), c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / total))
    print('total kernel time %.3f ms over %d dispatches' % (total / 1e6, sum(r[1] for r in rows)))


def detail(db):
    """per (kernel, grid, workgroup) rows: which launches of a many-call kernel carry its time"""
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute('pragma table_info(kernels)')]
    if 'grid_x' not in cols:
        print('columns:', cols)
        return
    g = 'grid_x, grid_y, grid_z, workgroup_x' if 'workgroup_x' in cols else 'grid_x, grid_y, grid_z, 0'
    rows = con.execute('select name, %s, count(*), sum(duration), avg(duration), max(duration) from kernels '
                       'group by name, %s order by sum(duration) desc' % (g, g)).fetchall()
    total = sum(r[6] for r in rows)
    print('%-56s %22s %5s %6s %11s %10s %10s %6s' % ('kernel', 'grid (work-items)', 'wg_x', 'calls', 'total_us', 'avg_us', 'max_us', '%'))
    for n, x, y, z, w, c, s_, a, mx in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 80]:
        print('%-56s %22s %5d %6d %11.1f %10.2f %10.2f %6.2f' % (short(n)[:56], '%dx%dx%d' % (x, y, z), w, c, s_ / 1e3, a / 1e3,
                                                                 mx / 1e3, 100.0 * s_ / total))


def pmc(db):
    con = sqlite3.connect(db)
    rows = con.execute('select kernel_name, counter_name, count(*), avg(value), sum(value), avg(duration) '
                       'from counters_collection group by kernel_name, counter_name order by sum(value) desc').fetchall()
    print('%-72s %-14s %7s %16s %16s %11s' % ('kernel', 'counter', 'calls', 'mean/dispatch', 'sum', 'avg_us'))
    for n, cn, c, a, s, d in rows:
        print('%-72s %-14s %7d %16.1f %16.1f %11.2f' % (short(n), cn, c, a, s, d / 1e3))


if __name__ == '__main__':
    {'stats': stats, 'pmc': pmc, 'detail': detail}[sys.argv[1]](sys.argv[2])
