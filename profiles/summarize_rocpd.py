#!/usr/bin/env python3
"""Turns rocprofv3's rocpd SQLite output (gpurun_out/<run>/*_results.db) into the text summaries kept here.

usage: summarize_rocpd.py stats <db>            per-kernel calls / total / avg / min / max (what --stats reports)
       summarize_rocpd.py pmc <db> [<db> ...]   per-kernel average of every collected counter
"""
import sqlite3
import sys


def stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print("%-110s %7s %14s %12s %12s %12s %6s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "%"))
    for name, calls, tot, avg, mn, mx in rows:
        print("%-110s %7d %14d %12.0f %12d %12d %6.2f" % (name[:110], calls, tot, avg, mn, mx, 100.0 * tot / total))


def pmc(paths):
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                           "group by kernel_name, counter_name order by avg(value) * count(*) desc").fetchall()
        print("# %s" % path)
        print("%-110s %-12s %7s %16s %12s" % ("kernel", "counter", "calls", "avg_value", "avg_ns"))
        for name, ctr, calls, val, dur in rows:
            print("%-110s %-12s %7d %16.1f %12.0f" % (name[:110], ctr, calls, val, dur))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
