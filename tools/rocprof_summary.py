#!/usr/bin/env python3
"""Turns rocprofv3's rocpd SQLite output (ROCm 7.2 default format) into the text summaries
kept under profiles/: per-kernel time (== `--stats`), and per-kernel PMC averages."""
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    i = name.find("(")
    return name[:i] if i > 0 else name


def stats(path):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    out = ["%-46s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for n, c, t, a, p in rows:
        out.append("%-46s %8d %14.1f %12.3f %6.2f%%" % (short(n)[:46], c, t, a, p))
    return "\n".join(out)


def pmc(path):
    db = sqlite3.connect(path)
    q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), "
         "avg(duration) from counters_collection group by kernel_name, counter_name "
         "order by avg(value)*count(*) desc")
    out = ["%-46s %-12s %7s %14s %12s %12s %10s" % ("kernel", "counter", "calls", "avg", "min",
                                                     "max", "avg_ns")]
    for n, cn, c, a, mn, mx, d in db.execute(q):
        out.append("%-46s %-12s %7d %14.3f %12.3f %12.3f %10.0f" % (short(n)[:46], cn, c, a, mn,
                                                                     mx, d))
    return "\n".join(out)


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    print(stats(path) if mode == "stats" else pmc(path))
