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


def plain(name):
    """void ydc::k_match_pass<1>(...) -> k_match_pass (the name bench.py uses)."""
    n = short(name)
    n = n.split("<")[0]
    return n.split("::")[-1].strip()


def hbmjson(fetch_db, write_db):
    """HBM bytes per launch per kernel from the two PMC passes. FETCH_SIZE / WRITE_SIZE are
    reported in KB; gfx950: FETCH_SIZE reports exactly half the bytes of wide coalesced
    streaming reads (MI355X_MICROARCH.md, HBM section), so it is doubled; WRITE_SIZE is taken
    as reported (uncalibrated)."""
    import json
    out = {}
    q = ("select kernel_name, count(*), avg(value), avg(duration) from counters_collection "
         "where counter_name = ? group by kernel_name")
    for db, counter, key in ((fetch_db, "FETCH_SIZE", "fetch_kb"), (write_db, "WRITE_SIZE", "write_kb")):
        for n, c, a, d in sqlite3.connect(db).execute(q, (counter,)):
            e = out.setdefault(plain(n), {"launches_profiled": c, "avg_ns": d})
            e[key] = e.get(key, 0.0) + a  # template instances of one kernel are merged
    for e in out.values():
        e["hbm_bytes_per_launch"] = 1024.0 * (2.0 * e.get("fetch_kb", 0.0) + e.get("write_kb", 0.0))
    return json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)",
                       "correction": "hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes)",
                       "kernels": out}, indent=1)


def hbmtable(specs):
    """specs: tag=path/to/pmc_hbm.json ... -> the per-kernel HBM traffic / bandwidth table."""
    import json
    out = ["# HBM traffic per launch and bandwidth per kernel, from the rocprofv3 PMC passes in this directory",
           "# (r01_<cfg>_pmc_hbm.json: FETCH_SIZE and WRITE_SIZE in separate passes, KB per dispatch).",
           "# 'corrected' doubles FETCH_SIZE (gfx950 reports half the bytes of wide coalesced reads,",
           "# MI355X_MICROARCH.md HBM section); 'raw' takes both counters as reported. Durations are",
           "# those of the profiled (counter-collecting) runs. Peak 8000 GB/s."]
    for spec in specs:
        tag, path = spec.split("=", 1)
        ks = json.load(open(path))["kernels"]
        out += ["", tag, "  %-26s %10s %10s %9s %16s %16s" % ("kernel", "fetch KB", "write KB", "avg us",
                                                               "raw GB/s (%pk)", "corr. GB/s (%pk)")]
        rows = [(k, v) for k, v in ks.items() if not k.startswith("__amd")]
        rows.sort(key=lambda kv: -(2 * kv[1]["fetch_kb"] + kv[1]["write_kb"]))
        for k, v in rows:
            raw = (v["fetch_kb"] + v["write_kb"]) * 1024 / v["avg_ns"]
            cor = (2 * v["fetch_kb"] + v["write_kb"]) * 1024 / v["avg_ns"]
            out.append("  %-26s %10.0f %10.0f %9.1f %9.0f (%4.1f%%) %9.0f (%4.1f%%)" % (
                k, v["fetch_kb"], v["write_kb"], v["avg_ns"] / 1e3, raw, raw / 80.0, cor, cor / 80.0))
    return "\n".join(out)


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "hbmjson":
        print(hbmjson(sys.argv[2], sys.argv[3]))
    elif mode == "hbmtable":
        print(hbmtable(sys.argv[2:]))
    else:
        print(stats(sys.argv[2]) if mode == "stats" else pmc(sys.argv[2]))
