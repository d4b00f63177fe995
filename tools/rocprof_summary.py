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


def calib(fetch_db, write_db):
    """tools/hbm_calib under the two PMC passes: counter value against the bytes every
    calibration kernel is known to move (1 GiB streamed once; the strided kernels touch one
    4-byte word in each of the 2^24 64-byte lines of the same GiB)."""
    gib = float(1 << 30)
    known = {"k_calib_read": gib, "k_calib_write": gib,
             "k_calib_read_strided": gib / 16, "k_calib_write_strided": gib / 16}
    q = ("select kernel_name, count(*), avg(value), avg(duration) from counters_collection "
         "where counter_name = ? group by kernel_name order by kernel_name")
    out = ["%-44s %-11s %6s %14s %14s %9s %9s" % ("kernel", "counter", "calls", "counter_KB",
                                                  "useful_bytes", "factor", "GB/s")]
    for db, counter in ((fetch_db, "FETCH_SIZE"), (write_db, "WRITE_SIZE")):
        for n, c, a, d in sqlite3.connect(db).execute(q, (counter,)):
            nm = short(n).replace("ydc::", "")
            base = nm.split("<")[0]
            if ("read" in base) != (counter == "FETCH_SIZE") or base not in known:
                continue
            useful = known[base]
            out.append("%-44s %-11s %6d %14.1f %14.0f %9.3f %9.1f" % (
                nm[:44], counter, c, a, useful, useful / (a * 1024.0) if a else float("nan"),
                useful / d if d else 0.0))
    out.append("factor = useful bytes / (counter_KB * 1024): what a counter reading has to be "
               "multiplied with to give bytes for that access width; GB/s = useful bytes / duration.")
    return "\n".join(out)


def plain(name):
    """void ydc::k_match_pass<1>(...) -> k_match_pass (the name bench.py uses)."""
    n = short(name)
    n = n.split("<")[0]
    return n.split("::")[-1].strip()


def hbmjson(fetch_db, write_db, f_fetch=2.0, f_write=1.0):
    """Fabric bytes per launch per kernel from the two PMC passes. FETCH_SIZE / WRITE_SIZE are
    reported in KB; f_fetch / f_write: calibration factors for the access width of these
    kernels (dword loads / stores), measured with tools/hbm_calib on the same box
    (profiles/*_hbm_calibration.txt). Raw counter values are kept beside the calibrated sum."""
    import json
    out = {}
    q = ("select kernel_name, count(*), avg(value), avg(duration) from counters_collection "
         "where counter_name = ? group by kernel_name")
    for db, counter, key in ((fetch_db, "FETCH_SIZE", "fetch_kb"), (write_db, "WRITE_SIZE", "write_kb")):
        for n, c, a, d in sqlite3.connect(db).execute(q, (counter,)):
            e = out.setdefault(plain(n), {"launches_profiled": c, "avg_ns": d})
            e[key] = e.get(key, 0.0) + a  # template instances of one kernel are merged
    for e in out.values():
        e["raw_bytes_per_launch"] = 1024.0 * (e.get("fetch_kb", 0.0) + e.get("write_kb", 0.0))
        e["hbm_bytes_per_launch"] = 1024.0 * (f_fetch * e.get("fetch_kb", 0.0) +
                                              f_write * e.get("write_kb", 0.0))
    return json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)",
                       "correction": "hbm_bytes = %.3f * FETCH_SIZE + %.3f * WRITE_SIZE (KB -> bytes); "
                                     "factors from tools/hbm_calib (dword streaming), raw sum beside it"
                                     % (f_fetch, f_write),
                       "note": "memory-side (fabric) request counters: Infinity-Cache hits are counted, "
                               "and every working set here fits the 256 MiB Infinity Cache",
                       "kernels": out}, indent=1)


def hbmtable(specs):
    """specs: tag=path/to/pmc_hbm.json ... -> the per-kernel fabric traffic / bandwidth table."""
    import json
    out = ["# Fabric traffic per launch and bandwidth per kernel, from the rocprofv3 PMC passes in this directory",
           "# (<round>_<cfg>_pmc_hbm.json: FETCH_SIZE and WRITE_SIZE in separate passes, KB per dispatch).",
           "# 'calibrated' = 2 x FETCH_SIZE + 1 x WRITE_SIZE: the factors tools/hbm_calib measured on this",
           "# hardware for byte / dword / dwordx2 / dwordx4 streaming (profiles/*_hbm_calibration.txt: FETCH_SIZE",
           "# reports half the bytes at every width, WRITE_SIZE all of them; a scattered 4-byte access moves a",
           "# 32-byte sector). 'raw' takes both counters as reported. These are memory-side request counters:",
           "# Infinity-Cache hits are counted, and the working sets here (<= ~100 MB) fit the 256 MiB cache, so",
           "# this is fabric traffic, an upper bound on HBM traffic. Durations are those of the profiled",
           "# (counter-collecting) runs. Peak 8000 GB/s."]
    for spec in specs:
        tag, path = spec.split("=", 1)
        ks = json.load(open(path))["kernels"]
        out += ["", tag, "  %-26s %10s %10s %9s %16s %16s" % ("kernel", "fetch KB", "write KB", "avg us",
                                                               "raw GB/s (%pk)", "calib GB/s (%pk)")]
        rows = [(k, v) for k, v in ks.items() if not k.startswith("__amd")]
        rows.sort(key=lambda kv: -(2 * kv[1].get("fetch_kb", 0) + kv[1].get("write_kb", 0)))
        for k, v in rows:
            f, w = v.get("fetch_kb", 0.0), v.get("write_kb", 0.0)
            raw = (f + w) * 1024 / v["avg_ns"]
            cor = (2 * f + w) * 1024 / v["avg_ns"]
            out.append("  %-26s %10.0f %10.0f %9.1f %9.0f (%4.1f%%) %9.0f (%4.1f%%)" % (
                k, f, w, v["avg_ns"] / 1e3, raw, raw / 80.0, cor, cor / 80.0))
    return "\n".join(out)


def timeline(path, last=40):
    """Start / duration / gap to the previous kernel's end of the last `last` dispatches (all queues)."""
    db = sqlite3.connect(path)
    try:
        rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
    except sqlite3.Error:
        try:
            rows = list(db.execute("select name, start, end, 0 from kernels order by start"))
        except sqlite3.Error as e:
            names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
            return "no kernels view (%s); have: %s" % (e, ", ".join(names))
    rows = rows[-last:]
    t0 = rows[0][1]
    out = ["%-44s %6s %12s %10s %10s" % ("kernel", "queue", "start_us", "dur_us", "gap_us")]
    prev_end = None
    for n, a, b, q in rows:
        gap = (a - prev_end) / 1e3 if prev_end is not None else 0.0
        out.append("%-44s %6s %12.2f %10.2f %10.2f" % (short(n)[:44], q, (a - t0) / 1e3, (b - a) / 1e3, gap))
        prev_end = max(prev_end or b, b)
    return "\n".join(out)


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "hbmjson":
        ff = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
        fw = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
        print(hbmjson(sys.argv[2], sys.argv[3], ff, fw))
    elif mode == "calib":
        print(calib(sys.argv[2], sys.argv[3]))
    elif mode == "hbmtable":
        print(hbmtable(sys.argv[2:]))
    elif mode == "timeline":
        print(timeline(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40))
    else:
        print(stats(sys.argv[2]) if mode == "stats" else pmc(sys.argv[2]))
