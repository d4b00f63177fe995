#!/usr/bin/env python3
"""Renders td_surface.latency of a bench line (python bench.py's JSON) as the text table kept under
profiles/.   usage: python tools/latency_table.py profiles/r05_bench_driver_line.json"""
import json
import sys


def main():
    path = sys.argv[1]
    d = json.load(open(path))
    L = d["td_surface"]["latency"]
    print("td_surface.latency of %s (python bench.py --steps %d --warmup %d, one MI355X box):" % (path, d["steps"], d["warmup"]))
    print("microseconds per CALL, p50 / p99 — ydc_td_wait_for_starting_new_task(s), registry warm, grants freed between calls,")
    print("a heartbeat before every fourth call. tick = resident small-batch kernel (default); launch = one launch per call")
    print("(YDC_TUNE=resident=0); pipeline = the batch pipeline alone (small_batch=0, the round-4 path); reference = the verbatim")
    print("reference class on the same host, one thread (a batch of n = n consecutive WaitForStartingNewTask calls).")
    for S in sorted((k for k in L if k.startswith("servants_")), key=lambda k: int(k.split("_")[1])):
        t = L[S]
        x = t.get("crossover_batch")
        print("\n%s   (%s)" % (S, "the pipeline is faster from batches of %s on" % x if x else
                              "the pipeline is not faster at any size measured; beyond 64 requests it is what runs"))
        print("  %-12s %16s %16s %18s %19s" % ("call", "tick", "launch", "pipeline", "reference"))
        cols = [t.get("tick") or {}, t.get("tick_launch_per_call") or {}, t.get("batch_pipeline_only") or {},
                t.get("reference") or {}]
        for call in cols[0]:
            if call == "batch_1":
                continue
            cells = []
            for c in cols:
                v = c.get(call)
                cells.append("%7.1f / %6.1f" % (v["p50"], v["p99"]) if isinstance(v, dict) and "p50" in v else "      -        ")
            print("  %-12s %16s %16s %18s %19s" % (call, *cells))


if __name__ == "__main__":
    main()
