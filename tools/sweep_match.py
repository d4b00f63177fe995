#!/usr/bin/env python3
"""Developer tool: sweeps the matching kernel's chunk count (YDC_TARGET_CHUNKS) and the sort
tile size (YDC_SORT_ITEMS) through bench.py and prints one line per setting."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfgs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["cfg2", "cfg3"]
for cfg in cfgs:
    for chunks in (1024, 2048, 4096, 8192, 16384):
        for lds in (2, 4, 8):  # sort items per thread
            env = dict(os.environ, YDC_TARGET_CHUNKS=str(chunks), YDC_SORT_ITEMS=str(lds))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg,
                                  "--steps", "30", "--warmup", "5", "--no-cpu-baseline"],
                                 env=env, capture_output=True, text=True)
            try:
                j = json.loads(out.stdout.strip().splitlines()[-1])
            except Exception:
                print(cfg, chunks, lds, "FAILED", out.stderr[-300:])
                continue
            k = j["kernels_us_per_step"]
            print("%s chunks=%5d sort_items=%d  %.3f ms/step  %.0f M/s  rounds=%d sims=%d match=%.0f us" % (
                cfg, j["stats"]["n_chunks"], lds, j["ms_per_step"], j["value"] / 1e6,
                j["stats"]["rounds"], j["stats"]["chunk_sims"], k.get("k_match_round", 0)), flush=True)
