#!/usr/bin/env python3
"""Developer tool: throughput of the host class path (strings in, strings out) —
ydc_td_wait_for_starting_new_tasks on a 2k-servant registry, 10k requests per call, every
grant freed again between calls."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yadcc_amd import dispatcher as D  # noqa: E402

td = D.GpuTaskDispatcher(device=0, fake_clock=False)
rng = np.random.default_rng(1)
digests = ["%064x" % (0xc0ffee + i) for i in range(4)]
for i in range(2000):
    nproc = int(rng.choice([64, 96, 128, 192, 256]))
    ded = rng.random() < 0.3
    td.keep_servant_alive("10.%d.%d.%d:8335" % (i >> 16, (i >> 8) & 255, i & 255),
                          [d for d in digests if rng.random() < 0.5] or digests[:1],
                          (nproc * (95 if ded else 40)) // 100, nproc, int(rng.integers(0, nproc)),
                          priority=1 if ded else 2, version=20, total_memory=256 << 30,
                          memory_available=64 << 30, expires_in_ms=30000)
n = 10_000
ips = ["172.16.%d.%d" % (i >> 8 & 255, i & 255) for i in range(n)]
dg = [digests[i % 4] for i in range(n)]
mv = [20] * n
for rep in range(6):
    t0 = time.perf_counter()
    st, ids, locs = td.wait_for_starting_new_tasks(ips, dg, mv, expires_in_ms=15000)
    t1 = time.perf_counter()
    granted = ids[st == 0]
    for g in granted:
        td.free_task(int(g))
    t2 = time.perf_counter()
    print("batch of %d: %d granted in %.2f ms (%.2f M/s incl. ctypes marshalling); freeing them: %.1f ms"
          % (n, len(granted), 1e3 * (t1 - t0), n / (t1 - t0) / 1e6, 1e3 * (t2 - t1)))
td.close()
