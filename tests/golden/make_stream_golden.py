"""Generates tests/golden/ref_cfg5_stream_200_ticks.npz: BASELINE.json configs[4]'s event stream
(yadcc_amd.streaming.EventStream, seed 44: 200 heartbeats + 10k frees + 10k requests per tick on
2000 servants) replayed for 200 ticks through the VERBATIM reference (oracle/_ref) — heartbeats
as KeepServantAlive, frees by grant id, the requests as sequential WaitForStartingNewTask calls —
the stream being fed the reference's own placements. Stored per tick: an order-sensitive digest
of the placement vector, the grant count and a digest of running_tasks after the tick; the GPU
test (tests/test_streaming_gpu.py::test_stream_cfg5_200_ticks_against_the_reference) replays the
same stream through ydc_stream_tick and compares tick by tick. Run in the build container
(needs /root/reference): python tests/golden/make_stream_golden.py   (~1 min)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refbind as R  # noqa: E402
from yadcc_amd import streaming, synth  # noqa: E402

TICKS = 200


def main():
    assert R.available(), "oracle/_ref is not built (needs /root/reference)"
    sv, _ = synth.make_config("cfg5")
    es = streaming.EventStream(sv, 10_000, 10_000)
    ref = R.RefDispatcher()
    ref.load_servants(sv)
    ref_ids = np.empty(0, np.uint64)  # grant id of every live grant, stream order
    digest = np.zeros(TICKS, np.uint64)
    granted = np.zeros(TICKS, np.uint32)
    run_digest = np.zeros(TICKS, np.uint64)
    t0 = time.time()
    for t in range(TICKS):
        who, rows, rel, tk = es.next_tick()
        hb = {k: v[who] for k, v in es.sv.items()}
        hb["running_tasks"] = np.zeros(len(who), np.uint32)  # (kept by a renewal anyway)
        ref.load_servants(hb)
        ref.free_tasks(ref_ids[es.last_freed])
        ref_ids = ref_ids[es.last_kept]
        ridx, rids, _, _ = ref.dispatch_batch(tk, want_latency=True)
        ok = ridx < R.IDX_ENV_NOT_FOUND
        ref_ids = np.concatenate([ref_ids, rids[ok]])
        es.commit(ridx)
        digest[t] = np.uint64(synth.placement_hash(ridx))
        granted[t] = int(ok.sum())
        run_digest[t] = np.uint64(synth.placement_hash(es.running.astype(np.uint32)))
    ref.close()
    out = os.path.join(ROOT, "tests", "golden", "ref_cfg5_stream_200_ticks.npz")
    np.savez_compressed(out, digest=digest, granted=granted, run_digest=run_digest,
                        ticks=np.uint32(TICKS))
    print("wrote %s: %d ticks in %.0f s, %d grants" % (out, TICKS, time.time() - t0, int(granted.sum())))


if __name__ == "__main__":
    main()
