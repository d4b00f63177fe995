"""One rank of a multi-PROCESS sharded dispatch (tests/test_sharded_multiprocess_gpu.py, and
bench.py's launcher does the same over gloo): own process, own HIP runtime, own ydc_context on
the given device; the ranks find each other through files in a directory (the "side channel"
of include/yadcc_dispatch.h) and exchange through the mailbox transport of libydc.so
(ydc_group_ipc_export / ydc_group_init_ipc) — HIP IPC device memory, or the shared host
segment when a rank cannot open a peer's IPC handle (every rank then switches alike).

usage: python -m tests.mp_rank_worker <dir> <rank> <n_ranks> <spec.json>
spec: {"case": {random_case kwargs} | "config": name, "cuts": [...], "batches": k, "commit": bool,
       "transport": "ipc" | "ipc-host", "device": 0}
Writes <dir>/result_<rank>.npz (idx, util, running of every batch, stats, transport)."""
import json
import os
import sys
import time

import numpy as np


def wait_for(paths, seconds, what):
    t0 = time.time()
    while not all(os.path.exists(p) for p in paths):
        if time.time() - t0 > seconds:
            raise TimeoutError("rank timed out waiting for %s: %s" % (
                what, [p for p in paths if not os.path.exists(p)]))
        time.sleep(0.01)


def publish(path, data):
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, path)  # atomic: readers never see a partial file


def main():
    d, rank, G = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    spec = json.load(open(sys.argv[4]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(spec.get("env", {}))  # (options the library reads when a context is created)
    from tests import cases
    from yadcc_amd import binding, pack, synth
    DA = binding.DeviceArray
    if "config" in spec:
        sv, tk = synth.make_config(spec["config"], **spec.get("config_kw", {}))
    else:
        sv, tk = cases.random_case(**spec["case"])
    dev = int(spec.get("device", 0))
    ctx = binding.Context(device=dev)
    ctx.upload_servants(pack.to_abi_columns(sv))
    want = {"ipc": binding.TRANSPORT_IPC_DEVICE, "ipc-host": binding.TRANSPORT_IPC_HOST}[
        spec.get("transport", "ipc")]
    publish(os.path.join(d, "handle_%d" % rank), ctx.group_ipc_export(rank, G))
    hpaths = [os.path.join(d, "handle_%d" % r) for r in range(G)]
    wait_for(hpaths, 120, "handles")
    handles = [open(p, "rb").read() for p in hpaths]
    # Agree on the flavour: everybody tries the wanted one and says how it went; one failure
    # anywhere sends everybody to the host segment.
    note = ""
    try:
        ctx.group_init_ipc(handles, rank, G, want)
        ok = True
    except binding.YdcError as e:
        ok, note = False, str(e)
    publish(os.path.join(d, "try_%d" % rank), b"1" if ok else b"0")
    tpaths = [os.path.join(d, "try_%d" % r) for r in range(G)]
    wait_for(tpaths, 120, "first-attempt verdicts")
    if not all(open(p, "rb").read() == b"1" for p in tpaths):
        if want == binding.TRANSPORT_IPC_HOST:
            raise RuntimeError("host mailbox transport failed: %s" % note)
        ctx.group_init_ipc(handles, rank, G, binding.TRANSPORT_IPC_HOST)
    transport = ctx.group_transport()
    cuts = spec["cuts"]
    nb = int(spec.get("batches", 1))
    n = len(tk["env_id"])
    per = n // nb
    out = {}
    for b in range(nb):
        b0, b1 = b * per, (n if b == nb - 1 else (b + 1) * per)
        c = cuts[b] if isinstance(cuts[0], list) else cuts
        lo, hi = b0 + c[rank], b0 + c[rank + 1]
        assert c[0] == 0 and b0 + c[-1] == b1, (c, b0, b1)
        cols = [DA.from_numpy(tk[k][lo:hi], dev) for k in ("env_id", "min_version", "requestor_ip")]
        d_out = DA(hi - lo, np.uint32, dev)
        d_util = DA(hi - lo, np.float64, dev)
        d_run = DA(len(sv["version"]), np.uint32, dev)
        t0 = time.perf_counter()
        ctx.dispatch_sharded(cols[0], cols[1], cols[2], d_out, d_util, d_run,
                             commit=bool(spec.get("commit")))
        out["ms_%d" % b] = np.float64((time.perf_counter() - t0) * 1e3)
        out["idx_%d" % b] = d_out.numpy()
        out["util_%d" % b] = d_util.numpy()
        out["run_%d" % b] = d_run.numpy()
        st = ctx.stats()
        out["stats_%d" % b] = np.array(json.dumps({k: v for k, v in st.items() if k != "stage_ms"}))
    out["resident_running"] = ctx.get_running()
    out["transport"] = np.int32(transport)
    out["note"] = np.array(note)
    tmp = os.path.join(d, "result_%d.tmp.npz" % rank)
    np.savez(tmp, **out)
    os.replace(tmp, os.path.join(d, "result_%d.npz" % rank))
    # Nobody tears its mailbox down while a peer may still be mapping or reading it.
    wait_for([os.path.join(d, "result_%d.npz" % r) for r in range(G)], 300, "the other ranks")
    ctx.group_destroy()
    ctx.close()
    print("RANK-%d-OK transport=%s" % (rank, binding.TRANSPORT_NAMES[transport]))


if __name__ == "__main__":
    main()
