"""Test tool: ctypes binding of tests/model/libmodel.so (CPU replay of the device pipeline)."""
import ctypes as C
import os

import numpy as np

from yadcc_amd import pack

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("n_slots", "n_classes", "key_bits", "n_chunks",
                                          "rounds", "chunk_sims", "force_fp64")]


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(_HERE, "libmodel.so"))
        _lib.model_dispatch_wide.restype = C.c_int
    return _lib


def dispatch(sv, tk, chunk_size=256, force_fp64=False, min_memory=pack.MIN_MEMORY_DEFAULT):
    a = pack.to_abi_columns(sv, min_memory)
    S = len(a["version"])
    t = {k: np.ascontiguousarray(tk[k], dtype=np.uint32) for k in ("env_id", "min_version",
                                                                  "requestor_ip")}
    N = len(t["env_id"])
    out = np.empty(N, np.uint32)
    util = np.empty(N, np.float64)
    run = np.empty(S, np.uint32)
    st = Stats()
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    words = a["env_mask"].shape[1] if a["env_mask"].ndim == 2 else 1
    rc = lib().model_dispatch_wide(
        C.c_uint32(S), p(a["version"]), p(a["num_processors"]), p(a["current_load"]),
        p(a["max_tasks"]), p(a["running_tasks"]), p(a["flags"]), p(a["env_mask"]),
        C.c_uint32(words), p(a["ip_id"]),
        C.c_uint32(N), p(t["env_id"]), p(t["min_version"]), p(t["requestor_ip"]),
        C.c_uint32(chunk_size), C.c_int(int(force_fp64)), p(out), p(util), p(run), C.byref(st))
    if rc:
        raise RuntimeError("model_dispatch rc=%d" % rc)
    return out, util, run, st
