"""Oracle (test infrastructure, NOT product code).

ctypes binding of oracle/_ref/libyadcc_ref.so — the reference's own
task_dispatcher.cc compiled verbatim against oracle/shims (see ref_driver.h).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libyadcc_ref.so")

OK, ENV_NOT_FOUND, TIMEOUT = 0, 1, 2
IDX_TIMEOUT = 0xFFFFFFFF
IDX_ENV_NOT_FOUND = 0xFFFFFFFE

_lib = None


def available():
    return os.path.exists(REF_SO)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(REF_SO)
        u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.ref_create.restype = C.c_void_p
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_clock_advance_ms.argtypes = [C.c_int64]
        L.ref_clock_now_ns.restype = C.c_int64
        L.ref_keep_servant_alive.argtypes = [
            C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_size_t,
            C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
            C.c_int64]
        L.ref_wait_for_starting_new_task.argtypes = [
            C.c_void_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_int64, C.c_int64, C.c_int,
            u64p, C.c_char_p, C.c_size_t]
        L.ref_wait_for_starting_new_task.restype = C.c_int
        L.ref_keep_task_alive.argtypes = [C.c_void_p, C.c_uint64, C.c_int64]
        L.ref_keep_task_alive.restype = C.c_int
        L.ref_free_task.argtypes = [C.c_void_p, C.c_uint64]
        L.ref_notify_servant_running_tasks.argtypes = [
            C.c_void_p, C.c_char_p, u64p, u64p, C.c_size_t, u64p, C.c_size_t]
        L.ref_notify_servant_running_tasks.restype = C.c_size_t
        L.ref_get_running_tasks.argtypes = [C.c_void_p, u64p, u64p, C.c_size_t]
        L.ref_get_running_tasks.restype = C.c_size_t
        L.ref_load_servants.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 11
        L.ref_load_servants_wide.argtypes = ([C.c_void_p, C.c_size_t] + [C.c_void_p] * 9 +
                                             [C.c_uint32] + [C.c_void_p] * 2)
        L.ref_free_tasks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ref_dump_internals.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.ref_dump_internals.restype = C.c_size_t
        L.ref_dispatch_batch.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 6
        L.ref_dispatch_batch.restype = C.c_double
        L.ref_digest_name.argtypes = [C.c_uint32, C.c_char_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class RefDispatcher:
    """The reference TaskDispatcher (task_dispatcher.h:120-303), method for method."""

    def __init__(self):
        self._h = lib().ref_create()

    def close(self):
        if self._h:
            lib().ref_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def keep_servant_alive(self, location, envs, max_tasks, num_processors, current_load,
                           priority=2, version=8, total_memory=0,
                           memory_available=50 << 30, expires_in_ms=10000, reported=None,
                           reason=0):
        arr = (C.c_char_p * len(envs))(*[e.encode() for e in envs])
        lib().ref_keep_servant_alive(
            self._h, version, location.encode(), (reported or location).encode(), arr,
            len(envs), num_processors, current_load, total_memory, memory_available,
            max_tasks, priority, reason, expires_in_ms)

    def wait_for_starting_new_task(self, requestor_ip, digest, min_version=8,
                                   expires_in_ms=1000, timeout_in_ms=0, prefetching=False):
        tid = C.c_uint64(0)
        buf = C.create_string_buffer(128)
        st = lib().ref_wait_for_starting_new_task(
            self._h, requestor_ip.encode(), min_version, digest.encode(), expires_in_ms,
            timeout_in_ms, int(prefetching), C.byref(tid), buf, 128)
        if st != OK:
            return st, None, None
        return OK, tid.value, buf.value.decode()

    def keep_task_alive(self, task_id, ms):
        return bool(lib().ref_keep_task_alive(self._h, task_id, ms))

    def free_task(self, task_id):
        lib().ref_free_task(self._h, task_id)

    def dump_internals(self):
        """TaskDispatcher::DumpInternals (task_dispatcher.cc:538-614) as a dict."""
        import json
        n = lib().ref_dump_internals(self._h, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib().ref_dump_internals(self._h, buf, n + 1)
        return json.loads(buf.value.decode())

    def free_tasks(self, task_ids):
        a = np.ascontiguousarray(task_ids, dtype=np.uint64)
        lib().ref_free_tasks(self._h, _p(a), len(a))

    def notify_servant_running_tasks(self, location, grant_ids, servant_task_ids=None):
        g = np.asarray(grant_ids, dtype=np.uint64)
        s = None if servant_task_ids is None else np.asarray(servant_task_ids, dtype=np.uint64)
        out = np.zeros(max(len(g), 1), dtype=np.uint64)
        n = lib().ref_notify_servant_running_tasks(
            self._h, location.encode(), C.cast(_p(s), C.POINTER(C.c_uint64)),
            C.cast(_p(g), C.POINTER(C.c_uint64)), len(g),
            C.cast(_p(out), C.POINTER(C.c_uint64)), len(out))
        return [int(x) for x in out[:n]]

    def get_running_tasks(self, cap=1 << 16):
        st = np.zeros(cap, dtype=np.uint64)
        gr = np.zeros(cap, dtype=np.uint64)
        n = lib().ref_get_running_tasks(
            self._h, C.cast(_p(st), C.POINTER(C.c_uint64)),
            C.cast(_p(gr), C.POINTER(C.c_uint64)), cap)
        return list(zip(st[:n].tolist(), gr[:n].tolist()))

    # -- snapshot helpers -------------------------------------------------
    def load_servants(self, sv):
        """sv: dict of numpy columns (see yadcc_amd.synth)."""
        n = len(sv["version"])
        cols = [np.ascontiguousarray(sv[k], dtype=dt) for k, dt in (
            ("version", np.uint32), ("num_processors", np.uint32), ("current_load", np.uint32),
            ("max_tasks", np.uint32), ("running_tasks", np.uint32), ("priority", np.uint32),
            ("total_memory", np.uint64), ("memory_available", np.uint64),
            ("env_mask", np.uint64), ("ip", np.uint32), ("port", np.uint32))]
        words = cols[8].shape[1] if cols[8].ndim == 2 else 1  # env_mask: (n,) or (n, env_words)
        lib().ref_load_servants_wide(self._h, n, *[_p(c) for c in cols[:9]], words,
                                     _p(cols[9]), _p(cols[10]))

    def dispatch_batch(self, tk, want_latency=False):
        n = len(tk["env_id"])
        env = np.ascontiguousarray(tk["env_id"], dtype=np.uint32)
        mv = np.ascontiguousarray(tk["min_version"], dtype=np.uint32)
        ip = np.ascontiguousarray(tk["requestor_ip"], dtype=np.uint32)
        out = np.empty(n, dtype=np.uint32)
        ids = np.empty(n, dtype=np.uint64)
        lat = np.empty(n, dtype=np.uint64) if want_latency else None
        secs = lib().ref_dispatch_batch(self._h, n, _p(env), _p(mv), _p(ip), _p(out), _p(ids),
                                        _p(lat))
        return out, ids, secs, lat


def clock_advance_ms(ms):
    lib().ref_clock_advance_ms(ms)


def fire_timers():
    lib().ref_fire_timers()


def digest_name(env_id):
    buf = C.create_string_buffer(65)
    lib().ref_digest_name(env_id, buf)
    return buf.value.decode()
