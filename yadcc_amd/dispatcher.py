"""ctypes view of the host class GpuTaskDispatcher (ydc_td_*, include/yadcc_dispatch.h):
the reference's TaskDispatcher surface (yadcc/scheduler/task_dispatcher.h:139-181),
method for method. Placement always runs on the GPU through libydc.so; a dispatcher
created with device=-1 only keeps the registry / lease state and fails every wait with
YDC_ERR_NO_DEVICE (no CPU placement exists in this package).
"""
import ctypes as C
import json

import numpy as np

from . import binding

GRANTED, ENV_NOT_FOUND, TIMEOUT = 0, 1, 2
PRIORITY_DEDICATED, PRIORITY_USER = 1, 2

TD_SYMBOLS = (
    "ydc_td_create", "ydc_td_destroy", "ydc_td_device_status", "ydc_td_set_clock_ns",
    "ydc_td_keep_servant_alive", "ydc_td_wait_for_starting_new_task",
    "ydc_td_wait_for_starting_new_tasks", "ydc_td_keep_task_alive", "ydc_td_free_task",
    "ydc_td_free_tasks", "ydc_td_host_stats", "ydc_td_running_tasks_acquire", "ydc_td_running_tasks_release",
    "ydc_td_notify_servant_running_tasks", "ydc_td_get_running_tasks",
    "ydc_td_on_expiration_timer", "ydc_td_dump_internals", "ydc_td_oplog_enable", "ydc_td_oplog_take",
)


class _Servant(C.Structure):
    _fields_ = [("version", C.c_int32), ("observed_location", C.c_char_p),
                ("reported_location", C.c_char_p), ("env_digests", C.POINTER(C.c_char_p)),
                ("n_envs", C.c_size_t), ("num_processors", C.c_uint64),
                ("current_load", C.c_uint64), ("total_memory_in_bytes", C.c_uint64),
                ("memory_available_in_bytes", C.c_uint64), ("max_tasks", C.c_uint64),
                ("priority", C.c_int32), ("not_accepting_task_reason", C.c_int32)]


class _RunningTask(C.Structure):
    _fields_ = [("servant_task_id", C.c_uint64), ("task_grant_id", C.c_uint64),
                ("servant_location", C.c_char_p), ("task_digest", C.c_char_p)]


class _RunningView(C.Structure):
    _fields_ = [("n", C.c_size_t), ("servant_task_ids", C.POINTER(C.c_uint64)),
                ("task_grant_ids", C.POINTER(C.c_uint64)), ("location_off", C.POINTER(C.c_uint32)),
                ("location_len", C.POINTER(C.c_uint32)), ("digest_off", C.POINTER(C.c_uint32)),
                ("digest_len", C.POINTER(C.c_uint32)), ("strings", C.c_void_p)]


class _TdStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("requests", "batches", "device_ns", "host_ns", "heartbeats",
                                          "heartbeats_unchanged", "bookkeeper_rebuilds", "lease_pages",
                                          "timer_ticks", "timer_lease_entries_seen", "timer_last_ns",
                                          "timer_max_ns", "lease_wheel_entries")]


def type_td_functions(L):
    """Declares the ydc_td_* prototypes on a loaded library (once per library object)."""
    if not getattr(L, "_ydc_td_typed", False):
        u64p = C.POINTER(C.c_uint64)
        L.ydc_strerror.restype = C.c_char_p
        L.ydc_strerror.argtypes = [C.c_int]
        L.ydc_td_create.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.ydc_td_destroy.argtypes = [C.c_void_p]
        L.ydc_td_device_status.argtypes = [C.c_void_p]
        L.ydc_td_set_clock_ns.argtypes = [C.c_void_p, C.c_int64]
        L.ydc_td_keep_servant_alive.argtypes = [C.c_void_p, C.POINTER(_Servant), C.c_int64]
        L.ydc_td_wait_for_starting_new_task.argtypes = [
            C.c_void_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_int64, C.c_int64, C.c_int, u64p,
            C.c_char_p, C.c_size_t]
        L.ydc_td_wait_for_starting_new_tasks.argtypes = [
            C.c_void_p, C.c_size_t, C.POINTER(C.c_char_p), C.c_void_p, C.POINTER(C.c_char_p),
            C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        L.ydc_td_keep_task_alive.argtypes = [C.c_void_p, C.c_uint64, C.c_int64]
        L.ydc_td_free_task.argtypes = [C.c_void_p, C.c_uint64]
        L.ydc_td_free_tasks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ydc_td_host_stats.argtypes = [C.c_void_p, C.POINTER(_TdStats)]
        L.ydc_td_running_tasks_acquire.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(_RunningView)]
        L.ydc_td_running_tasks_release.argtypes = [C.c_void_p]
        L.ydc_td_notify_servant_running_tasks.argtypes = [
            C.c_void_p, C.c_char_p, C.POINTER(_RunningTask), C.c_size_t, C.c_void_p, C.c_size_t]
        L.ydc_td_notify_servant_running_tasks.restype = C.c_int64
        L.ydc_td_get_running_tasks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p,
                                               C.c_size_t, C.c_char_p, C.c_size_t, C.c_size_t]
        L.ydc_td_get_running_tasks.restype = C.c_int64
        L.ydc_td_on_expiration_timer.argtypes = [C.c_void_p]
        L.ydc_td_dump_internals.argtypes = [C.c_void_p]
        L.ydc_td_dump_internals.restype = C.c_char_p
        L.ydc_td_oplog_enable.argtypes = [C.c_void_p, C.c_int]
        L.ydc_td_oplog_take.argtypes = [C.c_void_p]
        L.ydc_td_oplog_take.restype = C.c_char_p
        L._ydc_td_typed = True
    return L


def _lib():
    return type_td_functions(binding.lib())


MS = 1_000_000  # ns


class GpuTaskDispatcher:
    """Same method names / argument meaning as oracle.refbind.RefDispatcher, which wraps the
    reference class itself — the parity tests drive both with the same calls."""

    LOC = 128

    @staticmethod
    def _load():
        """The library behind this object: libydc.so (HIP; no CPU placement exists in it)."""
        return _lib()

    def __init__(self, device=0, min_memory=None, start_timer=False, fake_clock=True):
        self._L = self._load()
        h = C.c_void_p()
        binding.compose_tune()
        rc = self._L.ydc_td_create(device, min_memory.encode() if min_memory else None,
                                  int(start_timer), int(fake_clock), C.byref(h))
        if rc:
            raise binding.YdcError("ydc_td_create: %s" % self._L.ydc_strerror(rc).decode())
        self._h = h
        self._now_ns = 0

    def close(self):
        if self._h:
            self._L.ydc_td_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_status(self):
        return self._L.ydc_td_device_status(self._h)

    def clock_advance_ms(self, ms):
        self._now_ns += int(ms) * MS
        self._L.ydc_td_set_clock_ns(self._h, self._now_ns)

    def keep_servant_alive(self, location, envs, max_tasks, num_processors, current_load,
                           priority=PRIORITY_USER, version=8, total_memory=0,
                           memory_available=50 << 30, expires_in_ms=10000, reported=None,
                           reason=0):
        arr = (C.c_char_p * max(len(envs), 1))(*[e.encode() for e in envs])
        s = _Servant(version, location.encode(), (reported or location).encode(), arr, len(envs),
                     num_processors, current_load, total_memory, memory_available, max_tasks,
                     priority, reason)
        rc = self._L.ydc_td_keep_servant_alive(self._h, C.byref(s), expires_in_ms * MS)
        assert rc == 0

    def wait_for_starting_new_task(self, requestor_ip, digest, min_version=8,
                                   expires_in_ms=1000, timeout_in_ms=0, prefetching=False):
        tid = C.c_uint64(0)
        buf = C.create_string_buffer(self.LOC)
        st = self._L.ydc_td_wait_for_starting_new_task(
            self._h, requestor_ip.encode(), min_version, digest.encode(), expires_in_ms * MS,
            timeout_in_ms * MS, int(prefetching), C.byref(tid), buf, self.LOC)
        if st < 0:
            raise binding.YdcError("wait_for_starting_new_task: %s" %
                                   self._L.ydc_strerror(st).decode())
        if st != GRANTED:
            return st, None, None
        return GRANTED, tid.value, buf.value.decode()

    def wait_for_starting_new_tasks(self, requestor_ips, digests, min_versions,
                                    expires_in_ms=1000, prefetching=None):
        """One device batch == len(digests) back-to-back calls with timeout == now.
        Returns (status[n], task_id[n], location[n])."""
        n = len(digests)
        ips = (C.c_char_p * max(n, 1))(*[x.encode() for x in requestor_ips])
        dg = (C.c_char_p * max(n, 1))(*[x.encode() for x in digests])
        mv = np.ascontiguousarray(min_versions, dtype=np.uint32)
        pf = None if prefetching is None else np.ascontiguousarray(prefetching, dtype=np.uint8)
        st = np.empty(n, dtype=np.int32)
        ids = np.empty(n, dtype=np.uint64)
        locs = C.create_string_buffer(max(n, 1) * self.LOC)
        rc = self._L.ydc_td_wait_for_starting_new_tasks(
            self._h, n, ips, mv.ctypes.data, dg, expires_in_ms * MS,
            pf.ctypes.data if pf is not None else None, st.ctypes.data, ids.ctypes.data, locs,
            self.LOC)
        if rc < 0:
            raise binding.YdcError("wait_for_starting_new_tasks: %s" %
                                   self._L.ydc_strerror(rc).decode())
        raw = locs.raw
        out_locs = [raw[i * self.LOC:(i + 1) * self.LOC].split(b"\0", 1)[0].decode()
                    for i in range(n)]
        return st, ids, out_locs

    def keep_task_alive(self, task_id, ms):
        return bool(self._L.ydc_td_keep_task_alive(self._h, task_id, ms * MS))

    def free_task(self, task_id):
        self._L.ydc_td_free_task(self._h, task_id)

    def free_tasks(self, task_ids):
        a = np.ascontiguousarray(task_ids, dtype=np.uint64)
        assert self._L.ydc_td_free_tasks(self._h, a.ctypes.data, len(a)) == 0

    def host_stats(self):
        st = _TdStats()
        assert self._L.ydc_td_host_stats(self._h, C.byref(st)) == 0
        return {k: int(getattr(st, k)) for k, _ in _TdStats._fields_}

    def notify_servant_running_tasks(self, location, grant_ids, servant_task_ids=None,
                                     digests=None):
        n = len(grant_ids)
        arr = (_RunningTask * max(n, 1))()
        for i, g in enumerate(grant_ids):
            arr[i].servant_task_id = int(servant_task_ids[i]) if servant_task_ids is not None else i
            arr[i].task_grant_id = int(g)
            arr[i].servant_location = location.encode()
            arr[i].task_digest = digests[i].encode() if digests else None
        out = np.zeros(max(n, 1), dtype=np.uint64)
        cnt = self._L.ydc_td_notify_servant_running_tasks(self._h, location.encode(), arr, n,
                                                         out.ctypes.data, len(out))
        assert cnt >= 0
        return [int(x) for x in out[:cnt]]

    def get_running_tasks(self, cap=1 << 16, with_strings=False):
        st = np.zeros(cap, dtype=np.uint64)
        gr = np.zeros(cap, dtype=np.uint64)
        locs = C.create_string_buffer(cap * self.LOC) if with_strings else None
        n = self._L.ydc_td_get_running_tasks(self._h, st.ctypes.data, gr.ctypes.data, locs,
                                            self.LOC if with_strings else 0, None, 0, cap)
        pairs = list(zip(st[:n].tolist(), gr[:n].tolist()))
        if not with_strings:
            return pairs
        raw = locs.raw
        return [(a, b, raw[i * self.LOC:(i + 1) * self.LOC].split(b"\0", 1)[0].decode())
                for i, (a, b) in enumerate(pairs)]

    def running_tasks_view(self):
        """The shared snapshot behind GetRunningTasks without the copy (ydc_td_running_tasks_acquire
        / _release): [(servant_task_id, task_grant_id, servant_location, task_digest), ...]."""
        h, v = C.c_void_p(), _RunningView()
        assert self._L.ydc_td_running_tasks_acquire(self._h, C.byref(h), C.byref(v)) == 0
        try:
            out = []
            for i in range(v.n):
                loc = C.string_at(v.strings + v.location_off[i], v.location_len[i]).decode()
                dig = C.string_at(v.strings + v.digest_off[i], v.digest_len[i]).decode()
                assert C.string_at(v.strings + v.location_off[i]) == loc.encode()  # NUL-terminated in the pool
                out.append((int(v.servant_task_ids[i]), int(v.task_grant_ids[i]), loc, dig))
            return out
        finally:
            self._L.ydc_td_running_tasks_release(h)

    def on_expiration_timer(self):
        self._L.ydc_td_on_expiration_timer(self._h)

    def dump_internals(self):
        return json.loads(self._L.ydc_td_dump_internals(self._h).decode())

    def oplog_enable(self, on=True):
        """Test switch: record the order in which calls take effect (ydc_td_oplog_enable)."""
        assert self._L.ydc_td_oplog_enable(self._h, int(on)) == 0

    def oplog_take(self):
        return json.loads(self._L.ydc_td_oplog_take(self._h).decode())

    def set_clock_ns(self, ns):
        self._now_ns = int(ns)
        self._L.ydc_td_set_clock_ns(self._h, self._now_ns)
