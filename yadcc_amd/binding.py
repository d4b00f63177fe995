"""ctypes binding of yadcc_amd/libydc.so (include/yadcc_dispatch.h).

The library is the product; this file only marshals numpy / torch buffers. If the
shared object is missing or no GPU is present every entry point raises — there
is no CPU fallback on purpose.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (YDC_LIB: a measurement build of the same library, e.g. libydc_probe.so — tools/phase_probe.py)
LIB_PATH = os.environ.get("YDC_LIB") or os.path.join(_HERE, "libydc.so")

ABI_VERSION = 6
IPC_HANDLE_BYTES = 256
TRANSPORT_NONE, TRANSPORT_RCCL, TRANSPORT_LOCAL, TRANSPORT_IPC_DEVICE, TRANSPORT_IPC_HOST = range(5)
TRANSPORT_NAMES = ("none", "rccl", "local", "ipc", "ipc-host")
IDX_TIMEOUT = 0xFFFFFFFF
IDX_ENV_NOT_FOUND = 0xFFFFFFFE
DISPATCH_COMMIT = 1
STAGES = ("servant_scan", "slot_gen", "sort", "class_lists", "task_classify", "match", "finalize",
          "total")

# Every symbol include/yadcc_dispatch.h declares (tests check they are all exported).
ABI_SYMBOLS = (
    "ydc_strerror", "ydc_last_error", "ydc_abi_version", "ydc_create", "ydc_destroy",
    "ydc_upload_servants", "ydc_update_servants", "ydc_update_servants_wide",
    "ydc_set_host_aliases", "ydc_remove_servants", "ydc_release_slots", "ydc_release_slots_device", "ydc_set_running",
    "ydc_get_running", "ydc_dispatch", "ydc_dispatch_tick", "ydc_dispatch_device", "ydc_dispatch_device_async",
    "ydc_dispatch_wait", "ydc_synchronize",
    "ydc_set_profiling", "ydc_get_stats", "ydc_kernel_profile", "ydc_device_count",
    "ydc_device_malloc", "ydc_device_free", "ydc_memcpy_h2d", "ydc_memcpy_d2h",
    "ydc_host_register", "ydc_host_unregister", "ydc_host_alloc", "ydc_host_free",
    "ydc_stream_begin", "ydc_stream_tick", "ydc_stream_tick_wide", "ydc_stream_buffers_get", "ydc_stream_end",
    "ydc_group_unique_id", "ydc_group_init", "ydc_group_init_local", "ydc_group_destroy",
    "ydc_group_size", "ydc_group_ipc_export", "ydc_group_init_ipc", "ydc_group_transport",
    "ydc_dispatch_sharded",
    # host class wrapper (yadcc_amd/dispatcher.py types them)
    "ydc_td_create", "ydc_td_destroy", "ydc_td_device_status", "ydc_td_set_clock_ns",
    "ydc_td_keep_servant_alive", "ydc_td_wait_for_starting_new_task",
    "ydc_td_wait_for_starting_new_tasks", "ydc_td_keep_task_alive", "ydc_td_free_task",
    "ydc_td_free_tasks", "ydc_td_host_stats", "ydc_td_running_tasks_acquire", "ydc_td_running_tasks_release",
    "ydc_td_notify_servant_running_tasks", "ydc_td_get_running_tasks",
    "ydc_td_on_expiration_timer", "ydc_td_dump_internals", "ydc_td_oplog_enable", "ydc_td_oplog_take",
)


class ServantSoA(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("version", "num_processors", "current_load",
                                           "max_tasks", "running_tasks", "flags", "env_mask",
                                           "ip_id")] + [("env_words", C.c_uint32)]


# numpy view of ydc_servant_row (32 bytes)
ROW_DTYPE = np.dtype([("version", "<u4"), ("num_processors", "<u4"), ("current_load", "<u4"),
                      ("max_tasks", "<u4"), ("flags", "<u4"), ("ip_id", "<u4"), ("env_mask", "<u8")])


class ServantRow(C.Structure):
    _fields_ = [("version", C.c_uint32), ("num_processors", C.c_uint32),
                ("current_load", C.c_uint32), ("max_tasks", C.c_uint32), ("flags", C.c_uint32),
                ("ip_id", C.c_uint32), ("env_mask", C.c_uint64)]


class TaskSoA(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("env_id", "min_version", "requestor_ip")]


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in (
        "n_tasks", "n_servants", "n_classes", "n_slots", "key_bits", "radix_passes", "n_chunks",
        "rounds", "chunk_sims", "granted", "timeouts", "env_not_found", "shard_sort_batches",
        "shard_sort_misses", "small_batch", "zone_rows", "tick_resident_calls", "tick_launched_calls",
        "pipeline_batches")] + [
            ("stage_ms", C.c_float * 16)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "stage_ms"}
        d["stage_ms"] = {name: float(self.stage_ms[i]) for i, name in enumerate(STAGES)}
        return d


_lib = None


class YdcError(RuntimeError):
    pass


def lib():
    """Loads libydc.so or raises (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise YdcError("%s is missing: build it with `make lib` (hipcc --offload-arch=gfx950); "
                           "this package has no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.ydc_strerror.restype = C.c_char_p
        L.ydc_strerror.argtypes = [C.c_int]
        L.ydc_last_error.restype = C.c_char_p
        L.ydc_last_error.argtypes = [C.c_void_p]
        L.ydc_abi_version.restype = C.c_uint32
        L.ydc_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                 C.POINTER(C.c_void_p)]
        L.ydc_destroy.argtypes = [C.c_void_p]
        L.ydc_upload_servants.argtypes = [C.c_void_p, C.POINTER(ServantSoA), C.c_uint32]
        L.ydc_update_servants.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ServantRow),
                                          C.c_uint32]
        L.ydc_update_servants_wide.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ServantRow),
                                               C.c_void_p, C.c_uint32, C.c_uint32]
        L.ydc_set_host_aliases.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.ydc_remove_servants.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.ydc_release_slots.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.ydc_release_slots_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.ydc_set_running.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.ydc_get_running.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.ydc_dispatch.argtypes = [C.c_void_p, C.POINTER(TaskSoA), C.c_uint32, C.c_uint32,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
        L.ydc_dispatch_tick.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                        C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(TaskSoA),
                                        C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.ydc_dispatch_device.argtypes = L.ydc_dispatch.argtypes
        L.ydc_dispatch_device_async.argtypes = L.ydc_dispatch.argtypes
        L.ydc_dispatch_wait.argtypes = [C.c_void_p]
        L.ydc_synchronize.argtypes = [C.c_void_p]
        L.ydc_stream_begin.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.ydc_stream_tick.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                      C.c_uint32, C.POINTER(TaskSoA), C.c_uint32, C.c_void_p]
        L.ydc_stream_tick_wide.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                           C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(TaskSoA),
                                           C.c_uint32, C.c_void_p]
        L.ydc_stream_buffers_get.argtypes = [C.c_void_p, C.c_void_p]
        L.ydc_stream_end.argtypes = [C.c_void_p]
        L.ydc_group_unique_id.argtypes = [C.c_void_p]
        L.ydc_group_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ydc_group_init_local.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.ydc_group_destroy.argtypes = [C.c_void_p]
        L.ydc_group_size.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ydc_group_ipc_export.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ydc_group_init_ipc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.ydc_group_transport.argtypes = [C.c_void_p]
        L.ydc_dispatch_sharded.argtypes = L.ydc_dispatch.argtypes
        L.ydc_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.ydc_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.ydc_kernel_profile.argtypes = [C.c_void_p]
        L.ydc_kernel_profile.restype = C.c_char_p
        L.ydc_device_count.restype = C.c_int
        L.ydc_device_malloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
        L.ydc_device_free.argtypes = [C.c_void_p]
        L.ydc_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ydc_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ydc_host_register.argtypes = [C.c_void_p, C.c_size_t]
        L.ydc_host_unregister.argtypes = [C.c_void_p]
        L.ydc_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        L.ydc_host_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if isinstance(a, DeviceArray):
        return a.ptr
    if isinstance(a, int):
        return a
    return a.data_ptr()  # anything torch-like


# The library reads ONE developer variable, YDC_TUNE="key=value,...". The tests and measurement
# scripts of this repository name their switches YDC_<KEY>=<value>: compose_tune() folds those
# into YDC_TUNE right before a context is created (and takes out again what it folded in last
# time, so that a switch a test has dropped is gone). Not an interface of the package.
_TUNE_KEYS = ("DEBUG_SIM", "CHUNK_SIZE", "TARGET_CHUNKS", "FUSED_CLASS", "OWN_GUESS", "PAIR", "RING_TOTAL",
              "DENSE", "SPLIT_GEN", "XCD_TILES", "TILE_TAB", "GROUP_WALK", "WALK_PACKED", "ZONE_GUESS", "ZONE_LEAD", "ZONE_TRAIL", "ZONE_MAX_CHUNKS", "SCAN_MULTI", "CLASSIFY_PER_THREAD",
              "PACKED_CLASS", "SHARD_SORT", "PACKED_SORT", "BINSORT", "FUSE_PASSES", "WARM_UP",
              "HAND_TRIES", "CP_EVERY", "WALK_PARK", "LEVEL_TAB", "WIDE", "WALK_PREFETCH", "WIDE_LISTS", "GROUP_BINSORT",
              "ZERO_COPY", "HOST_IN", "BINSORT_VERIFY", "BINSORT_MAX_SLOTS", "SHARD_MARGIN",
              "ROUNDS_PER_CHECK", "WALK_AFTER", "OUTCOME_STORE", "STREAM_ZERO_COPY", "STREAM_GRAPH", "COMMIT_SWAP", "RELEASE_COUNTED", "SMALL_BATCH", "RESIDENT", "RESIDENT_IDLE_MS", "PACKED_TICK", "SORT_ITEMS", "IPC_SLOT_WORDS", "IPC_TIMEOUT_MS", "IPC_COARSE")
_tune_injected = ""


def compose_tune():
    """The folded YDC_<KEY> switches go FIRST: the library takes the first match of a key, so a
    test's switch wins over the same key in an inherited YDC_TUNE."""
    global _tune_injected
    cur = os.environ.get("YDC_TUNE", "")
    if _tune_injected and cur.startswith(_tune_injected):
        cur = cur[len(_tune_injected):].lstrip(",")
    mine = ",".join("%s=%s" % (k.lower(), os.environ["YDC_" + k]) for k in _TUNE_KEYS
                    if "YDC_" + k in os.environ)
    _tune_injected = mine
    both = ",".join(x for x in (mine, cur) if x)
    if both:
        os.environ["YDC_TUNE"] = both
    else:
        os.environ.pop("YDC_TUNE", None)


def device_count():
    """GPUs visible to the library's HIP runtime (0 when the library is missing)."""
    try:
        return int(lib().ydc_device_count())
    except (YdcError, OSError):
        return 0


def pinned_empty(n, dtype):
    """A numpy array in page-locked host memory the device can address (ydc_host_alloc): request
    columns and result arrays of this kind go through ydc_dispatch without any staging copy.
    The memory is freed when the array (and every view of it) is gone."""
    dt = np.dtype(dtype)
    p = C.c_void_p()
    rc = lib().ydc_host_alloc(int(n) * dt.itemsize, C.byref(p))
    if rc:
        raise YdcError("ydc_host_alloc: %s (%s)" % (lib().ydc_strerror(rc).decode(),
                                                    lib().ydc_last_error(None).decode()))
    buf = (C.c_char * max(1, int(n) * dt.itemsize)).from_address(p.value)
    import weakref
    weakref.finalize(buf, lib().ydc_host_free, p.value)  # numpy keeps `buf` alive through .base
    return np.frombuffer(buf, dtype=dt, count=int(n))


def host_register(a):
    """Page-locks the memory of a contiguous numpy array in place (ydc_host_register); the array
    must stay alive and unmoved until host_unregister(a)."""
    assert a.flags.c_contiguous
    rc = lib().ydc_host_register(a.ctypes.data, a.nbytes)
    if rc:
        raise YdcError("ydc_host_register: %s (%s)" % (lib().ydc_strerror(rc).decode(),
                                                       lib().ydc_last_error(None).decode()))


def host_unregister(a):
    rc = lib().ydc_host_unregister(a.ctypes.data)
    if rc:
        raise YdcError("ydc_host_unregister: %s" % lib().ydc_strerror(rc).decode())


def group_unique_id():
    """128-byte id for Context.group_init (ncclGetUniqueId); call on one rank, hand to all."""
    buf = C.create_string_buffer(128)
    rc = lib().ydc_group_unique_id(buf)
    if rc:
        raise YdcError("ydc_group_unique_id: %s (%s)" % (lib().ydc_strerror(rc).decode(),
                                                         lib().ydc_last_error(None).decode()))
    return buf.raw


def group_init_local(contexts):
    """Makes the given contexts (one process, one device) the ranks of a group that exchanges
    through device copies; each rank's dispatch_sharded must then run in its own thread."""
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    rc = lib().ydc_group_init_local(arr, len(contexts))
    if rc:
        raise YdcError("ydc_group_init_local: %s" % lib().ydc_strerror(rc).decode())


class DeviceArray:
    """A typed buffer in HBM owned by the caller (hipMalloc through the C-ABI)."""

    def __init__(self, n, dtype, device=0):
        self.n, self.dtype, self.device = int(n), np.dtype(dtype), device
        p = C.c_void_p()
        rc = lib().ydc_device_malloc(device, self.n * self.dtype.itemsize, C.byref(p))
        if rc:
            raise YdcError("ydc_device_malloc: %s (%s)" % (
                lib().ydc_strerror(rc).decode(), lib().ydc_last_error(None).decode()))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a, device=0):
        a = np.ascontiguousarray(a)
        d = cls(a.size, a.dtype, device)
        if a.size and lib().ydc_memcpy_h2d(d.ptr, a.ctypes.data, a.nbytes):
            raise YdcError("ydc_memcpy_h2d failed")
        return d

    def numpy(self):
        out = np.empty(self.n, self.dtype)
        if self.n and lib().ydc_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes):
            raise YdcError("ydc_memcpy_d2h failed")
        return out

    def numel(self):
        return self.n

    def free(self):
        if self.ptr:
            lib().ydc_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One ydc_context: a resident servant registry on one GPU + batch dispatch."""

    def __init__(self, device=0, max_servants=0, max_tasks=0, max_slots=0, stream=None):
        h = C.c_void_p()
        compose_tune()
        rc = lib().ydc_create(device, max_servants, max_tasks, max_slots, stream, C.byref(h))
        if rc:
            raise YdcError("ydc_create: %s (%s)" % (lib().ydc_strerror(rc).decode(),
                                                    lib().ydc_last_error(None).decode()))
        self._h = h
        self.n_servants = 0
        self._stream_caps = (0, 0, 0)

    def close(self):
        if getattr(self, "_h", None):
            lib().ydc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc:
            raise YdcError("%s: %s (%s)" % (what, lib().ydc_strerror(rc).decode(),
                                            lib().ydc_last_error(self._h).decode()))

    def upload_servants(self, cols):
        """cols: dict of numpy columns named like ydc_servant_soa (see pack.to_abi_columns).
        env_mask: (n,) for up to 64 interned digests, or (n, env_words)."""
        keep = {}
        soa = ServantSoA()
        for k, dt in (("version", np.uint32), ("num_processors", np.uint32),
                      ("current_load", np.uint32), ("max_tasks", np.uint32),
                      ("running_tasks", np.uint32), ("flags", np.uint32),
                      ("env_mask", np.uint64), ("ip_id", np.uint32)):
            keep[k] = np.ascontiguousarray(cols[k], dtype=dt)
            setattr(soa, k, keep[k].ctypes.data)
        n = len(keep["version"])
        soa.env_words = keep["env_mask"].shape[1] if keep["env_mask"].ndim == 2 else 1
        self._check(lib().ydc_upload_servants(self._h, C.byref(soa), n), "ydc_upload_servants")
        self.n_servants = n

    def update_servants(self, idx, rows, env_masks=None):
        """rows: dicts named like ydc_servant_row. env_masks: optional (len(rows), env_words)
        uint64 array for registries with more than 64 interned digests (rows' env_mask is
        ignored then)."""
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        arr = (ServantRow * len(rows))()
        for i, r in enumerate(rows):
            for k in ("version", "num_processors", "current_load", "max_tasks", "flags", "ip_id",
                      "env_mask"):
                setattr(arr[i], k, int(r.get(k, 0)) if isinstance(r, dict) else int(r[k]))
        if env_masks is None:
            self._check(lib().ydc_update_servants(self._h, idx.ctypes.data, arr, len(rows)),
                        "ydc_update_servants")
        else:
            em = np.ascontiguousarray(env_masks, dtype=np.uint64).reshape(len(rows), -1)
            self._check(lib().ydc_update_servants_wide(self._h, idx.ctypes.data, arr,
                                                       em.ctypes.data, em.shape[1], len(rows)),
                        "ydc_update_servants_wide")
        if len(idx):
            self.n_servants = max(self.n_servants, int(idx.max()) + 1)

    def set_host_aliases(self, ip_id, servant_idx):
        """Further (host id, servant row) entries of the requestor-address lookup table."""
        a = np.ascontiguousarray(ip_id, dtype=np.uint32)
        b = np.ascontiguousarray(servant_idx, dtype=np.uint32)
        assert len(a) == len(b)
        self._check(lib().ydc_set_host_aliases(self._h, a.ctypes.data, b.ctypes.data, len(a)),
                    "ydc_set_host_aliases")

    def remove_servants(self, idx):
        """Servant expiry: rows idx (ascending) leave, the rest keeps its order."""
        a = np.ascontiguousarray(idx, dtype=np.uint32)
        self._check(lib().ydc_remove_servants(self._h, a.ctypes.data, len(a)),
                    "ydc_remove_servants")
        self.n_servants -= len(a)

    def release_slots(self, servant_idx):
        a = np.ascontiguousarray(servant_idx, dtype=np.uint32)
        self._check(lib().ydc_release_slots(self._h, a.ctypes.data, len(a)), "ydc_release_slots")

    def release_slots_device(self, d_servant_idx):
        """Indexes already in device memory (a DeviceArray / tensor, e.g. a batch's placement output):
        entries that are no servant index are skipped; no staging copy."""
        self._check(lib().ydc_release_slots_device(self._h, _ptr(d_servant_idx), int(d_servant_idx.numel())),
                    "ydc_release_slots_device")

    def set_running(self, running):
        a = np.ascontiguousarray(running, dtype=np.uint32)
        self._check(lib().ydc_set_running(self._h, a.ctypes.data, len(a)), "ydc_set_running")

    def get_running(self):
        out = np.empty(self.n_servants, np.uint32)
        self._check(lib().ydc_get_running(self._h, out.ctypes.data, len(out)), "ydc_get_running")
        return out

    def dispatch(self, tasks, commit=False, want_util=True, want_running=True, out_idx=None):
        """Host numpy columns in, numpy out: (servant_idx, utilization|None, running_after|None).
        out_idx: a uint32 array of the batch's length to write the placement into (a caller that
        dispatches batch after batch reuses its buffer; a fresh 400 KB array per call is page
        faults, not dispatch)."""
        keep = [np.ascontiguousarray(tasks[k], dtype=np.uint32)
                for k in ("env_id", "min_version", "requestor_ip")]
        n = len(keep[0])
        soa = TaskSoA(*[a.ctypes.data for a in keep])
        out = out_idx if out_idx is not None else np.empty(n, np.uint32)
        assert out.dtype == np.uint32 and out.size == n and out.flags.c_contiguous
        util = np.empty(n, np.float64) if want_util else None
        run = np.empty(self.n_servants, np.uint32) if want_running else None
        self._check(lib().ydc_dispatch(self._h, C.byref(soa), n, DISPATCH_COMMIT if commit else 0,
                                       _ptr(out), _ptr(util), _ptr(run)), "ydc_dispatch")
        return out, util, run

    def dispatch_tick(self, tasks, upd_idx=(), upd_rows=None, release_idx=(), env_masks=None,
                      commit=True, want_util=False):
        """One scheduler turn (ydc_dispatch_tick): heartbeat rows (structured array of ROW_DTYPE),
        released grants (servant index each), then the requests. Returns (servant_idx, util|None)."""
        ui = np.ascontiguousarray(upd_idx, dtype=np.uint32)
        if len(ui):
            self.n_servants = max(self.n_servants, int(ui.max()) + 1)
        ur = np.ascontiguousarray(upd_rows if upd_rows is not None else np.zeros(0, ROW_DTYPE),
                                  dtype=ROW_DTYPE)
        assert len(ur) == len(ui)
        rel = np.ascontiguousarray(release_idx, dtype=np.uint32)
        keep = [np.ascontiguousarray(tasks[k], dtype=np.uint32)
                for k in ("env_id", "min_version", "requestor_ip")]
        n = len(keep[0])
        soa = TaskSoA(*[a.ctypes.data for a in keep])
        out = np.empty(n, np.uint32)
        util = np.empty(n, np.float64) if want_util else None
        em = None
        if env_masks is not None:
            em = np.ascontiguousarray(env_masks, dtype=np.uint64)
            em = em.reshape(len(ui), em.shape[-1] if em.ndim == 2 else max(1, em.size // max(1, len(ui))))
        self._check(lib().ydc_dispatch_tick(self._h, ui.ctypes.data, ur.ctypes.data,
                                            em.ctypes.data if em is not None else None,
                                            em.shape[1] if em is not None else 1, len(ui),
                                            rel.ctypes.data, len(rel), C.byref(soa), n,
                                            DISPATCH_COMMIT if commit else 0, out.ctypes.data,
                                            _ptr(util)), "ydc_dispatch_tick")
        return out, util

    def dispatch_device(self, d_env, d_minv, d_ip, d_out_idx=None, d_out_util=None,
                        d_out_running=None, commit=False):
        """Device buffers (torch tensors): task columns and outputs already in HBM."""
        soa = TaskSoA(_ptr(d_env), _ptr(d_minv), _ptr(d_ip))
        n = int(d_env.numel())
        self._check(lib().ydc_dispatch_device(self._h, C.byref(soa), n,
                                              DISPATCH_COMMIT if commit else 0, _ptr(d_out_idx),
                                              _ptr(d_out_util), _ptr(d_out_running)),
                    "ydc_dispatch_device")

    def dispatch_device_async(self, d_env, d_minv, d_ip, d_out_idx=None, d_out_util=None,
                              d_out_running=None, commit=False):
        """Pipelined dispatch_device: enqueues the batch and returns; dispatch_wait() waits for
        the oldest outstanding batch (at most two may be outstanding)."""
        soa = TaskSoA(_ptr(d_env), _ptr(d_minv), _ptr(d_ip))
        n = int(d_env.numel())
        self._check(lib().ydc_dispatch_device_async(self._h, C.byref(soa), n,
                                                    DISPATCH_COMMIT if commit else 0,
                                                    _ptr(d_out_idx), _ptr(d_out_util),
                                                    _ptr(d_out_running)),
                    "ydc_dispatch_device_async")

    def dispatch_wait(self):
        self._check(lib().ydc_dispatch_wait(self._h), "ydc_dispatch_wait")

    def synchronize(self):
        self._check(lib().ydc_synchronize(self._h), "ydc_synchronize")

    # -- multi-GPU group ---------------------------------------------------------------
    def group_init(self, unique_id, rank, n_ranks):
        """Collective (like ncclCommInitRank). unique_id: the 128 bytes of group_unique_id()."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(lib().ydc_group_init(self._h, buf, rank, n_ranks), "ydc_group_init")

    def group_size(self):
        """(ranks, is_rccl): for an RCCL group the count is ncclCommCount of the communicator."""
        n, r = C.c_int(0), C.c_int(0)
        self._check(lib().ydc_group_size(self._h, C.byref(n), C.byref(r)), "ydc_group_size")
        return n.value, bool(r.value)

    def group_ipc_export(self, rank, n_ranks):
        """This rank's mailbox handle (IPC_HANDLE_BYTES bytes) for the RCCL-free inter-process
        transport; the launcher all-gathers the handles and every rank calls group_init_ipc."""
        buf = C.create_string_buffer(IPC_HANDLE_BYTES)
        self._check(lib().ydc_group_ipc_export(self._h, rank, n_ranks, buf), "ydc_group_ipc_export")
        return buf.raw

    def group_init_ipc(self, handles, rank, n_ranks, transport=TRANSPORT_IPC_DEVICE):
        """handles: the n_ranks exported handles in rank order. Every rank passes the same
        transport (TRANSPORT_IPC_DEVICE: HIP IPC device memory; TRANSPORT_IPC_HOST: shared host
        segment). Raises if a peer's mailbox cannot be mapped — the export stays, so the
        launcher may agree on the other flavour and call again."""
        blob = b"".join(bytes(h) for h in handles)
        assert len(blob) == n_ranks * IPC_HANDLE_BYTES
        buf = C.create_string_buffer(blob, len(blob))
        self._check(lib().ydc_group_init_ipc(self._h, buf, rank, n_ranks, transport),
                    "ydc_group_init_ipc")

    def group_transport(self):
        """TRANSPORT_* of the group this context is a rank of."""
        return int(lib().ydc_group_transport(self._h))

    def group_destroy(self):
        self._check(lib().ydc_group_destroy(self._h), "ydc_group_destroy")

    def dispatch_sharded(self, d_env, d_minv, d_ip, d_out_idx=None, d_out_util=None,
                         d_out_running=None, commit=False):
        """Collective: this rank's slice of the global batch (device buffers)."""
        soa = TaskSoA(_ptr(d_env), _ptr(d_minv), _ptr(d_ip))
        n = int(d_env.numel())
        self._check(lib().ydc_dispatch_sharded(self._h, C.byref(soa), n,
                                               DISPATCH_COMMIT if commit else 0, _ptr(d_out_idx),
                                               _ptr(d_out_util), _ptr(d_out_running)),
                    "ydc_dispatch_sharded")

    # -- streaming mode: one captured step per tick ---------------------------------
    def stream_begin(self, max_updates, max_releases, max_tasks):
        self._check(lib().ydc_stream_begin(self._h, max_updates, max_releases, max_tasks),
                    "ydc_stream_begin")
        self._stream_caps = (int(max_updates), int(max_releases), int(max_tasks))

    def stream_tick(self, upd_idx, upd_rows, release_idx, tasks, env_masks=None):
        """upd_rows: numpy structured array of ROW_DTYPE (one heartbeat per entry of upd_idx);
        release_idx: servant index of every freed grant; tasks: dict of request columns.
        env_masks: optional (len(upd_idx), env_words) uint64 array — the heartbeats' environment
        sets on a registry with more than 64 digests (ydc_stream_tick_wide).
        Returns the servant index (or IDX_*) of every request."""
        ui = np.ascontiguousarray(upd_idx, dtype=np.uint32)
        if len(ui):
            self.n_servants = max(self.n_servants, int(ui.max()) + 1)
        ur = np.ascontiguousarray(upd_rows, dtype=ROW_DTYPE)
        rel = np.ascontiguousarray(release_idx, dtype=np.uint32)
        keep = [np.ascontiguousarray(tasks[k], dtype=np.uint32)
                for k in ("env_id", "min_version", "requestor_ip")]
        n = len(keep[0])
        soa = TaskSoA(*[a.ctypes.data for a in keep])
        out = np.empty(n, np.uint32)
        if env_masks is None:
            self._check(lib().ydc_stream_tick(self._h, ui.ctypes.data, ur.ctypes.data, len(ui),
                                              rel.ctypes.data, len(rel), C.byref(soa), n,
                                              out.ctypes.data), "ydc_stream_tick")
        else:
            em = np.ascontiguousarray(env_masks, dtype=np.uint64).reshape(len(ui), -1)
            self._check(lib().ydc_stream_tick_wide(self._h, ui.ctypes.data, ur.ctypes.data,
                                                   em.ctypes.data, em.shape[1], len(ui),
                                                   rel.ctypes.data, len(rel), C.byref(soa), n,
                                                   out.ctypes.data), "ydc_stream_tick_wide")
        return out

    def stream_buffers(self, max_updates=None, max_releases=None, max_tasks=None):
        """numpy views of the page-locked arrays a tick is staged in (ydc_stream_buffers_get): fill
        them in place and call stream_tick_inplace — nothing is copied on either side. The views
        have the capacities stream_begin was given (arguments, if any, may only ask for less)."""
        caps = self._stream_caps
        want = (max_updates, max_releases, max_tasks)
        if any(w is not None and w > c for w, c in zip(want, caps)):
            raise YdcError("stream_buffers%r: beyond the capacities of stream_begin%r" % (want, caps))
        max_updates, max_releases, max_tasks = [c if w is None else w for w, c in zip(want, caps)]
        class _B(C.Structure):
            _fields_ = [(k, C.c_void_p) for k in ("upd_idx", "upd_rows", "release_servant_idx", "env_id",
                                                   "min_version", "requestor_ip", "out_servant_idx")]
        b = _B()
        self._check(lib().ydc_stream_buffers_get(self._h, C.byref(b)), "ydc_stream_buffers_get")

        def view(p, n, dt):
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(p)
            return np.frombuffer(buf, dtype=dt, count=n)
        self._stream_views = {
            "upd_idx": view(b.upd_idx, max_updates, np.uint32), "upd_rows": view(b.upd_rows, max_updates, ROW_DTYPE),
            "release_idx": view(b.release_servant_idx, max_releases, np.uint32),
            "env_id": view(b.env_id, max_tasks, np.uint32), "min_version": view(b.min_version, max_tasks, np.uint32),
            "requestor_ip": view(b.requestor_ip, max_tasks, np.uint32), "out": view(b.out_servant_idx, max_tasks, np.uint32)}
        return self._stream_views

    def stream_tick_inplace(self, n_upd, n_rel, n_tasks):
        """One tick whose data already lies in stream_buffers(); the answers are in views["out"][:n_tasks]."""
        v = self._stream_views
        soa = TaskSoA(v["env_id"].ctypes.data, v["min_version"].ctypes.data, v["requestor_ip"].ctypes.data)
        self._check(lib().ydc_stream_tick(self._h, v["upd_idx"].ctypes.data, v["upd_rows"].ctypes.data, n_upd,
                                          v["release_idx"].ctypes.data, n_rel, C.byref(soa), n_tasks,
                                          v["out"].ctypes.data), "ydc_stream_tick")
        return v["out"][:n_tasks]

    def stream_end(self):
        self._check(lib().ydc_stream_end(self._h), "ydc_stream_end")

    def set_profiling(self, on):
        self._check(lib().ydc_set_profiling(self._h, int(on)), "ydc_set_profiling")

    def kernel_profile(self):
        """{"kernel": [launches, total_ms]} of the last dispatch (profiling must be on)."""
        import json
        return json.loads(lib().ydc_kernel_profile(self._h).decode() or "{}")

    def stats(self):
        st = Stats()
        self._check(lib().ydc_get_stats(self._h, C.byref(st)), "ydc_get_stats")
        return st.as_dict()
