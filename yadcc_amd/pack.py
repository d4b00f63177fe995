"""Column packing between "raw" servant personalities and the C-ABI layout.

Raw columns (yadcc_amd.synth, oracle) keep priority and the two memory fields
of ServantPersonality (reference yadcc/scheduler/task_dispatcher.h:80-116);
include/yadcc_dispatch.h folds them into `flags` the way
TaskDispatcher::GetCapacityAvailable / UnsafeTryPickDedicatedServantFor read
them (task_dispatcher.cc:286-287,405).
"""
import numpy as np

SERVANT_DEDICATED = 1
SERVANT_LOW_MEMORY = 2
PRIORITY_DEDICATED = 1
MIN_MEMORY_DEFAULT = 10 << 30  # FLAGS_servant_min_memory_for_accepting_new_task = "10G"


def parse_size(s):
    """TryParseSize (reference yadcc/common/parse_size.cc:25-45)."""
    if not s:
        return None
    scale = {"G": 1 << 30, "M": 1 << 20, "K": 1 << 10, "B": 1}.get(s[-1])
    body = s[:-1] if scale else s
    if not body.isdigit() or not body.isascii():
        return None
    return int(body) * (scale or 1)


def servant_flags(sv, min_memory=MIN_MEMORY_DEFAULT):
    ded = np.asarray(sv["priority"]) == PRIORITY_DEDICATED
    low = (np.asarray(sv["total_memory"]) != 0) & (
        np.asarray(sv["memory_available"]).astype(np.uint64) < np.uint64(min_memory))
    return (ded * SERVANT_DEDICATED + low * SERVANT_LOW_MEMORY).astype(np.uint32)


def to_abi_columns(sv, min_memory=MIN_MEMORY_DEFAULT):
    """dict of contiguous numpy columns named like ydc_servant_soa's members."""
    c = lambda k, dt: np.ascontiguousarray(sv[k], dtype=dt)
    return {
        "version": c("version", np.uint32),
        "num_processors": c("num_processors", np.uint32),
        "current_load": c("current_load", np.uint32),
        "max_tasks": c("max_tasks", np.uint32),
        "running_tasks": c("running_tasks", np.uint32),
        "flags": servant_flags(sv, min_memory),
        "env_mask": c("env_mask", np.uint64),  # (n,) or (n, env_words)
        "ip_id": c("ip", np.uint32),
    }
