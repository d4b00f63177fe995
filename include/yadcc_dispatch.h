/* yadcc_dispatch.h — C-ABI of the MI355X task-dispatch path.
 *
 * Drop-in boundary for the placement arithmetic of Tencent/yadcc's scheduler:
 * what TaskDispatcher::WaitForStartingNewTask
 * (reference yadcc/scheduler/task_dispatcher.cc:93-140, helpers :283-451)
 * decides for ONE request, decided here for a whole batch of pending requests
 * against the resident servant table, with results identical to issuing the
 * requests one after another in array order (timeout == now, no heartbeat,
 * timer or free in between).
 *
 * Plain pointers and sizes only; no C++ or torch types. Every function returns
 * YDC_OK (0) or a negative error code and never throws. A context owns one HIP
 * stream (or borrows the caller's) and is not re-entrant.
 *
 * The host-side mirror of the reference class (same six public methods as
 * task_dispatcher.h:139-181) is yadcc_amd/csrc/gpu_task_dispatcher.h; its C
 * wrapper is declared at the bottom of this file (ydc_td_*).
 */
#ifndef YADCC_DISPATCH_H_
#define YADCC_DISPATCH_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- result sentinels (out_servant_idx) ---------------------------------- */
#define YDC_IDX_TIMEOUT 0xFFFFFFFFu       /* WaitStatus::Timeout,             task_dispatcher.h:43 */
#define YDC_IDX_ENV_NOT_FOUND 0xFFFFFFFEu /* WaitStatus::EnvironmentNotFound, task_dispatcher.h:42 */

/* ---- error codes ---------------------------------------------------------- */
#define YDC_OK 0
#define YDC_ERR_INVALID_ARGUMENT (-1)
#define YDC_ERR_HIP (-2)             /* a HIP runtime call failed; see ydc_last_error() */
#define YDC_ERR_NO_DEVICE (-3)       /* no usable gfx950 device: there is NO CPU fallback */
#define YDC_ERR_CAPACITY (-4)        /* more servants/tasks/slots than the context was created for */
#define YDC_ERR_TOO_MANY_CLASSES (-5)/* more (env set, version) signatures than the entry point takes */
#define YDC_ERR_NOT_CONVERGED (-6)   /* internal invariant broken (never expected) */

/* Servant classes = distinct (env set, version) signatures among servants with max_tasks != 0.
 * Every entry point takes up to 65535. Up to 256 the lane-per-class kernel places the batch; above
 * that a wave-per-chunk kernel (up to 3072 classes) or a thread-per-chunk kernel does, a streaming
 * tick is enqueued instead of replayed from its captured graph, and the ranks of a group place
 * the whole batch redundantly (same results, no speed-up). */
#define YDC_MAX_CLASSES 65535u
#define YDC_MAX_FAST_CLASSES 256u
/* Interned compiler digests per 64-bit word of an environment mask. The number of words per
 * servant (env_words) is the caller's choice: the reference keeps an unbounded
 * std::vector<EnvironmentDesc> per servant (task_dispatcher.h:93-94, .cc:55-63), so there
 * is no limit on the number of distinct live digests here either. */
#define YDC_ENVS_PER_WORD 64u
#define YDC_MAX_ENV_WORDS 1024u

/* ---- servant flags --------------------------------------------------------- */
/* ServantPersonality::priority == SERVANT_PRIORITY_DEDICATED (task_dispatcher.cc:405) */
#define YDC_SERVANT_DEDICATED 1u
/* total_memory_in_bytes != 0 && memory_available_in_bytes < min_memory_for_new_task_
 * (task_dispatcher.cc:286-287), folded by the host packer. */
#define YDC_SERVANT_LOW_MEMORY 2u

/* Servant registry columns, registration order == array order
 * (ServantPersonality + ServantDesc, task_dispatcher.h:80-116,184-193).
 * All widths are the wire widths (api/scheduler.proto:76-97). */
typedef struct ydc_servant_soa {
  const uint32_t* version;        /* compared as unsigned, task_dispatcher.cc:333 */
  const uint32_t* num_processors;
  const uint32_t* current_load;
  const uint32_t* max_tasks;      /* 0 => never eligible, task_dispatcher.cc:330-332 */
  const uint32_t* running_tasks;  /* ServantDesc::running_tasks */
  const uint32_t* flags;          /* YDC_SERVANT_* */
  const uint64_t* env_mask;       /* env_words words per servant: bit j of word w of servant s
                                     (env_mask[s * env_words + w]) <=> advertises interned
                                     compiler digest 64 * w + j */
  const uint32_t* ip_id;          /* interned text before ':' of observed_location;
                                     equal ids <=> IsNetworkAddressEqual, task_dispatcher.cc:66-69 */
  uint32_t env_words;             /* 64-bit words per servant in env_mask; 0 is read as 1 */
} ydc_servant_soa;

/* One heartbeat's worth of a servant row (KeepServantAlive replaces the
 * personality but keeps running_tasks, task_dispatcher.cc:195-201). */
typedef struct ydc_servant_row {
  uint32_t version, num_processors, current_load, max_tasks, flags, ip_id;
  uint64_t env_mask; /* word 0 of the servant's mask (ydc_update_servants: the whole mask) */
} ydc_servant_row;

/* Pending requests in arrival order (TaskPersonality, task_dispatcher.h:48-66). */
typedef struct ydc_task_soa {
  const uint32_t* env_id;       /* interned compiler digest (bit number in the servants' masks);
                                   >= 64 * env_words of the resident table: nobody has it */
  const uint32_t* min_version;
  const uint32_t* requestor_ip; /* same interning as ydc_servant_soa::ip_id */
} ydc_task_soa;

typedef struct ydc_context ydc_context;

/* Counters of the most recent dispatch (debugging / bench). */
typedef struct ydc_stats {
  uint32_t n_tasks, n_servants, n_classes;
  uint32_t n_slots;        /* free (servant, running) slots generated */
  uint32_t key_bits;       /* significant bits of the slot sort key */
  uint32_t radix_passes;   /* key passes of the radix sort; 0: the bin sort ordered the slots */
  uint32_t n_chunks;       /* task chunks simulated in parallel */
  uint32_t rounds;         /* speculation rounds until the chunk states were consistent */
  uint32_t chunk_sims;     /* chunk simulations executed over all rounds */
  uint32_t granted, timeouts, env_not_found;
  /* multi-GPU group, cumulative: batches that ran with a sharded sort (each rank generated and
   * sorted only its key window) and how many of them had to be repeated with the full sort
   * because a window turned out too small. */
  uint32_t shard_sort_batches, shard_sort_misses;
  uint32_t small_batch;    /* 1: the one-launch path placed the batch (ydc_dispatch_tick) */
  uint32_t zone_rows;      /* chunks around the dedicated tier's end that started from a walked state (0: no walk) */
  /* cumulative over the context's life: calls answered by the resident tick kernel (no launch),
   * by a launched tick kernel, and batches placed by the batch pipeline */
  uint32_t tick_resident_calls, tick_launched_calls, pipeline_batches;
  float stage_ms[16];      /* per-stage GPU time when profiling is on (ydc_set_profiling) */
} ydc_stats;

/* stage indices of ydc_stats::stage_ms */
enum {
  YDC_STAGE_SERVANT_SCAN = 0, YDC_STAGE_SLOT_GEN, YDC_STAGE_SORT, YDC_STAGE_CLASS_LISTS,
  YDC_STAGE_TASK_CLASSIFY, YDC_STAGE_MATCH, YDC_STAGE_FINALIZE, YDC_STAGE_TOTAL, YDC_STAGE_COUNT
};

/* dispatch flags */
#define YDC_DISPATCH_COMMIT 1u /* add the grants to the resident running_tasks, like
                                  `++pick->running_tasks` (task_dispatcher.cc:123) */

const char* ydc_strerror(int code);
const char* ydc_last_error(const ydc_context* ctx); /* ctx == NULL: last error outside a context */
uint32_t ydc_abi_version(void);

/* Number of usable devices (0 if the HIP runtime cannot see one). */
int ydc_device_count(void);
/* Plain device buffers for callers that keep request columns / results in HBM
 * (bench, streaming): thin wrappers of hipMalloc/hipFree/hipMemcpy. */
int ydc_device_malloc(int device, size_t bytes, void** out);
int ydc_device_free(void* p);
int ydc_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes);
int ydc_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes);

/* stream: a hipStream_t to launch on, or NULL to create a private one. */
int ydc_create(int device, uint32_t max_servants, uint32_t max_tasks, uint32_t max_slots,
               void* stream, ydc_context** out);
int ydc_destroy(ydc_context* ctx);

/* Replace the whole resident servant table (host columns). */
int ydc_upload_servants(ydc_context* ctx, const ydc_servant_soa* servants, uint32_t n);
/* Heartbeats: overwrite rows idx[i] (idx[i] == current count appends a new servant
 * with running_tasks = 0, task_dispatcher.cc:205-210). */
int ydc_update_servants(ydc_context* ctx, const uint32_t* idx, const ydc_servant_row* rows,
                        uint32_t n);
/* Same for registries with more than 64 interned digests: the masks come in env_masks
 * (env_words words per row, env_masks[i * env_words + w]) and rows[i].env_mask is ignored.
 * A table that was uploaded with fewer words is widened (the missing words are zero). */
int ydc_update_servants_wide(ydc_context* ctx, const uint32_t* idx, const ydc_servant_row* rows,
                             const uint64_t* env_masks, uint32_t env_words, uint32_t n);
/* A servant answers to EVERY requestor address that is a prefix of its observed_location ending
 * right before a ':' (IsNetworkAddressEqual, task_dispatcher.cc:66-69) — "[::1]:8335" to "[::1]",
 * but also to "[:" and "[". ip_id carries one of them; the others are given here as further
 * (host id, servant row) pairs of the lookup table (replaces the previous list; n == 0 clears).
 * ydc_upload_servants and ydc_remove_servants drop the list (rows change), heartbeats keep it. */
int ydc_set_host_aliases(ydc_context* ctx, const uint32_t* ip_id, const uint32_t* servant_idx,
                         uint32_t n);
/* OnExpirationTimer's erase (task_dispatcher.cc:503-516): removes the rows idx[0..n) (strictly
 * ascending) from the resident table; the servants behind them move up, keeping their order
 * (registry order decides ties) and their running_tasks. Done on the device: no table upload. */
int ydc_remove_servants(ydc_context* ctx, const uint32_t* idx, uint32_t n);
/* FreeTask / zombie / orphan sweeps: running_tasks[servant_idx[i]] -= 1
 * (task_dispatcher.cc:181). */
int ydc_release_slots(ydc_context* ctx, const uint32_t* servant_idx, uint32_t n);
/* The same with the indexes in device memory (or page-locked host memory the device can address) —
 * e.g. the placement array of an earlier batch as it is: entries that are no servant index
 * (YDC_IDX_*) are skipped. No staging copy; asynchronous on the context's stream. */
int ydc_release_slots_device(ydc_context* ctx, const uint32_t* d_servant_idx, uint32_t n);
/* Overwrite / read back the resident running_tasks column. */
int ydc_set_running(ydc_context* ctx, const uint32_t* running, uint32_t n);
int ydc_get_running(ydc_context* ctx, uint32_t* out_running, uint32_t n);

/* Batch dispatch, host buffers, synchronous.
 * out_servant_idx[n_tasks]: registry index or YDC_IDX_*.
 * out_utilization (nullable) [n_tasks]: chosen servant's double(running)/capacity
 *   at pick time (task_dispatcher.cc:440-441), -1.0 if not granted.
 * out_running (nullable) [n_servants]: running_tasks after the batch. */
int ydc_dispatch(ydc_context* ctx, const ydc_task_soa* tasks, uint32_t n_tasks, uint32_t flags,
                 uint32_t* out_servant_idx, double* out_utilization, uint32_t* out_running);
/* (ydc_dispatch and ydc_dispatch_device send batches of up to 64 requests through the
 * one-launch path of ydc_dispatch_tick as well.) */

/* One scheduler turn for the reference's real call shape — a WaitForStartingTask RPC asks for
 * `waiters + 1` grants (daemon/local/task_grant_keeper.cc:145-146; the handler's loop,
 * scheduler_service_impl.cc:234-264): a handful of requests, plus whatever reached the registry
 * since the last turn. Applies, in this order, n_upd heartbeats (KeepServantAlive,
 * task_dispatcher.cc:195-201; idx[i] == current count appends a servant, as ydc_update_servants),
 * n_rel released grants (FreeTask's --running_tasks, :181) and places n_tasks requests
 * (n x WaitForStartingNewTask with timeout == now, :93-140). Same answers as
 * ydc_update_servants_wide + ydc_release_slots + ydc_dispatch — which is what it does when the
 * batch is large (more than 64 requests; 48 / 32 on registries beyond 4096 / 8192 servants), a
 * heartbeat changes structure (a new servant, other environments / version / host / capacity
 * bound) or the registry is beyond 16384 servants / 4096 classes. Otherwise ONE workgroup: it
 * reads the registry once into registers, applies the deltas, and makes the picks one after
 * another (the reference's own arg-min, :362-451, as a workgroup-wide min-reduction per pick;
 * the identical requests of one RPC as one merge); requests, deltas and results travel as kernel
 * arguments and stores to page-locked memory — no sort, no copy command, one wait. With
 * YDC_DISPATCH_COMMIT the kernel then STAYS on its CU and takes the following calls from a
 * page-locked mailbox (no launch, no column loads: a call is two PCIe round trips) until another
 * entry point of the context needs the registry or nobody has called for 50 ms.
 * upd_env_masks (nullable): env_words words per heartbeat row, as ydc_update_servants_wide.
 * Host buffers in and out, synchronous. out_utilization is nullable. */
int ydc_dispatch_tick(ydc_context* ctx, const uint32_t* upd_idx, const ydc_servant_row* upd_rows,
                      const uint64_t* upd_env_masks, uint32_t env_words, uint32_t n_upd,
                      const uint32_t* release_servant_idx, uint32_t n_rel,
                      const ydc_task_soa* tasks, uint32_t n_tasks, uint32_t flags,
                      uint32_t* out_servant_idx, double* out_utilization);

/* Page-locked host memory for ydc_dispatch: request columns and result arrays that lie in a
 * range registered here (or allocated here, or pinned by the caller's own hipHostMalloc /
 * hipHostRegister) are handed to the kernels as they are — the classification reads the
 * columns and the final kernel writes the results through the range's device address; no
 * staging copy and no copy command. A scheduler registers the buffers it reuses from batch to
 * batch once. Process-wide (not per context). */
int ydc_host_register(void* p, size_t bytes);
int ydc_host_unregister(void* p);
int ydc_host_alloc(size_t bytes, void** out);
int ydc_host_free(void* p);

/* Same with DEVICE pointers (task columns and outputs already in HBM);
 * asynchronous on the context stream except for one 16-byte convergence
 * read-back. out_* may be NULL. */
int ydc_dispatch_device(ydc_context* ctx, const ydc_task_soa* d_tasks, uint32_t n_tasks,
                        uint32_t flags, uint32_t* d_out_servant_idx, double* d_out_utilization,
                        uint32_t* d_out_running);

/* Pipelined form: ydc_dispatch_device_async enqueues the whole batch and returns without
 * waiting; ydc_dispatch_wait waits for the OLDEST outstanding batch, whose results are final
 * when it returns. At most two batches may be outstanding, so the usual loop is
 *   async(0); for k = 1..: async(k); wait();  ...  wait();
 * — the device works on batch k while the host looks at the outcome of batch k - 1 and enqueues
 * batch k + 1. Batches take effect strictly in order (COMMIT of batch k is what batch k + 1
 * sees), and results are exactly those of the synchronous calls: a batch that does not become
 * final within the matching passes enqueued for it takes no effect on the device, neither does
 * the batch behind it, and ydc_dispatch_wait places both again, in order. The caller's buffers
 * of a batch (request columns, outputs) must stay untouched until its wait has returned; no
 * other call on the context is allowed while batches are outstanding. */
int ydc_dispatch_device_async(ydc_context* ctx, const ydc_task_soa* d_tasks, uint32_t n_tasks,
                              uint32_t flags, uint32_t* d_out_servant_idx,
                              double* d_out_utilization, uint32_t* d_out_running);
int ydc_dispatch_wait(ydc_context* ctx);

/* ---- streaming mode (BASELINE.json configs[4]) -------------------------------
 * One tick applies, in this order: n_upd heartbeats of known servants
 * (KeepServantAlive: personality replaced, running_tasks kept, task_dispatcher.cc:195-201),
 * n_rel released grants (FreeTask's --running_tasks, :181) and n_tasks requests that are
 * dispatched and committed (n x WaitForStartingNewTask with timeout == now). The whole
 * step is captured into a hipGraph once and replayed per tick; counts may vary up to the
 * capacities given here. Host buffers in, host results out, synchronous.
 * Heartbeats that add a servant or change its environments / version / host / capacity
 * bound are applied eagerly (ydc_update_servants) and the step is captured again. With
 * env_words > 1 a tick's rows cannot carry a mask: upd_rows[i].env_mask is ignored and a known
 * servant keeps its environments — ydc_stream_tick_wide carries the masks. */
int ydc_stream_begin(ydc_context* ctx, uint32_t max_updates, uint32_t max_releases,
                     uint32_t max_tasks);
int ydc_stream_tick(ydc_context* ctx, const uint32_t* upd_idx, const ydc_servant_row* upd_rows,
                    uint32_t n_upd, const uint32_t* release_servant_idx, uint32_t n_rel,
                    const ydc_task_soa* tasks, uint32_t n_tasks, uint32_t* out_servant_idx);
/* The same tick for registries with more than 64 interned digests: upd_env_masks holds env_words
 * words per heartbeat row (upd_rows[i].env_mask is ignored), so a heartbeat may change what a
 * servant advertises — or add a servant — inside a tick (applied eagerly; the step is captured
 * again). ydc_stream_tick on such a table refuses a tick that adds a servant. */
int ydc_stream_tick_wide(ydc_context* ctx, const uint32_t* upd_idx, const ydc_servant_row* upd_rows,
                         const uint64_t* upd_env_masks, uint32_t env_words, uint32_t n_upd,
                         const uint32_t* release_servant_idx, uint32_t n_rel,
                         const ydc_task_soa* tasks, uint32_t n_tasks, uint32_t* out_servant_idx);
/* The page-locked arrays a tick is staged in (capacities as given to ydc_stream_begin; valid until
 * ydc_stream_end). A caller that assembles its tick right there and passes these very pointers to
 * ydc_stream_tick / _wide (requests as a ydc_task_soa of env_id / min_version / requestor_ip,
 * out_servant_idx for the answers) is not copied on either side: the captured step reads and
 * writes them in place. */
typedef struct ydc_stream_buffers {
  uint32_t* upd_idx;
  ydc_servant_row* upd_rows;
  uint32_t* release_servant_idx;
  uint32_t *env_id, *min_version, *requestor_ip;
  uint32_t* out_servant_idx;
} ydc_stream_buffers;
int ydc_stream_buffers_get(ydc_context* ctx, ydc_stream_buffers* out);
int ydc_stream_end(ydc_context* ctx);

/* ---- multi-GPU group: one batch sharded by rank range (BASELINE.json configs[3]) ------
 * One process per GPU; every rank creates its context and uploads the SAME servant table.
 * Rank 0 gets a 128-byte id (ncclGetUniqueId), the launcher hands it to every rank (any
 * side channel), every rank calls ydc_group_init — collective, like ncclCommInitRank.
 * librccl.so.1 is resolved with dlopen here, so single-GPU use has no RCCL dependency. */
#define YDC_GROUP_ID_BYTES 128
int ydc_group_unique_id(void* out_id128);
int ydc_group_init(ydc_context* ctx, const void* id128, int rank, int n_ranks);
/* Several contexts of ONE process on ONE device as the ranks of a group, exchanging through
 * device copies instead of RCCL (each rank's calls must come from its own thread): how the
 * sharding protocol is exercised on a single-GPU machine. */
int ydc_group_init_local(ydc_context** ctxs, int n);
/* The same protocol without RCCL, between processes of one node (one process per GPU, or several
 * processes sharing a GPU — RCCL refuses two ranks on one device): every rank owns a mailbox its
 * peers write into and its own kernels poll, in-stream, with a bounded wait. Every rank calls
 * ydc_group_ipc_export (allocates the mailbox, fills YDC_IPC_HANDLE_BYTES of out_handle), the
 * launcher all-gathers the handles over any side channel, every rank calls ydc_group_init_ipc with
 * all n_ranks handles in rank order and the SAME transport:
 *   YDC_TRANSPORT_IPC_DEVICE  mailboxes in device memory, opened through HIP IPC handles — peer
 *                             writes travel over xGMI (or stay on the device the ranks share);
 *   YDC_TRANSPORT_IPC_HOST    mailboxes in a shared host segment mapped by every rank (PCIe): for
 *                             boxes where HIP IPC is not available.
 * A failed init leaves the export in place, so the launcher can agree on the other flavour and
 * call ydc_group_init_ipc again. RCCL stays the default transport (ydc_group_init). */
#define YDC_IPC_HANDLE_BYTES 256
#define YDC_TRANSPORT_NONE 0
#define YDC_TRANSPORT_RCCL 1
#define YDC_TRANSPORT_LOCAL 2
#define YDC_TRANSPORT_IPC_DEVICE 3
#define YDC_TRANSPORT_IPC_HOST 4
int ydc_group_ipc_export(ydc_context* ctx, int rank, int n_ranks, void* out_handle);
int ydc_group_init_ipc(ydc_context* ctx, const void* handles, int rank, int n_ranks, int transport);
/* YDC_TRANSPORT_* of the group this context belongs to. */
int ydc_group_transport(ydc_context* ctx);
int ydc_group_destroy(ydc_context* ctx);
/* Ranks of the group this context belongs to (0: none). For an RCCL group the number comes
 * from the communicator (ncclCommCount) and *out_is_rccl (nullable) is 1. */
int ydc_group_size(ydc_context* ctx, int* out_ranks, int* out_is_rccl);
/* Collective. The global batch is the concatenation, in rank order, of the slices the ranks
 * pass in (device pointers; a slice may be empty); placement is identical to
 * ydc_dispatch_device of the whole batch on one GPU. d_out_servant_idx / d_out_utilization
 * cover the rank's own slice, d_out_running (nullable, n_servants entries) is the global
 * running_tasks after the batch, identical on all ranks; YDC_DISPATCH_COMMIT applies it.
 * Exchanges: all-gathers of 4 B, (n_classes + 1) * 16 B per matching pass, n_servants * 4 B
 * (the per-rank servant-slot deltas) per rank, and — when the slot sort is sharded as well (each
 * rank generates and sorts only the key window its rank range can reach; integer keys, one
 * independent part, one servant per host) — n_classes * 8 B (the windows). Registries with more
 * than 256 servant classes are not sharded: every rank gathers the whole batch and places it
 * redundantly (same results, no speed-up). */
int ydc_dispatch_sharded(ydc_context* ctx, const ydc_task_soa* d_tasks_slice, uint32_t n_slice,
                         uint32_t flags, uint32_t* d_out_servant_idx, double* d_out_utilization,
                         uint32_t* d_out_running);

int ydc_synchronize(ydc_context* ctx);
int ydc_set_profiling(ydc_context* ctx, int on);
int ydc_get_stats(const ydc_context* ctx, ydc_stats* out);
/* Profiling on: per-kernel totals of the most recent dispatch, measured with HIP
 * events on the context stream, as JSON {"kernel": [launches, total_ms], ...}.
 * The string lives until the next dispatch. */
const char* ydc_kernel_profile(const ydc_context* ctx);


/* ===========================================================================
 * ydc_td_* — C wrapper of the host class GpuTaskDispatcher
 * (yadcc_amd/csrc/gpu_task_dispatcher.h), i.e. of the reference's
 * TaskDispatcher public surface, task_dispatcher.h:139-181. One function per
 * method, strings as NUL-terminated char*, durations in nanoseconds.
 * Thread-safe like the reference class (one internal lock; concurrent
 * WaitForStartingNewTask callers are combined into one device batch).
 * =========================================================================== */
typedef struct ydc_td ydc_td;

/* return codes of the wait functions (>= 0); negative: YDC_ERR_* */
#define YDC_TD_GRANTED 0
#define YDC_TD_ENV_NOT_FOUND 1 /* WaitStatus::EnvironmentNotFound -> STATUS_ENVIRONMENT_NOT_AVAILABLE */
#define YDC_TD_TIMEOUT 2       /* WaitStatus::Timeout            -> STATUS_NO_QUOTA_AVAILABLE   */

/* ServantPersonality, task_dispatcher.h:80-116. */
typedef struct ydc_td_servant {
  int32_t version;
  const char* observed_location;
  const char* reported_location;
  const char* const* env_digests; /* EnvironmentDesc::compiler_digest of each environment */
  size_t n_envs;
  uint64_t num_processors, current_load, total_memory_in_bytes, memory_available_in_bytes,
      max_tasks;
  int32_t priority;                  /* ServantPriority, api/scheduler.proto:39-48 */
  int32_t not_accepting_task_reason; /* api/scheduler.proto:51-62 */
} ydc_td_servant;

/* RunningTask, api/scheduler.proto:233-238. */
typedef struct ydc_td_running_task {
  uint64_t servant_task_id, task_grant_id;
  const char* servant_location;
  const char* task_digest;
} ydc_td_running_task;

/* device: HIP ordinal, or -1 for a dispatcher without a device (registry and lease
 * bookkeeping only; every wait fails with YDC_ERR_NO_DEVICE — there is no CPU placement).
 * min_memory: --servant_min_memory_for_accepting_new_task, NULL = "10G" (task_dispatcher.cc:35-38).
 * start_timer: own 1 s expiration thread (task_dispatcher.cc:81-82).
 * fake_clock: time only moves through ydc_td_set_clock_ns (tests). */
int ydc_td_create(int device, const char* min_memory, int start_timer, int fake_clock,
                  ydc_td** out);
int ydc_td_destroy(ydc_td* td);
int ydc_td_device_status(const ydc_td* td); /* YDC_OK or why placement is unavailable */
int ydc_td_set_clock_ns(ydc_td* td, int64_t now_ns);

/* KeepServantAlive, task_dispatcher.h:167-168. */
int ydc_td_keep_servant_alive(ydc_td* td, const ydc_td_servant* servant, int64_t expires_in_ns);
/* WaitForStartingNewTask, task_dispatcher.h:139-141. timeout_in_ns is relative to now
 * (0: do not block). out_location receives TaskAllocation::servant_location; if it does not
 * fit in location_cap bytes the grant is given back and YDC_ERR_CAPACITY returned. */
int ydc_td_wait_for_starting_new_task(ydc_td* td, const char* requestor_ip, uint32_t min_version,
                                      const char* compiler_digest, int64_t expires_in_ns,
                                      int64_t timeout_in_ns, int prefetching,
                                      uint64_t* out_task_id, char* out_location,
                                      size_t location_cap);
/* n back-to-back WaitForStartingNewTask(timeout = now) calls as ONE device batch
 * (the loop of scheduler_service_impl.cc:234-264). out_status[i]: YDC_TD_*;
 * out_locations: n strings of location_stride bytes each (nullable). */
int ydc_td_wait_for_starting_new_tasks(ydc_td* td, size_t n, const char* const* requestor_ips,
                                       const uint32_t* min_versions,
                                       const char* const* compiler_digests, int64_t expires_in_ns,
                                       const uint8_t* prefetching, int32_t* out_status,
                                       uint64_t* out_task_ids, char* out_locations,
                                       size_t location_stride);
/* KeepTaskAlive, task_dispatcher.h:146-147: 1 renewed, 0 unknown or zombie. */
int ydc_td_keep_task_alive(ydc_td* td, uint64_t task_id, int64_t new_expires_in_ns);
/* FreeTask, task_dispatcher.h:154. */
int ydc_td_free_task(ydc_td* td, uint64_t task_id);
/* n FreeTask calls under one lock acquisition (what the handler of a FreeTask RPC carrying
 * several grant ids does in a loop, scheduler_service_impl.cc:307-309). */
int ydc_td_free_tasks(ydc_td* td, const uint64_t* task_ids, size_t n);
/* NotifyServantRunningTasks, task_dispatcher.h:175-176: returns the number of grant ids
 * unknown to the dispatcher; the first min(count, unknown_cap) are written to out_unknown. */
int64_t ydc_td_notify_servant_running_tasks(ydc_td* td, const char* servant_location,
                                            const ydc_td_running_task* tasks, size_t n,
                                            uint64_t* out_unknown, size_t unknown_cap);
/* GetRunningTasks, task_dispatcher.h:180: returns the total count; the first
 * min(count, cap) entries are written (string columns nullable, fixed stride). */
int64_t ydc_td_get_running_tasks(ydc_td* td, uint64_t* out_servant_task_ids,
                                 uint64_t* out_grant_ids, char* out_locations,
                                 size_t location_stride, char* out_digests, size_t digest_stride,
                                 size_t cap);
/* The same list without the copy: a view into the dispatcher's shared, immutable snapshot of it
 * (RunningTaskBookkeeper::GetRunningTasks, running_task_bookkeeper.cc:36-43 — every daemon polls it
 * once a second, daemon/local/running_task_keeper.cc:31-33, while it changes only with a servant's
 * report). Ids as arrays, strings as (offset, length) into one pool, NUL-terminated there. The view
 * stays valid — and unchanged — until the handle is released, whatever reports arrive meanwhile. */
typedef struct ydc_td_running_view {
  size_t n;
  const uint64_t* servant_task_ids;
  const uint64_t* task_grant_ids;
  const uint32_t *location_off, *location_len; /* servant_location of task i: strings + location_off[i] */
  const uint32_t *digest_off, *digest_len;     /* task_digest */
  const char* strings;
} ydc_td_running_view;
int ydc_td_running_tasks_acquire(ydc_td* td, void** out_handle, ydc_td_running_view* out_view);
int ydc_td_running_tasks_release(void* handle);
/* Where the host class spent its time so far (cumulative, nanoseconds of the steady clock):
 * device_ns inside the device API (registry deltas + ydc_dispatch), host_ns in the class itself
 * (string lookups, lease records, results), both over `requests` placed requests in `batches`
 * device batches; heartbeats seen / heartbeats that changed no device column; rebuilds of the
 * flattened GetRunningTasks list (the rest of the calls shared the previous one). */
typedef struct ydc_td_stats {
  uint64_t requests, batches, device_ns, host_ns;
  uint64_t heartbeats, heartbeats_unchanged, bookkeeper_rebuilds;
  uint64_t lease_pages; /* pages of the lease table in use (4096 grant ids each): bounded by the live leases */
  /* OnExpirationTimer (task_dispatcher.cc:498-536): ticks so far; lease entries the last tick looked
   * at (those filed under the seconds that were due — the reference looks at every lease, :523-535);
   * how long the last tick held the dispatcher's lock, and the longest any tick did; entries in
   * the expiry index at the moment (live leases + not yet discarded renewals). */
  uint64_t timer_ticks, timer_lease_entries_seen, timer_last_ns, timer_max_ns, lease_wheel_entries;
} ydc_td_stats;
int ydc_td_host_stats(ydc_td* td, ydc_td_stats* out);
/* OnExpirationTimer, task_dispatcher.cc:498-536 (for hosts that drive the 1 s tick themselves). */
int ydc_td_on_expiration_timer(ydc_td* td);
/* DumpInternals, task_dispatcher.cc:538-614, as JSON. Valid until the next call on td. */
const char* ydc_td_dump_internals(ydc_td* td);
/* Test switch: the order in which calls took effect. With the log on, every call appends one
 * record under the dispatcher's lock at the moment it reads or changes the state (each placement
 * attempt with its answer, each freed id, renewal, heartbeat, servant report, timer tick — with the
 * clock reading it used). ydc_td_oplog_take returns the records so far as a JSON array (valid until
 * the next call on td) and empties the log. Replaying them, single-threaded, through the
 * reference class (task_dispatcher.cc:93-140,167-188 ...) must give the same answers and the same
 * DumpInternals: that is what "concurrent callers are linearizable" means here, and what
 * tests/td_scenarios.py:concurrent_callers_linearize checks. */
int ydc_td_oplog_enable(ydc_td* td, int on);
const char* ydc_td_oplog_take(ydc_td* td);

#ifdef __cplusplus
}
#endif
#endif /* YADCC_DISPATCH_H_ */
