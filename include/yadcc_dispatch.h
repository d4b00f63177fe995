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
#define YDC_ERR_TOO_MANY_CLASSES (-5)/* > YDC_MAX_CLASSES distinct (env set, version) signatures */
#define YDC_ERR_NOT_CONVERGED (-6)   /* internal invariant broken (never expected) */

#define YDC_MAX_CLASSES 64u
#define YDC_MAX_ENVS 64u

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
  const uint64_t* env_mask;       /* bit j <=> advertises interned compiler digest j */
  const uint32_t* ip_id;          /* interned text before ':' of observed_location;
                                     equal ids <=> IsNetworkAddressEqual, task_dispatcher.cc:66-69 */
} ydc_servant_soa;

/* One heartbeat's worth of a servant row (KeepServantAlive replaces the
 * personality but keeps running_tasks, task_dispatcher.cc:195-201). */
typedef struct ydc_servant_row {
  uint32_t version, num_processors, current_load, max_tasks, flags, ip_id;
  uint64_t env_mask;
} ydc_servant_row;

/* Pending requests in arrival order (TaskPersonality, task_dispatcher.h:48-66). */
typedef struct ydc_task_soa {
  const uint32_t* env_id;       /* interned compiler digest; >= YDC_MAX_ENVS: nobody has it */
  const uint32_t* min_version;
  const uint32_t* requestor_ip; /* same interning as ydc_servant_soa::ip_id */
} ydc_task_soa;

typedef struct ydc_context ydc_context;

/* Counters of the most recent dispatch (debugging / bench). */
typedef struct ydc_stats {
  uint32_t n_tasks, n_servants, n_classes;
  uint32_t n_slots;        /* free (servant, running) slots generated */
  uint32_t key_bits;       /* significant bits of the slot sort key */
  uint32_t radix_passes;
  uint32_t n_chunks;       /* task chunks simulated in parallel */
  uint32_t rounds;         /* speculation rounds until the chunk states were consistent */
  uint32_t chunk_sims;     /* chunk simulations executed over all rounds */
  uint32_t granted, timeouts, env_not_found;
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
/* FreeTask / zombie / orphan sweeps: running_tasks[servant_idx[i]] -= 1
 * (task_dispatcher.cc:181). */
int ydc_release_slots(ydc_context* ctx, const uint32_t* servant_idx, uint32_t n);
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

/* Same with DEVICE pointers (task columns and outputs already in HBM);
 * asynchronous on the context stream except for one 16-byte convergence
 * read-back. out_* may be NULL. */
int ydc_dispatch_device(ydc_context* ctx, const ydc_task_soa* d_tasks, uint32_t n_tasks,
                        uint32_t flags, uint32_t* d_out_servant_idx, double* d_out_utilization,
                        uint32_t* d_out_running);

int ydc_synchronize(ydc_context* ctx);
int ydc_set_profiling(ydc_context* ctx, int on);
int ydc_get_stats(const ydc_context* ctx, ydc_stats* out);
/* Profiling on: per-kernel totals of the most recent dispatch, measured with HIP
 * events on the context stream, as JSON {"kernel": [launches, total_ms], ...}.
 * The string lives until the next dispatch. */
const char* ydc_kernel_profile(const ydc_context* ctx);

#ifdef __cplusplus
}
#endif
#endif /* YADCC_DISPATCH_H_ */
