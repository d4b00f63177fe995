/* Oracle (test infrastructure, NOT product code).
 *
 * C interface to the *verbatim* reference scheduler core. ref_driver.cc is
 * compiled together with /root/reference/yadcc/scheduler/task_dispatcher.cc,
 * running_task_bookkeeper.cc and yadcc/common/parse_size.cc (unmodified, read
 * where they lie) against the stand-in headers under oracle/shims/, into
 * oracle/_ref/libyadcc_ref.so. Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.
 *
 * Every function maps 1:1 onto a public TaskDispatcher method
 * (yadcc/scheduler/task_dispatcher.h:139-181); the snapshot helpers replay
 * the call pattern of SchedulerServiceImpl (scheduler_service_impl.cc:124-171,
 * 228-264): one KeepServantAlive per servant, then N sequential
 * WaitForStartingNewTask(timeout = now) calls.
 */
#ifndef ORACLE_REF_DRIVER_H_
#define ORACLE_REF_DRIVER_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ref_dispatcher ref_dispatcher;

#define REF_OK 0
#define REF_ENV_NOT_FOUND 1 /* WaitStatus::EnvironmentNotFound, task_dispatcher.h:42 */
#define REF_TIMEOUT 2       /* WaitStatus::Timeout, task_dispatcher.h:43 */

#define REF_IDX_TIMEOUT 0xFFFFFFFFu
#define REF_IDX_ENV_NOT_FOUND 0xFFFFFFFEu

ref_dispatcher* ref_create(void);
void ref_destroy(ref_dispatcher* d);

/* Fake coarse steady clock shared by all dispatchers of the process. */
void ref_clock_advance_ms(int64_t ms);
int64_t ref_clock_now_ns(void);
/* Fires every captured periodic timer once (== TaskDispatcher::OnExpirationTimer). */
void ref_fire_timers(void);

void ref_keep_servant_alive(ref_dispatcher* d, int version, const char* observed_location,
                            const char* reported_location, const char* const* env_digests,
                            size_t n_envs, uint64_t num_processors, uint64_t current_load,
                            uint64_t total_memory, uint64_t memory_available,
                            uint64_t max_tasks, int priority, int not_accepting_reason,
                            int64_t expires_in_ms);

int ref_wait_for_starting_new_task(ref_dispatcher* d, const char* requestor_ip,
                                   uint32_t min_version, const char* compiler_digest,
                                   int64_t expires_in_ms, int64_t timeout_in_ms, int prefetching,
                                   uint64_t* out_task_id, char* out_location,
                                   size_t location_cap);
int ref_keep_task_alive(ref_dispatcher* d, uint64_t task_id, int64_t new_expires_in_ms);
void ref_free_task(ref_dispatcher* d, uint64_t task_id);
size_t ref_notify_servant_running_tasks(ref_dispatcher* d, const char* servant_location,
                                        const uint64_t* servant_task_ids,
                                        const uint64_t* grant_ids, size_t n,
                                        uint64_t* out_unknown, size_t unknown_cap);
size_t ref_get_running_tasks(ref_dispatcher* d, uint64_t* out_servant_task_ids,
                             uint64_t* out_grant_ids, size_t cap);

/* ---- snapshot helpers (SoA, same columns as include/yadcc_dispatch.h) ---- */

/* Registers S servants in index order. Location = dotted(ip):port. Servant i
 * advertises digest j iff bit j of env_mask[i]; digest strings come from
 * ref_digest_name(). initial running_tasks[i] is primed by granting that many
 * tasks on a private digest before the real personality is installed (a
 * heartbeat replaces the personality but keeps running_tasks,
 * task_dispatcher.cc:195-201). */
void ref_load_servants(ref_dispatcher* d, size_t n, const uint32_t* version,
                       const uint32_t* num_processors, const uint32_t* current_load,
                       const uint32_t* max_tasks, const uint32_t* running_tasks,
                       const uint32_t* priority, const uint64_t* total_memory,
                       const uint64_t* memory_available, const uint64_t* env_mask,
                       const uint32_t* ip, const uint32_t* port);

/* Same with env_words 64-bit words per servant (digest j = bit j % 64 of word j / 64). */
void ref_load_servants_wide(ref_dispatcher* d, size_t n, const uint32_t* version,
                            const uint32_t* num_processors, const uint32_t* current_load,
                            const uint32_t* max_tasks, const uint32_t* running_tasks,
                            const uint32_t* priority, const uint64_t* total_memory,
                            const uint64_t* memory_available, const uint64_t* env_mask,
                            uint32_t env_words, const uint32_t* ip, const uint32_t* port);

/* TaskDispatcher::DumpInternals (task_dispatcher.cc:538-614) as JSON text (object keys sorted,
 * the way jsoncpp writes them). Returns the length needed; writes at most cap - 1 characters. */
size_t ref_dump_internals(ref_dispatcher* d, char* out, size_t cap);

/* n x FreeTask (one call across the binding instead of n). */
void ref_free_tasks(ref_dispatcher* d, const uint64_t* task_ids, size_t n);

/* N sequential WaitForStartingNewTask(timeout = now). out_servant_idx[i] is the
 * registry index of the granted servant, or REF_IDX_TIMEOUT / REF_IDX_ENV_NOT_FOUND.
 * env_id >= 64 * env_words of the loaded masks denotes a digest no servant has. out_latency_ns may be NULL.
 * Returns wall seconds spent inside the N calls. */
double ref_dispatch_batch(ref_dispatcher* d, size_t n, const uint32_t* env_id,
                          const uint32_t* min_version, const uint32_t* requestor_ip,
                          uint32_t* out_servant_idx, uint64_t* out_task_id,
                          uint64_t* out_latency_ns);

/* 64-hex-char digest string for env id j (deterministic). buf must hold 65 bytes. */
void ref_digest_name(uint32_t env_id, char* buf);

#ifdef __cplusplus
}
#endif
#endif
