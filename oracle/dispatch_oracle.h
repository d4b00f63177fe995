/* ORACLE — test infrastructure, NOT product code.
 *
 * Plain-C restatement of the placement arithmetic of Tencent/yadcc's scheduler
 * (reference: yadcc/scheduler/task_dispatcher.cc). "One batch" == N sequential
 * TaskDispatcher::WaitForStartingNewTask(timeout = now) calls on a frozen
 * servant registry (no heartbeat, timer or free in between).
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks this file against the
 * reference's own golden vectors (task_dispatcher_test.cc:29-144,146-186,
 * 216-298) and, case by case on random pools, against the reference's own
 * translation units compiled verbatim (oracle/_ref/libyadcc_ref.so).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call anything in here.
 */
#ifndef ORACLE_DISPATCH_ORACLE_H_
#define ORACLE_DISPATCH_ORACLE_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_IDX_TIMEOUT 0xFFFFFFFFu       /* WaitStatus::Timeout */
#define ORACLE_IDX_ENV_NOT_FOUND 0xFFFFFFFEu /* WaitStatus::EnvironmentNotFound */
#define ORACLE_PRIORITY_DEDICATED 1u         /* api/scheduler.proto:43 */
#define ORACLE_PRIORITY_USER 2u              /* api/scheduler.proto:47 */

/* Servant registry in registration order (task_dispatcher.cc:205-206): raw
 * ServantPersonality fields (task_dispatcher.h:80-116), wire widths
 * (api/scheduler.proto:76-103). env_mask bit j <=> advertises digest j.
 * ip: opaque id of the text before ':' in observed_location; a task's
 * requestor_ip equals it <=> IsNetworkAddressEqual (task_dispatcher.cc:66-69). */
typedef struct oracle_servants {
  size_t n;
  const uint32_t* version;
  const uint32_t* num_processors;
  const uint32_t* current_load;
  const uint32_t* max_tasks;
  const uint32_t* priority;
  const uint64_t* total_memory;
  const uint64_t* memory_available;
  const uint64_t* env_mask; /* env_words words per servant: digest j = bit j % 64 of word j / 64 */
  const uint32_t* ip;
  uint32_t env_words;       /* 0 is read as 1 */
} oracle_servants;

/* TaskPersonality (task_dispatcher.h:48-66) in arrival order. env_id >= 64 * env_words
 * stands for a digest nobody advertises. */
typedef struct oracle_tasks {
  size_t n;
  const uint32_t* env_id;
  const uint32_t* min_version;
  const uint32_t* requestor_ip;
} oracle_tasks;

/* Literal restatement: every request rescans every servant, exactly like
 * WaitForStartingNewTask (task_dispatcher.cc:93-140). O(N*S).
 * running[S] is ServantDesc::running_tasks, updated in place.
 * out_util (nullable): chosen servant's double(running)/capacity at pick time
 * (task_dispatcher.cc:440-441), -1.0 when not granted. Returns #granted. */
size_t oracle_dispatch_scan(const oracle_servants* sv, uint64_t min_memory_for_new_task,
                            const oracle_tasks* tk, uint32_t* running, uint32_t* out_servant_idx,
                            double* out_util);

/* Same result through the slot-order formulation:
 * all (servant, running) slots sorted by (tier, fp64 utilisation, servant),
 * servants grouped into (env_mask, version) classes, per-class cursors, self
 * skipped per task_dispatcher.cc:372-396. O(slots log slots + N*classes).
 * Used for full-size parity where the literal scan would take minutes; itself
 * checked against oracle_dispatch_scan and the verbatim reference in tests/.
 * Masks of more than one word are handed to oracle_dispatch_scan. */
size_t oracle_dispatch_sorted(const oracle_servants* sv, uint64_t min_memory_for_new_task,
                              const oracle_tasks* tk, uint32_t* running,
                              uint32_t* out_servant_idx, double* out_util);

/* GetCapacityAvailable (task_dispatcher.cc:283-313) for one servant. */
uint64_t oracle_capacity_available(uint32_t num_processors, uint32_t current_load,
                                   uint32_t max_tasks, uint64_t total_memory,
                                   uint64_t memory_available, uint64_t running,
                                   uint64_t min_memory_for_new_task);

/* TryParseSize (yadcc/common/parse_size.cc:25-45). Returns 0 on success. */
int oracle_try_parse_size(const char* s, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif
