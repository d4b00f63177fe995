// gpu_task_dispatcher.h — host-side mirror of the reference scheduler core.
//
// Same public surface as yadcc::scheduler::TaskDispatcher (reference
// yadcc/scheduler/task_dispatcher.h:139-181) and RunningTaskBookkeeper
// (running_task_bookkeeper.h:28-43): same method names, argument meaning and
// error behaviour, so SchedulerServiceImpl's call sites
// (scheduler_service_impl.cc:171,180,235,255,292,308,315) read unchanged. What
// is behind it is different: placement runs on the MI355X through the C-ABI
// (include/yadcc_dispatch.h), on integer SoA columns; the host keeps leases,
// zombies/orphans and the registry with hash indexes instead of the reference's
// per-call scans over all servants / all tasks.
//
// Plain C++17, no flare / protobuf types: TaskPersonality carries the compiler
// digest string that EnvironmentDesc wraps (api/env_desc.proto:27-28), and
// RunningTask the four fields of api/scheduler.proto:233-238.
#ifndef YADCC_AMD_GPU_TASK_DISPATCHER_H_
#define YADCC_AMD_GPU_TASK_DISPATCHER_H_

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

struct ydc_context;

namespace ydc {

// task_dispatcher.h:41-44 (same values).
enum class WaitStatus { EnvironmentNotFound = 0, Timeout = 1 };

// api/scheduler.proto:39-48.
enum ServantPriority { kServantPriorityUnknown = 0, kServantPriorityDedicated = 1, kServantPriorityUser = 2 };

// task_dispatcher.h:48-66.
struct TaskPersonality {
  std::string requestor_ip;
  std::uint32_t min_version = 0;
  std::string compiler_digest;  // EnvironmentDesc::compiler_digest
};

// task_dispatcher.h:69-77.
struct TaskAllocation {
  std::uint64_t task_id = 0;
  std::string servant_location;
};

// task_dispatcher.h:80-116.
struct ServantPersonality {
  int version = 0;
  std::string observed_location;
  std::string reported_location;
  std::vector<std::string> environments;  // compiler digests
  std::size_t num_processors = 0;
  std::size_t current_load = 0;
  std::size_t total_memory_in_bytes = 0;
  std::size_t memory_available_in_bytes = 0;
  std::size_t max_tasks = 0;
  int priority = kServantPriorityUnknown;
  int not_accepting_task_reason = 0;
};

// api/scheduler.proto:233-238.
struct RunningTask {
  std::uint64_t servant_task_id = 0;
  std::uint64_t task_grant_id = 0;
  std::string servant_location;
  std::string task_digest;
};

// Value-typed result of WaitForStartingNewTask (the reference's
// flare::Expected<TaskAllocation, WaitStatus>, task_dispatcher.h:139).
struct WaitResult {
  bool ok = false;
  TaskAllocation allocation;                           // valid when ok
  WaitStatus status = WaitStatus::EnvironmentNotFound; // valid when !ok
  int device_error = 0;  // != 0: the GPU path failed (YDC_ERR_*); never a silent fallback
  explicit operator bool() const { return ok; }
  const TaskAllocation* operator->() const { return &allocation; }
  WaitStatus error() const { return status; }
};

// running_task_bookkeeper.h:28-43. The flattened list is cached between
// changes: every daemon polls GetRunningTasks once a second
// (daemon/local/running_task_keeper.cc:31-33) while a servant's entry changes
// only with its own heartbeat.
class RunningTaskBookkeeper {
 public:
  void SetServantRunningTasks(const std::string& servant_location, std::vector<RunningTask> tasks);
  void DropServant(const std::string& servant_location);
  std::vector<RunningTask> GetRunningTasks() const;

 private:
  mutable std::mutex lock_;
  std::unordered_map<std::string, std::vector<RunningTask>> running_tasks_;
  mutable std::vector<RunningTask> flattened_;
  mutable bool flattened_valid_ = false;
};

class GpuTaskDispatcher {
 public:
  using Clock = std::chrono::steady_clock;
  struct Options {
    // HIP device ordinal. -1: no device — the registry / lease bookkeeping works,
    // every WaitForStartingNewTask fails with device_error = YDC_ERR_NO_DEVICE
    // (there is no CPU placement path).
    int device = 0;
    // --servant_min_memory_for_accepting_new_task (task_dispatcher.cc:35-38).
    std::string servant_min_memory_for_accepting_new_task = "10G";
    // Own 1 s expiration thread (task_dispatcher.cc:81-82). Tests drive
    // OnExpirationTimer() themselves.
    bool start_expiration_timer = true;
    // Coarse steady clock (flare::ReadCoarseSteadyClock); tests inject a fake one.
    std::function<Clock::time_point()> clock;
  };

  explicit GpuTaskDispatcher(const Options& options);
  ~GpuTaskDispatcher();
  GpuTaskDispatcher(const GpuTaskDispatcher&) = delete;
  GpuTaskDispatcher& operator=(const GpuTaskDispatcher&) = delete;

  // The dispatcher's coarse steady clock (flare::ReadCoarseSteadyClock in the reference).
  Clock::time_point Now() const { return options_.clock ? options_.clock() : Clock::now(); }

  // != 0 when the device context could not be created (YDC_ERR_*).
  int device_status() const { return device_status_; }
  const std::string& device_error_message() const { return device_error_; }

  // ---- the six methods of task_dispatcher.h:139-181 ----
  WaitResult WaitForStartingNewTask(const TaskPersonality& personality,
                                    std::chrono::nanoseconds expires_in,
                                    Clock::time_point timeout, bool prefetching);
  bool KeepTaskAlive(std::uint64_t task_id, std::chrono::nanoseconds new_expires_in);
  void FreeTask(std::uint64_t task_id);
  void KeepServantAlive(const ServantPersonality& servant, std::chrono::nanoseconds expires_in);
  std::vector<std::uint64_t> NotifyServantRunningTasks(const std::string& servant_location,
                                                       std::vector<RunningTask> tasks);
  std::vector<RunningTask> GetRunningTasks() const;

  // ---- batch flavour of the first one ----
  // Equivalent to calling WaitForStartingNewTask(personalities[i], expires_in,
  // now, prefetching[i]) for i = 0..n-1 back to back (what
  // SchedulerServiceImpl::WaitForStartingTask does for every grant after the
  // first, scheduler_service_impl.cc:234-264), as ONE device batch.
  std::vector<WaitResult> WaitForStartingNewTasks(const std::vector<TaskPersonality>& personalities,
                                                  std::chrono::nanoseconds expires_in,
                                                  const std::vector<bool>& prefetching);

  // task_dispatcher.cc:498-536. Public so that a host without the timer thread
  // (tests, a fiber runtime's own timer) can drive it.
  void OnExpirationTimer();
  // task_dispatcher.cc:538-614, same keys, as a JSON string.
  std::string DumpInternals();

 private:
  struct Servant {
    std::uint64_t uid;  // stable identity (the reference's ServantDesc pointer)
    ServantPersonality personality;
    Clock::time_point discovered_at, expires_at;
    std::size_t running_tasks = 0;
    std::size_t ever_assigned_tasks = 0;
    std::vector<std::uint32_t> env_bits;  // interned digests (bit numbers), listing order
    std::uint32_t ip_id = 0;     // interned longest requestor address the location answers to
    // The shorter ones (location prefixes ending before an earlier ':'; none for "a.b.c.d:port").
    std::vector<std::string> shorter_prefixes;
    std::unordered_set<std::uint64_t> grants;  // live task ids (incl. zombies) on this servant
  };
  struct Task {
    std::uint64_t task_id;
    TaskPersonality personality;
    std::uint64_t servant_uid;
    Clock::time_point started_at, expires_at;
    bool is_prefetch = false;
    bool zombie = false;
  };
  struct Pending {
    const TaskPersonality* personality;
    std::chrono::nanoseconds expires_in;
    Clock::time_point deadline;
    bool prefetching;
    bool done = false;
    std::uint64_t tried_epoch = ~0ull;  // wake epoch of the last failed attempt
    WaitResult result;
  };

  std::size_t CapacityAvailable(const Servant& s) const;  // task_dispatcher.cc:283-313
  std::uint32_t InternIp(const std::string& ip, bool create);
  // Host id of a requestor address (0: no servant answers to it).
  std::uint32_t RequestorId(const std::string& ip);
  std::uint32_t LookupEnv(const std::string& digest) const;
  std::vector<std::uint32_t> AcquireEnvBits(const std::vector<std::string>& digests);
  void ReleaseEnvBits(const std::vector<std::string>& digests);
  // 64-bit words an environment mask needs for every bit number handed out so far.
  std::uint32_t EnvWords() const { return next_env_bit_ ? (next_env_bit_ + 63) / 64 : 1; }
  void UnsafeFreeTasks(const std::vector<std::uint64_t>& task_ids);  // :167-188
  void UnsafeSweepZombiesOf(Servant* servant, const std::unordered_set<std::uint64_t>& running);
  void UnsafeSweepOrphans();
  int UnsafeSyncDevice();
  int UnsafeSyncAliases();
  // Places `batch` (arrival order) as one device batch and registers the grants.
  void UnsafeDispatch(const std::vector<Pending*>& batch);
  void UnsafeDrainQueue();
  void TimerLoop();

  Options options_;
  ydc_context* ctx_ = nullptr;
  int device_status_ = 0;
  std::string device_error_;
  std::size_t min_memory_for_new_task_ = 0;

  mutable std::mutex allocation_lock_;           // task_dispatcher.h:289
  std::condition_variable allocation_cv_;        // task_dispatcher.h:290
  std::vector<std::unique_ptr<Servant>> servants_;  // registration order == tie-break order
  std::unordered_map<std::string, std::uint32_t> index_of_location_;
  std::unordered_map<std::uint64_t, std::uint32_t> index_of_uid_;
  std::uint64_t next_servant_uid_ = 1;
  std::unordered_map<std::uint64_t, Task> tasks_;
  std::uint64_t next_task_id_ = 0;  // task_dispatcher.h:218
  std::uint64_t wake_epoch_ = 0;  // bumped where the reference notifies its waiters (:187)

  // interning
  std::unordered_map<std::string, std::uint32_t> ip_ids_;
  // A location with several ':' answers to several requestor addresses (IsNetworkAddressEqual,
  // task_dispatcher.cc:66-69: every prefix that ends right before a ':'). The servant's ip_id is
  // the longest of them; the shorter ones are counted here and become table aliases
  // (ydc_set_host_aliases) the first time a requestor actually presents one.
  std::unordered_map<std::string, std::uint32_t> shorter_prefix_refs_;  // prefix -> servants having it
  std::unordered_map<std::string, std::uint32_t> alias_ids_;            // presented ones -> host id
  bool aliases_dirty_ = false;
  std::unordered_map<std::string, std::pair<std::uint32_t, std::uint32_t>> env_ids_;  // digest -> (bit, refs)
  std::vector<std::uint32_t> free_env_bits_;
  std::uint32_t next_env_bit_ = 0;  // bit numbers handed out so far (freed ones are reused)

  // device mirror bookkeeping (deltas applied before the next dispatch)
  bool need_full_upload_ = true;
  std::vector<std::uint32_t> dirty_rows_;
  std::vector<std::uint8_t> row_is_dirty_;
  std::vector<std::uint32_t> pending_release_;

  // Request columns and the result array of a device batch, reused from batch to batch and
  // page-locked where the runtime lets us (ydc_host_alloc): ydc_dispatch then reads and writes
  // them in place. Guarded by allocation_lock_.
  struct HostColumn {
    std::uint32_t* p = nullptr;
    std::size_t cap = 0;
    bool pinned = false;
    std::uint32_t* ensure(std::size_t n);
    ~HostColumn();
  };
  HostColumn col_env_, col_minv_, col_rip_, col_out_;

  // request combining
  std::mutex queue_lock_;
  std::deque<Pending*> queue_;
  std::vector<Pending*> waiting_;  // found no free servant; arrival order

  RunningTaskBookkeeper running_task_bookkeeper_;

  std::thread timer_;
  std::mutex timer_lock_;
  std::condition_variable timer_cv_;
  bool stopping_ = false;
};

}  // namespace ydc
#endif  // YADCC_AMD_GPU_TASK_DISPATCHER_H_
