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

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "flat_string_map.h"

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

// ---- string-view forms of the inputs ----
// What the RPC layer holds are bytes on the wire; building std::strings out of them only to
// look them up costs more than the lookups. The C-ABI wrappers (td_api.cc) and the batch entry
// points below pass views; the std::string methods above them are the reference's signatures.
struct RequestView {
  std::string_view requestor_ip;
  std::string_view compiler_digest;
  std::uint32_t min_version = 0;
  bool prefetching = false;
};

struct ServantView {
  int version = 0;
  std::string_view observed_location, reported_location;
  const std::string_view* environments = nullptr;
  std::size_t n_environments = 0;
  std::size_t num_processors = 0, current_load = 0, total_memory_in_bytes = 0,
              memory_available_in_bytes = 0, max_tasks = 0;
  int priority = kServantPriorityUnknown;
  int not_accepting_task_reason = 0;
};

struct RunningTaskView {
  std::uint64_t servant_task_id = 0;
  std::uint64_t task_grant_id = 0;
  std::string_view servant_location, task_digest;
};

// running_task_bookkeeper.h:28-43. The flattened list is cached between
// changes: every daemon polls GetRunningTasks once a second
// (daemon/local/running_task_keeper.cc:31-33) while a servant's entry changes
// only with its own heartbeat — and most heartbeats report the list they
// reported a second ago, so a report is compared with what is stored before
// anything is replaced or invalidated.
// The flattened running-task list as columns (what the C-ABI hands out without a copy,
// ydc_td_running_tasks_acquire): ids as arrays, strings as offsets into one pool — every string
// NUL-terminated there, a servant's location stored once for all its tasks, a digest once per
// distinct value.
struct RunningTaskColumns {
  std::vector<std::uint64_t> servant_task_ids, task_grant_ids;
  std::vector<std::uint32_t> location_off, digest_off;  // per task: where its string starts in `strings`
  std::vector<std::uint32_t> location_len, digest_len;
  std::string strings;
};

class RunningTaskBookkeeper {
 public:
  using Snapshot = std::shared_ptr<const std::vector<RunningTask>>;
  using ColumnsSnapshot = std::shared_ptr<const RunningTaskColumns>;
  void SetServantRunningTasks(const std::string& servant_location, std::vector<RunningTask> tasks);
  // The entries `keep[0..n_keep)` of `tasks` (report order).
  void SetServantRunningTasks(std::string_view servant_location, const RunningTaskView* tasks,
                              const std::uint32_t* keep, std::size_t n_keep);
  void DropServant(std::string_view servant_location);
  std::vector<RunningTask> GetRunningTasks() const;
  // The flattened list without the copy: shared with later callers until a report changes it.
  Snapshot GetRunningTasksShared() const;
  // ... and as columns: built once per change of the list, shared by every caller until the next.
  ColumnsSnapshot GetRunningTasksColumns() const;
  std::uint64_t rebuilds() const { return rebuilds_; }

 private:
  mutable std::mutex lock_;
  FlatStringMap<std::vector<RunningTask>> running_tasks_;
  mutable Snapshot flattened_;
  mutable ColumnsSnapshot columns_;
  mutable std::uint64_t rebuilds_ = 0;
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
    // The expiry index keeps entries of renewed and freed leases until their second comes round;
    // they are swept when they outnumber the live leases four to one by this much (tests: small).
    std::size_t lease_sweep_slack = 1u << 20;
    // How long a caller whose request is queued spins for its answer (pause instructions; a
    // device turn takes 7 - 40 us) before it sleeps until the end of the next turn.
    int caller_spins = 3000;
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
  // The same on views, results into the caller's arrays (no std::string per request on either
  // side): out_status[i] = 0 granted / 1 EnvironmentNotFound / 2 Timeout, or the negative
  // YDC_ERR_* of a failed device batch; out_task_ids[i] = the grant id or ~0; the location of
  // the granted servant NUL-terminated at out_locations + i * location_stride (nullable). A
  // location that does not fit its stride gives the grant back at once and reports
  // YDC_ERR_CAPACITY for that request. Returns 0, or the last negative status.
  int WaitForStartingNewTasksInto(std::size_t n, const RequestView* requests,
                                  std::chrono::nanoseconds expires_in, std::int32_t* out_status,
                                  std::uint64_t* out_task_ids, char* out_locations,
                                  std::size_t location_stride);

  // ---- view forms of the registry methods (same behaviour as the std::string ones) ----
  void KeepServantAlive(const ServantView& servant, std::chrono::nanoseconds expires_in);
  std::vector<std::uint64_t> NotifyServantRunningTasks(std::string_view servant_location,
                                                       const RunningTaskView* tasks, std::size_t n);
  RunningTaskBookkeeper::Snapshot GetRunningTasksShared() const {
    return running_task_bookkeeper_.GetRunningTasksShared();
  }
  RunningTaskBookkeeper::ColumnsSnapshot GetRunningTasksColumns() const {
    return running_task_bookkeeper_.GetRunningTasksColumns();
  }
  void FreeTasks(const std::uint64_t* task_ids, std::size_t n);  // n FreeTask calls, one lock

  // Where the host class spends its time (cumulative): inside the device API (registry
  // deltas + ydc_dispatch, the GPU's share) and in the class itself.
  struct HostStats {
    std::uint64_t requests = 0, batches = 0, device_ns = 0, host_ns = 0;
    std::uint64_t heartbeats = 0, heartbeats_unchanged = 0, bookkeeper_rebuilds = 0;
    std::uint64_t lease_pages = 0;
    // What the last OnExpirationTimer looked at (lease entries of the due buckets, not all leases),
    // how long it held the lock, and the longest any of them did.
    std::uint64_t timer_ticks = 0, timer_lease_entries_seen = 0, timer_last_ns = 0, timer_max_ns = 0;
    std::uint64_t lease_wheel_entries = 0;
  };
  HostStats host_stats() const;

  // ---- test switch: the order in which calls took effect ----
  // With the log on, every call that reads or changes the dispatcher's state appends one record
  // at the moment it takes effect, under allocation_lock_: each placement attempt of a request
  // (requestor, digest, min_version, lease, clock reading, answer), each FreeTask id, lease renewal,
  // heartbeat (the whole personality), servant report and timer tick. Concurrent callers are
  // linearizable iff replaying that sequence through the reference, single-threaded, gives the
  // same answers and the same final state — tests/td_scenarios.py does exactly that. Off by
  // default; costs one branch per call when off.
  void EnableOpLog(bool on);
  std::string TakeOpLog();  // a JSON array; the log is emptied

  // task_dispatcher.cc:498-536. Public so that a host without the timer thread
  // (tests, a fiber runtime's own timer) can drive it.
  void OnExpirationTimer();
  // task_dispatcher.cc:538-614, same keys, as a JSON string.
  std::string DumpInternals();

 private:
  static constexpr std::uint64_t kNoTask = ~0ull;
  struct Servant {
    std::uint64_t uid;  // stable identity (the reference's ServantDesc pointer)
    std::uint32_t index = 0;  // current registry position
    bool removed = false;     // expired: out of the registry, alive until its orphans are freed
    ServantPersonality personality;
    Clock::time_point discovered_at, expires_at;
    std::size_t running_tasks = 0;
    std::size_t ever_assigned_tasks = 0;
    std::vector<std::uint32_t> env_bits;  // interned digests (bit numbers), listing order
    // observed_location again, NUL-terminated, when it fits: what a grant copies out.
    char location_short[24] = {0};
    std::uint8_t location_len = 0xFF;  // 0xFF: longer than that, read personality.observed_location
    std::uint32_t ip_id = 0;     // interned longest requestor address the location answers to
    // The shorter ones (location prefixes ending before an earlier ':'; none for "a.b.c.d:port").
    std::vector<std::string> shorter_prefixes;
    // Live task ids (incl. zombies) on this servant: a doubly linked list threaded through the
    // task records (Task::prev / next), newest first. What the reference rebuilds with a scan
    // over every task per heartbeat (task_dispatcher.cc:257-262,453-476).
    std::uint64_t grants_head = kNoTask;
    std::size_t n_grants = 0, n_zombies = 0;
  };
  // One lease. The reference keeps TaskDesc{TaskPersonality (two strings), RefPtr<ServantDesc>, …}
  // in an unordered_map node per grant (task_dispatcher.h:199-216); here a fixed-size record in
  // a table indexed by the task id itself (ids are handed out in sequence, task_dispatcher.cc:127),
  // strings as pool ids (digest) or in place (requestor addresses of up to 15 characters).
  struct alignas(64) Task {  // one cache line
    Servant* servant = nullptr;  // outlives the task: orphans are freed by the sweep that removes it
    Clock::time_point started_at, expires_at;
    std::uint64_t prev = kNoTask, next = kNoTask;  // neighbours in servant's grant list
    std::uint32_t digest_name = 0;                 // NamePool id of the compiler digest
    char ip_inline[16];       // the requestor address; or, pooled, its NamePool id in the first 4 bytes
    std::uint8_t ip_len = 0;  // 0xFF: pooled
    bool live = false, is_prefetch = false, zombie = false;
  };
  static_assert(sizeof(Task) == 64, "a lease is one cache line");
  // Task records by id: pages of 4096 consecutive ids, a page is freed when its last task is.
  class TaskTable {
   public:
    static constexpr unsigned kPageBits = 12;
    Task* find(std::uint64_t id);
    // The record of a task that is known to be live, by address arithmetic alone: a neighbour
    // in a grant list is only ever WRITTEN through this (no load from its cold cache line).
    Task* slot(std::uint64_t id) {
      return &pages_[(id >> kPageBits) - first_page_]->tasks[id & ((1u << kPageBits) - 1)];
    }
    Task* create(std::uint64_t id);  // id = the next one in sequence
    void erase(std::uint64_t id);
    std::size_t size() const { return live_; }
    std::size_t pages() const { return pages_.size(); }  // (tests: stays bounded)
    template <class F>
    void for_each(F&& f) {
      for (std::size_t p = 0; p != pages_.size(); ++p) {
        if (!pages_[p]) continue;
        for (std::size_t k = 0; k != (1u << kPageBits); ++k)
          if (pages_[p]->tasks[k].live) f(((first_page_ + p) << kPageBits) + k, pages_[p]->tasks[k]);
      }
    }

   private:
    struct Page {
      Task tasks[1u << kPageBits];
      std::uint32_t live = 0;
    };
    std::deque<std::unique_ptr<Page>> pages_;
    std::vector<std::unique_ptr<Page>> spare_;  // emptied pages, reused before the allocator is asked
    std::uint64_t first_page_ = 0;
    std::uint64_t next_id_ = 0;  // the id create() hands out next
    std::size_t live_ = 0;
  };
  // Strings that task records refer to by number (permanent: bounded by the distinct digests
  // servants ever advertised and the distinct long requestor addresses that were granted).
  class NamePool {
   public:
    std::uint32_t intern(std::string_view s);
    const std::string& name(std::uint32_t id) const { return names_[id]; }

   private:
    FlatStringMap<std::uint32_t> ids_;
    std::deque<std::string> names_;
  };
  // Leases filed under the second in which they run out. The reference's timer walks every
  // lease once a second under the lock (task_dispatcher.cc:523-535); here OnExpirationTimer opens
  // only the buckets that are due. A renewal (KeepTaskAlive) files the lease again under its
  // new second and leaves the old entry behind: an entry counts only while the lease still
  // exists, is no zombie, and runs out in the second of the bucket it is found in. Stale
  // entries go with their bucket — or, when they outnumber the live leases four to one, in a sweep.
  class LeaseWheel {
   public:
    static std::int64_t SecondOf(Clock::time_point t) {
      const std::int64_t ns = t.time_since_epoch().count();
      return ns >= 0 ? ns / 1000000000 : -((-ns + 999999999) / 1000000000);  // floor
    }
    void File(std::uint64_t id, Clock::time_point expires_at) {
      // (the grants of one batch run out at the same instant: no division, no lookup for them)
      if (!last_ || expires_at != last_time_) {
        const std::int64_t second = SecondOf(expires_at);
        if (!last_ || last_second_ != second) {
          last_ = &buckets_[second];  // (std::map: the address of a mapped vector is stable)
          last_second_ = second;
        }
        last_time_ = expires_at;
      }
      last_->push_back(id);
      ++entries_;
    }
    // still(id, second) -> the entry counts; expire(id) -> true: done with it (marked), false: its
    // lease runs out later in the second `now` lies in — the entry stays.
    template <class Still, class Expire>
    void Due(Clock::time_point now, Still&& still, Expire&& expire) {
      const std::int64_t this_second = SecondOf(now);
      while (!buckets_.empty() && buckets_.begin()->first <= this_second) {
        auto it = buckets_.begin();
        std::vector<std::uint64_t>& ids = it->second;
        std::size_t kept = 0;
        for (std::uint64_t id : ids)
          if (still(id, it->first) && !expire(id)) ids[kept++] = id;
        entries_ -= ids.size() - kept;
        if (kept != 0 && it->first == this_second) {
          ids.resize(kept);
          break;  // (the bucket of the running second: looked at again next time)
        }
        if (last_ == &ids) last_ = nullptr;
        buckets_.erase(it);
      }
    }
    template <class Still>
    void Sweep(Still&& still) {
      for (auto it = buckets_.begin(); it != buckets_.end();) {
        std::vector<std::uint64_t>& ids = it->second;
        std::size_t kept = 0;
        for (std::uint64_t id : ids)
          if (still(id, it->first)) ids[kept++] = id;
        entries_ -= ids.size() - kept;
        ids.resize(kept);
        if (kept == 0) {
          if (last_ == &ids) last_ = nullptr;
          it = buckets_.erase(it);
        } else {
          ++it;
        }
      }
    }
    std::size_t entries() const { return entries_; }
    std::size_t buckets() const { return buckets_.size(); }

   private:
    std::map<std::int64_t, std::vector<std::uint64_t>> buckets_;
    std::vector<std::uint64_t>* last_ = nullptr;  // the bucket of the last File (a batch's grants share it)
    std::int64_t last_second_ = 0;
    Clock::time_point last_time_{};
    std::size_t entries_ = 0;
  };
  struct EnvEntry {
    std::uint32_t bit = 0, refs = 0, name = 0;  // mask bit, servants' references, NamePool id
  };
  struct Pending {
    RequestView request;
    std::chrono::nanoseconds expires_in;
    Clock::time_point deadline;
    bool done = false;                   // (the thread that holds allocation_lock_)
    std::uint32_t tries = 0;            // placement attempts so far (1 unless it was parked)
    // The request as the device sees it (interned digest, host id of the requestor), looked up once
    // and kept while the interning tables stay as they are (registry_epoch_): a parked request is
    // placed again at every wake-up. sig: the number of its (digest, min_version, host) triple
    // among the parked requests' (UnsafeDispatchSegmented).
    std::uint64_t cols_epoch = ~0ull;
    std::uint32_t col_env = 0, col_rip = 0, col_name = 0, sig = 0;
    // Parked (found no free servant): a member of its triple's list, in arrival order.
    std::uint64_t arrival = 0;
    Pending *park_prev = nullptr, *park_next = nullptr;
    bool parked = false;
    WaitResult result;
    // A parked owner sleeps on its OWN condition variable (with allocation_lock_): whoever frees a
    // slot re-places the parked requests itself, as one device batch, and wakes only the owners of
    // those it served. The reference wakes every waiter for every FreeTask and lets each of them
    // scan the registry under the one lock (task_dispatcher.h:281-288 says it does not scale,
    // .cc:116-118,185-187).
    std::condition_variable cv;
    bool sleeping = false;  // inside cv.wait (guarded by allocation_lock_)
    // Set by the thread that placed this request for its owner, as its LAST access to the record:
    // the owner — spinning for it instead of sleeping on the lock — may return at once.
    std::atomic<bool> published{false};
  };
  // FreeTask never waits for a device turn of somebody else's: when allocation_lock_ is taken its
  // ids are queued, and whoever holds the lock applies them — on entering its critical section
  // and again after leaving it (Section) — in arrival order, each id a FreeTask call of its own.
  // sleepers_: threads asleep (or about to be) on their condition variable hold no lock and apply
  // nothing; a FreeTask that finds one takes the lock itself, so that its wake-up is not lost.
  // wake_pending_: where the reference calls notify_all (:187) this is set; the thread that set it
  // re-places the parked requests before it lets go of the lock (UnsafeServeWoken).
  std::vector<std::uint64_t> free_queue_;  // guarded by queue_lock_
  std::atomic<std::uint32_t> free_queued_{0};
  std::atomic<std::uint32_t> sleepers_{0};
  // allocation_lock_ is held (a hint for the spinners of WaitForStartingNewTask: they read this
  // word — shared, in their caches — and go for the lock's own cache line only when it says free).
  std::atomic<bool> busy_{false};
  // Callers whose request is queued spin briefly for their answer and then sleep until the end of
  // the next device turn: turn_seq_ counts the turns, and a sleeper waits (futex) for it to move.
  // Nothing of a request's own record is touched after its answer is published.
  std::atomic<std::uint32_t> turn_seq_{0};
  std::atomic<std::uint32_t> turn_sleepers_{0};
  void WakeTurnSleepers();
  bool wake_pending_ = false;  // guarded by allocation_lock_
  void UnsafeServeWoken();
  bool UnsafeApplyQueuedFrees();  // true: something was applied (waiters were woken)
  bool UnsafeApplyFrees(const std::vector<std::uint64_t>& ids);
  // allocation_lock_ held for a scope.
  class Section {
   public:
    explicit Section(GpuTaskDispatcher* d) : d_(d), lk_(d->allocation_lock_) {
      d->busy_.store(true, std::memory_order_relaxed);
      d->UnsafeApplyQueuedFrees();
    }
    Section(GpuTaskDispatcher* d, std::unique_lock<std::mutex>&& held) : d_(d), lk_(std::move(held)) {
      d->busy_.store(true, std::memory_order_relaxed);
      d->UnsafeApplyQueuedFrees();
    }
    ~Section() {
      // Whoever holds the lock serves what has queued up behind it before it goes: requests of
      // callers that are spinning or asleep for their answer (a bounded number of turns — then
      // the lock is let go and one of them takes over), and the ids a FreeTask left behind when
      // it found the lock taken: nobody else may come by soon.
      for (int turns = 0;; ++turns) {
        d_->UnsafeServeWoken();
        if (turns < kExitTurns && d_->queued_.load(std::memory_order_relaxed) != 0) {
          d_->UnsafeDrainQueue();
          continue;
        }
        d_->busy_.store(false, std::memory_order_relaxed);
        lk_.unlock();
        // Store-buffering with FreeTasks ("queue the ids; fence; try_lock" there, "unlock; fence;
        // look at the queue" here): with a full fence on both sides either that try_lock sees the
        // lock free, or this load sees the queued ids. A failed try_lock orders nothing by itself
        // in the C++ model — the fences do not lean on the mutex implementation's own barriers.
        std::atomic_thread_fence(std::memory_order_seq_cst);
        const bool requests = d_->queued_.load(std::memory_order_seq_cst) != 0;
        if (d_->free_queued_.load(std::memory_order_seq_cst) == 0 && !(requests && turns < kExitTurns)) {
          if (requests) d_->WakeTurnSleepers();  // (this thread has done its share: one of them takes the lock)
          return;
        }
        if (!lk_.try_lock()) return;  // (the new holder serves them, on its way in or out)
        d_->busy_.store(true, std::memory_order_relaxed);
        d_->UnsafeApplyQueuedFrees();
      }
    }
    static constexpr int kExitTurns = 2;
    std::unique_lock<std::mutex>& lock() { return lk_; }
   private:
    GpuTaskDispatcher* d_;
    std::unique_lock<std::mutex> lk_;
  };

  std::size_t CapacityAvailable(const Servant& s) const;  // task_dispatcher.cc:283-313
  std::uint32_t InternIp(std::string_view ip, bool create);
  // Host id of a requestor address (0: no servant answers to it).
  std::uint32_t RequestorId(std::string_view ip);
  const EnvEntry* LookupEnv(std::string_view digest) const;
  template <class Strings>
  std::vector<std::uint32_t> AcquireEnvBits(const Strings& digests, std::size_t n);
  void ReleaseEnvBits(const std::vector<std::string>& digests);
  // 64-bit words an environment mask needs for every bit number handed out so far.
  std::uint32_t EnvWords() const { return next_env_bit_ ? (next_env_bit_ + 63) / 64 : 1; }
  void UnsafeFreeTasks(const std::uint64_t* task_ids, std::size_t n);  // :167-188
  void UnsafeSweepZombiesOf(Servant* servant, const RunningTaskView* reported, std::size_t n);
  int UnsafeSyncDevice();
  int UnsafeSyncAliases();
  std::uint32_t DeviceFlags(const Servant& s) const;
  void UnsafePackDirtyRows(std::uint32_t* env_words);
  // A batch in arrival order: contiguous views, or the requests of parked / queued callers.
  struct RequestSpan {
    const RequestView* views = nullptr;
    Pending* const* pending = nullptr;
    std::size_t n = 0;
    bool cached = false;  // pending[i]'s columns are valid (UnsafeCacheColumns)
    const RequestView& operator[](std::size_t i) const { return views ? views[i] : pending[i]->request; }
  };
  // Places the span as one device batch (COMMIT); the answers (registry index or YDC_IDX_*) are
  // in col_out_. Returns YDC_OK or the device error.
  int UnsafePlace(const RequestSpan& batch);
  // Registers the grant of `r` on servants_[servant_index]; returns the task id.
  std::uint64_t UnsafeGrant(const RequestView& r, std::uint32_t digest_name, std::uint32_t servant_index,
                            std::chrono::nanoseconds expires_in, Clock::time_point now);
  template <class Sink>
  int UnsafePlaceAndGrant(std::size_t n, const RequestView* requests, std::chrono::nanoseconds expires_in,
                          Sink&& sink);
  void UnsafeDispatch(const std::vector<Pending*>& batch);
  // Parked requests are kept per (digest, min_version, requestor host) triple, each list in
  // arrival order. When the waiters are woken (wake_epoch_ moved) they are placed again in global
  // arrival order — a merge of the lists' heads — in segments of what a device turn takes; a triple
  // that has come back Timeout in this turn does so for every later request of the turn (grants
  // only take capacity away), so the rest of its list is not even looked at: a FreeTask on a pool
  // with ten thousand parked waiters of a handful of triples costs one small device turn.
  struct ParkedList {
    Pending *head = nullptr, *tail = nullptr;
  };
  std::vector<ParkedList> parked_;        // by Pending::sig, valid for parked_epoch_
  std::uint64_t parked_epoch_ = ~0ull;    // the registry_epoch_ the triples were numbered in
  std::size_t parked_count_ = 0;
  std::uint64_t next_arrival_ = 0;
  std::uint64_t retried_epoch_ = 0;       // wake epoch at which the parked requests were last placed again
  void UnsafePark(Pending* r);
  void UnsafeUnpark(Pending* r);
  void UnsafeReindexParked();
  void UnsafeCacheColumns(Pending* r);
  void UnsafeDrainQueue();
  void TimerLoop();
  std::string TaskRequestorIp(const Task& t) const;

  Options options_;
  ydc_context* ctx_ = nullptr;
  int device_status_ = 0;
  std::string device_error_;
  std::size_t min_memory_for_new_task_ = 0;

  mutable std::mutex allocation_lock_;           // task_dispatcher.h:289
  // (task_dispatcher.h:290's one condition variable is one per parked request here: Pending::cv)
  std::vector<std::unique_ptr<Servant>> servants_;  // registration order == tie-break order
  FlatStringMap<std::uint32_t> index_of_location_;
  std::uint64_t next_servant_uid_ = 1;
  // expires_at of servants_[i], side by side: what the timer scans every second (16k servants:
  // 128 KB in a row instead of 16k records behind pointers).
  std::vector<Clock::time_point> servant_expires_at_;
  TaskTable tasks_;
  LeaseWheel lease_wheel_;
  void FileLease(std::uint64_t id, Clock::time_point expires_at);
  NamePool names_;
  std::uint64_t next_task_id_ = 0;  // task_dispatcher.h:218
  std::uint64_t wake_epoch_ = 0;  // bumped where the reference notifies its waiters (:187)
  // Bumped whenever a digest / requestor-host lookup may answer differently than before (a servant
  // registered, changed its environments or expired; an alias was created).
  std::uint64_t registry_epoch_ = 0;
  struct SigKey {
    std::uint32_t env, minv, rip;
    bool operator==(const SigKey& o) const { return env == o.env && minv == o.minv && rip == o.rip; }
  };
  struct SigHash {
    std::size_t operator()(const SigKey& k) const {
      return (std::size_t)((k.env * 0x9E3779B97F4A7C15ull) ^ ((std::uint64_t)k.rip << 20) ^ k.minv);
    }
  };
  std::unordered_map<SigKey, std::uint32_t, SigHash> sig_ids_;  // valid for sig_epoch_
  std::uint64_t sig_epoch_ = ~0ull;
  std::vector<std::uint8_t> sig_dead_;

  // interning
  FlatStringMap<std::uint32_t> ip_ids_;
  // A location with several ':' answers to several requestor addresses (IsNetworkAddressEqual,
  // task_dispatcher.cc:66-69: every prefix that ends right before a ':'). The servant's ip_id is
  // the longest of them; the shorter ones are counted here and become table aliases
  // (ydc_set_host_aliases) the first time a requestor actually presents one.
  FlatStringMap<std::uint32_t> shorter_prefix_refs_;  // prefix -> servants having it
  FlatStringMap<std::uint32_t> alias_ids_;            // presented ones -> host id
  bool aliases_dirty_ = false;
  FlatStringMap<EnvEntry> env_ids_;  // digest -> (bit, refs, name)
  std::vector<std::uint32_t> free_env_bits_;
  std::uint32_t next_env_bit_ = 0;  // bit numbers handed out so far (freed ones are reused)

  // device mirror bookkeeping (deltas applied before the next dispatch)
  bool need_full_upload_ = true;
  std::vector<std::uint32_t> dirty_rows_;
  std::vector<std::uint8_t> row_is_dirty_;
  std::vector<std::uint32_t> pending_release_;
  struct SyncRow {  // layout of ydc_servant_row (include/yadcc_dispatch.h)
    std::uint32_t version, num_processors, current_load, max_tasks, flags, ip_id;
    std::uint64_t env_mask;
  };
  std::vector<SyncRow> sync_rows_;       // dirty_rows_ packed for the device API
  std::vector<std::uint64_t> sync_env_;  // ... their environment masks

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
  std::vector<std::uint32_t> col_digest_name_;  // NamePool id of every request's digest (known ones)

  // request combining
  // (held for a push or a swap of a vector: a thread that finds it taken is a few nanoseconds
  // from getting it — a futex sleep costs ten thousand times that)
  class SpinLock {
   public:
    void lock() {
      for (;;) {
        if (!flag_.exchange(true, std::memory_order_acquire)) return;
        while (flag_.load(std::memory_order_relaxed)) {
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
        }
      }
    }
    bool try_lock() { return !flag_.exchange(true, std::memory_order_acquire); }
    void unlock() { flag_.store(false, std::memory_order_release); }
   private:
    std::atomic<bool> flag_{false};
  };
  SpinLock queue_lock_;
  std::deque<Pending*> queue_;
  std::atomic<std::uint32_t> queued_{0};  // entries of queue_ (read without the lock)

  RunningTaskBookkeeper running_task_bookkeeper_;
  HostStats host_stats_;  // guarded by allocation_lock_
  bool oplog_on_ = false;  // guarded by allocation_lock_, like the log itself
  std::string oplog_;
  void LogWait(const RequestView& r, std::chrono::nanoseconds lease, Clock::time_point now, int status,
               std::uint64_t id, const Servant* pick, std::uint32_t attempt = 1);

  std::thread timer_;
  std::mutex timer_lock_;
  std::condition_variable timer_cv_;
  bool stopping_ = false;
};

}  // namespace ydc
#endif  // YADCC_AMD_GPU_TASK_DISPATCHER_H_
