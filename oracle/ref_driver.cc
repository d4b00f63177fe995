// Oracle (test infrastructure, NOT product code). See ref_driver.h.
#include "ref_driver.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "flare/base/clock_shim.h"
#include "flare/fiber/timer.h"
#include "yadcc/scheduler/task_dispatcher.h"  // the reference's own header

using namespace std::literals;
using yadcc::scheduler::RunningTask;
using yadcc::scheduler::ServantPersonality;
using yadcc::scheduler::TaskDispatcher;
using yadcc::scheduler::TaskPersonality;
using yadcc::scheduler::WaitStatus;

// DumpInternals is private in the reference; its header befriends the gtest fixture class
// SchedulerServiceImpl_TokenWithIntersection_Test (task_dispatcher.h:277 through FRIEND_TEST).
// The driver lends that name to reach the dump without touching the reference's sources.
namespace yadcc::scheduler {
class SchedulerServiceImpl_TokenWithIntersection_Test {
 public:
  static std::string Dump(TaskDispatcher* d) { return d->DumpInternals().Dump(); }
  // The same door reaches the registry: presets `running_tasks` of the servant registered last
  // (what millions of priming grants would leave there, without the minutes they take). Only
  // placement reads it afterwards (GetCapacityAvailable, the pick); no TaskDesc stands behind
  // the count, so such a registry is for dispatch-only cases.
  static void PresetRunningOfLast(TaskDispatcher* d, std::size_t running) {
    d->servants_.servants.back()->running_tasks = running;
  }
};
}  // namespace yadcc::scheduler

struct ref_dispatcher {
  TaskDispatcher impl;
  // location -> registry index, valid while no servant has expired.
  std::unordered_map<std::string, std::uint32_t> index_of;
  std::vector<std::string> digest_names;
  std::uint32_t env_bits = 64;  // digests the loaded masks can name (64 * env_words)
};

namespace {

constexpr std::uint32_t kPresetRunningFrom = 1u << 20;

std::string Dotted(std::uint32_t ip, std::uint32_t port, bool with_port) {
  char buf[64];
  if (with_port) {
    std::snprintf(buf, sizeof(buf), "%u.%u.%u.%u:%u", ip >> 24, (ip >> 16) & 255,
                  (ip >> 8) & 255, ip & 255, port);
  } else {
    std::snprintf(buf, sizeof(buf), "%u.%u.%u.%u", ip >> 24, (ip >> 16) & 255,
                  (ip >> 8) & 255, ip & 255);
  }
  return buf;
}

std::string DigestName(std::uint32_t env_id) {
  // 64 lower-case hex chars, like a BLAKE3 digest (api/env_desc.proto:27-28).
  // All digests share a long common prefix so that string compares are not
  // decided by the first byte (the realistic worst case for operator==).
  char buf[65];
  std::uint64_t x = 0x9E3779B97F4A7C15ull * (env_id + 1);
  std::snprintf(buf, sizeof(buf), "c0ffee00c0ffee00c0ffee00c0ffee00c0ffee00c0ffee00%016llx",
                static_cast<unsigned long long>(x));
  return buf;
}

const std::string& CachedDigest(ref_dispatcher* d, std::uint32_t env_id) {
  if (d->digest_names.size() <= env_id) {
    auto old = d->digest_names.size();
    d->digest_names.resize(env_id + 1);
    for (auto i = old; i <= env_id; ++i) d->digest_names[i] = DigestName(i);
  }
  return d->digest_names[env_id];
}

}  // namespace

extern "C" {

ref_dispatcher* ref_create(void) { return new ref_dispatcher(); }
void ref_destroy(ref_dispatcher* d) { delete d; }

void ref_clock_advance_ms(int64_t ms) { flare::shim::FakeNowNs() += ms * 1'000'000; }
int64_t ref_clock_now_ns(void) { return flare::shim::FakeNowNs(); }
void ref_fire_timers(void) {
  auto copy = flare::shim::Timers();
  for (auto&& [id, cb] : copy) cb();
}

void ref_keep_servant_alive(ref_dispatcher* d, int version, const char* observed_location,
                            const char* reported_location, const char* const* env_digests,
                            size_t n_envs, uint64_t num_processors, uint64_t current_load,
                            uint64_t total_memory, uint64_t memory_available,
                            uint64_t max_tasks, int priority, int not_accepting_reason,
                            int64_t expires_in_ms) {
  ServantPersonality s{};
  s.version = version;
  s.observed_location = observed_location;
  s.reported_location = reported_location;
  for (size_t i = 0; i != n_envs; ++i) {
    s.environments.emplace_back().set_compiler_digest(env_digests[i]);
  }
  s.num_processors = num_processors;
  s.current_load = current_load;
  s.total_memory_in_bytes = total_memory;
  s.memory_available_in_bytes = memory_available;
  s.max_tasks = max_tasks;
  s.priority = static_cast<yadcc::scheduler::ServantPriority>(priority);
  s.not_accepting_task_reason =
      static_cast<yadcc::scheduler::NotAcceptingTaskReason>(not_accepting_reason);
  d->impl.KeepServantAlive(s, expires_in_ms * 1ms);
  d->index_of.emplace(s.observed_location, static_cast<std::uint32_t>(d->index_of.size()));
}

int ref_wait_for_starting_new_task(ref_dispatcher* d, const char* requestor_ip,
                                   uint32_t min_version, const char* compiler_digest,
                                   int64_t expires_in_ms, int64_t timeout_in_ms, int prefetching,
                                   uint64_t* out_task_id, char* out_location,
                                   size_t location_cap) {
  TaskPersonality t;
  t.requestor_ip = requestor_ip;
  t.min_version = min_version;
  t.env_desc.set_compiler_digest(compiler_digest);
  auto r = d->impl.WaitForStartingNewTask(t, expires_in_ms * 1ms,
                                          flare::ReadCoarseSteadyClock() + timeout_in_ms * 1ms,
                                          prefetching != 0);
  if (!r) {
    return r.error() == WaitStatus::EnvironmentNotFound ? REF_ENV_NOT_FOUND : REF_TIMEOUT;
  }
  if (out_task_id) *out_task_id = r->task_id;
  if (out_location && location_cap) {
    std::snprintf(out_location, location_cap, "%s", r->servant_location.c_str());
  }
  return REF_OK;
}

int ref_keep_task_alive(ref_dispatcher* d, uint64_t task_id, int64_t new_expires_in_ms) {
  return d->impl.KeepTaskAlive(task_id, new_expires_in_ms * 1ms) ? 1 : 0;
}

void ref_free_task(ref_dispatcher* d, uint64_t task_id) { d->impl.FreeTask(task_id); }

size_t ref_notify_servant_running_tasks(ref_dispatcher* d, const char* servant_location,
                                        const uint64_t* servant_task_ids,
                                        const uint64_t* grant_ids, size_t n,
                                        uint64_t* out_unknown, size_t unknown_cap) {
  std::vector<RunningTask> tasks(n);
  for (size_t i = 0; i != n; ++i) {
    tasks[i].set_servant_task_id(servant_task_ids ? servant_task_ids[i] : i);
    tasks[i].set_task_grant_id(grant_ids[i]);
    tasks[i].set_servant_location(servant_location);
  }
  auto unknown = d->impl.NotifyServantRunningTasks(servant_location, std::move(tasks));
  for (size_t i = 0; i != unknown.size() && i < unknown_cap; ++i) out_unknown[i] = unknown[i];
  return unknown.size();
}

size_t ref_get_running_tasks(ref_dispatcher* d, uint64_t* out_servant_task_ids,
                             uint64_t* out_grant_ids, size_t cap) {
  auto tasks = d->impl.GetRunningTasks();
  for (size_t i = 0; i != tasks.size() && i < cap; ++i) {
    if (out_servant_task_ids) out_servant_task_ids[i] = tasks[i].servant_task_id();
    if (out_grant_ids) out_grant_ids[i] = tasks[i].task_grant_id();
  }
  return tasks.size();
}

void ref_digest_name(uint32_t env_id, char* buf) {
  std::snprintf(buf, 65, "%s", DigestName(env_id).c_str());
}

void ref_load_servants(ref_dispatcher* d, size_t n, const uint32_t* version,
                       const uint32_t* num_processors, const uint32_t* current_load,
                       const uint32_t* max_tasks, const uint32_t* running_tasks,
                       const uint32_t* priority, const uint64_t* total_memory,
                       const uint64_t* memory_available, const uint64_t* env_mask,
                       const uint32_t* ip, const uint32_t* port) {
  ref_load_servants_wide(d, n, version, num_processors, current_load, max_tasks, running_tasks,
                         priority, total_memory, memory_available, env_mask, 1, ip, port);
}

void ref_load_servants_wide(ref_dispatcher* d, size_t n, const uint32_t* version,
                            const uint32_t* num_processors, const uint32_t* current_load,
                            const uint32_t* max_tasks, const uint32_t* running_tasks,
                            const uint32_t* priority, const uint64_t* total_memory,
                            const uint64_t* memory_available, const uint64_t* env_mask,
                            uint32_t env_words, const uint32_t* ip, const uint32_t* port) {
  d->env_bits = 64 * (env_words ? env_words : 1);
  for (size_t i = 0; i != n; ++i) {
    auto location = Dotted(ip[i], port[i], true);
    bool is_new = d->index_of.count(location) == 0;
    std::uint32_t prime = (running_tasks && is_new) ? running_tasks[i] : 0;
    // Millions of priming grants take minutes: such counts are preset through the friend door.
    const bool preset = prime >= kPresetRunningFrom;
    if (prime && !preset) {
      // Install a wide-open personality under a digest only this servant has,
      // grant `prime` tasks to it, then install the real personality.
      ServantPersonality s{};
      s.version = 0x7fffffff;
      s.observed_location = s.reported_location = location;
      auto private_digest = "prime:" + location;
      s.environments.emplace_back().set_compiler_digest(private_digest);
      s.num_processors = s.max_tasks = static_cast<std::size_t>(prime) + 1;
      s.current_load = 0;
      s.priority = yadcc::scheduler::SERVANT_PRIORITY_USER;
      d->impl.KeepServantAlive(s, 30s);
      TaskPersonality t;
      t.requestor_ip = "0.0.0.0";
      t.min_version = 0;
      t.env_desc.set_compiler_digest(private_digest);
      for (std::uint32_t k = 0; k != prime; ++k) {
        auto r = d->impl.WaitForStartingNewTask(t, 3600s, flare::ReadCoarseSteadyClock(), false);
        if (!r) std::abort();
      }
    }
    ServantPersonality s{};
    s.version = static_cast<int>(version[i]);
    s.observed_location = s.reported_location = location;
    for (std::uint32_t j = 0; j != d->env_bits; ++j) {
      if (env_mask[i * (d->env_bits / 64) + j / 64] >> (j % 64) & 1) {
        s.environments.emplace_back().set_compiler_digest(CachedDigest(d, j));
      }
    }
    s.num_processors = num_processors[i];
    s.current_load = current_load[i];
    s.total_memory_in_bytes = total_memory[i];
    s.memory_available_in_bytes = memory_available[i];
    s.max_tasks = max_tasks[i];
    s.priority = static_cast<yadcc::scheduler::ServantPriority>(priority[i]);
    s.not_accepting_task_reason = yadcc::scheduler::NOT_ACCEPTING_TASK_REASON_UNKNOWN;
    d->impl.KeepServantAlive(s, 30s);
    if (preset)
      yadcc::scheduler::SchedulerServiceImpl_TokenWithIntersection_Test::PresetRunningOfLast(&d->impl, prime);
    d->index_of.emplace(location, static_cast<std::uint32_t>(d->index_of.size()));
  }
}

size_t ref_dump_internals(ref_dispatcher* d, char* out, size_t cap) {
  const std::string j = yadcc::scheduler::SchedulerServiceImpl_TokenWithIntersection_Test::Dump(&d->impl);
  if (out && cap) std::snprintf(out, cap, "%s", j.c_str());
  return j.size();
}

void ref_free_tasks(ref_dispatcher* d, const uint64_t* task_ids, size_t n) {
  for (size_t i = 0; i != n; ++i) d->impl.FreeTask(task_ids[i]);
}

double ref_dispatch_batch(ref_dispatcher* d, size_t n, const uint32_t* env_id,
                          const uint32_t* min_version, const uint32_t* requestor_ip,
                          uint32_t* out_servant_idx, uint64_t* out_task_id,
                          uint64_t* out_latency_ns) {
  // Task descriptors are built outside the timed region: the RPC layer hands
  // TaskDispatcher a ready TaskPersonality (scheduler_service_impl.cc:228-231).
  std::vector<TaskPersonality> tasks(n);
  for (size_t i = 0; i != n; ++i) {
    tasks[i].requestor_ip = Dotted(requestor_ip[i], 0, false);
    tasks[i].min_version = min_version[i];
    tasks[i].env_desc.set_compiler_digest(env_id[i] < d->env_bits ? CachedDigest(d, env_id[i])
                                                                  : std::string("unknown-digest"));
  }
  std::vector<const std::string*> granted(n, nullptr);
  std::vector<int> status(n, 0);
  auto now = flare::ReadCoarseSteadyClock();
  auto t0 = std::chrono::steady_clock::now();
  for (size_t i = 0; i != n; ++i) {
    std::chrono::steady_clock::time_point c0;
    if (out_latency_ns) c0 = std::chrono::steady_clock::now();
    auto r = d->impl.WaitForStartingNewTask(tasks[i], 15s, now, false);
    if (r) {
      if (out_task_id) out_task_id[i] = r->task_id;
      auto it = d->index_of.find(r->servant_location);
      out_servant_idx[i] = it == d->index_of.end() ? 0xFFFFFFFDu : it->second;
    } else {
      if (out_task_id) out_task_id[i] = ~0ull;
      out_servant_idx[i] = r.error() == WaitStatus::EnvironmentNotFound ? REF_IDX_ENV_NOT_FOUND
                                                                        : REF_IDX_TIMEOUT;
    }
    if (out_latency_ns) {
      out_latency_ns[i] = std::chrono::duration_cast<std::chrono::nanoseconds>(
                              std::chrono::steady_clock::now() - c0)
                              .count();
    }
  }
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
