// td_api.cc — extern "C" wrapper (ydc_td_*, include/yadcc_dispatch.h) of
// GpuTaskDispatcher: one function per public method of the reference's
// TaskDispatcher (yadcc/scheduler/task_dispatcher.h:139-181).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <exception>
#include <string>
#include <string_view>
#include <vector>

#include "../../include/yadcc_dispatch.h"
#include "gpu_task_dispatcher.h"

using namespace std::literals;

struct ydc_td {
  std::atomic<std::int64_t> fake_now_ns{0};
  std::unique_ptr<ydc::GpuTaskDispatcher> impl;
  std::string dump, oplog;
};

namespace {

// false: `s` does not fit (the buffer gets the truncated text; the caller reports it).
bool CopyString(const std::string& s, char* out, size_t cap) {
  if (!out || !cap) return true;
  const size_t n = s.size() < cap ? s.size() : cap - 1;
  std::memcpy(out, s.data(), n);
  out[n] = 0;
  return s.size() < cap;
}

int StatusOf(const ydc::WaitResult& r) {
  if (r.device_error) return r.device_error;
  if (r.ok) return YDC_TD_GRANTED;
  return r.status == ydc::WaitStatus::EnvironmentNotFound ? YDC_TD_ENV_NOT_FOUND : YDC_TD_TIMEOUT;
}

}  // namespace

extern "C" {

int ydc_td_create(int device, const char* min_memory, int start_timer, int fake_clock,
                  ydc_td** out) {
  if (!out) return YDC_ERR_INVALID_ARGUMENT;
  auto* td = new ydc_td();
  ydc::GpuTaskDispatcher::Options opt;
  opt.device = device;
  if (min_memory) opt.servant_min_memory_for_accepting_new_task = min_memory;
  opt.start_expiration_timer = start_timer != 0;
  if (fake_clock) {
    opt.clock = [td] {
      return ydc::GpuTaskDispatcher::Clock::time_point(std::chrono::nanoseconds(td->fake_now_ns.load()));
    };
  }
  if (const char* e = std::getenv("YDC_TD_SPINS")) opt.caller_spins = std::max(1, std::atoi(e));  // (measurements)
  td->impl = std::make_unique<ydc::GpuTaskDispatcher>(opt);
  *out = td;
  return YDC_OK;
}

int ydc_td_destroy(ydc_td* td) {
  delete td;
  return YDC_OK;
}

int ydc_td_device_status(const ydc_td* td) {
  return td ? td->impl->device_status() : YDC_ERR_INVALID_ARGUMENT;
}

int ydc_td_set_clock_ns(ydc_td* td, int64_t now_ns) {
  if (!td) return YDC_ERR_INVALID_ARGUMENT;
  td->fake_now_ns.store(now_ns);
  return YDC_OK;
}

int ydc_td_keep_servant_alive(ydc_td* td, const ydc_td_servant* s, int64_t expires_in_ns) {
  if (!td || !s || !s->observed_location || (s->n_envs && !s->env_digests)) return YDC_ERR_INVALID_ARGUMENT;
  static thread_local std::vector<std::string_view> envs;
  envs.clear();
  for (size_t i = 0; i != s->n_envs; ++i) envs.emplace_back(s->env_digests[i]);
  ydc::ServantView p;
  p.version = s->version;
  p.observed_location = s->observed_location;
  p.reported_location = s->reported_location ? s->reported_location : s->observed_location;
  p.environments = envs.data();
  p.n_environments = envs.size();
  p.num_processors = s->num_processors;
  p.current_load = s->current_load;
  p.total_memory_in_bytes = s->total_memory_in_bytes;
  p.memory_available_in_bytes = s->memory_available_in_bytes;
  p.max_tasks = s->max_tasks;
  p.priority = s->priority;
  p.not_accepting_task_reason = s->not_accepting_task_reason;
  td->impl->KeepServantAlive(p, std::chrono::nanoseconds(expires_in_ns));
  return YDC_OK;
}

int ydc_td_wait_for_starting_new_task(ydc_td* td, const char* requestor_ip, uint32_t min_version,
                                      const char* compiler_digest, int64_t expires_in_ns,
                                      int64_t timeout_in_ns, int prefetching,
                                      uint64_t* out_task_id, char* out_location,
                                      size_t location_cap) {
  if (!td || !requestor_ip || !compiler_digest) return YDC_ERR_INVALID_ARGUMENT;
  ydc::TaskPersonality t;
  t.requestor_ip = requestor_ip;
  t.min_version = min_version;
  t.compiler_digest = compiler_digest;
  auto deadline = td->impl->Now() + std::chrono::nanoseconds(timeout_in_ns);
  auto r = td->impl->WaitForStartingNewTask(t, std::chrono::nanoseconds(expires_in_ns), deadline,
                                            prefetching != 0);
  if (r.ok) {
    if (out_task_id) *out_task_id = r.allocation.task_id;
    if (!CopyString(r.allocation.servant_location, out_location, location_cap)) {
      // A grant whose servant the caller cannot name is useless: give it back at once
      // instead of leaking the slot until the lease runs out.
      td->impl->FreeTask(r.allocation.task_id);
      return YDC_ERR_CAPACITY;
    }
  }
  return StatusOf(r);
}

int ydc_td_wait_for_starting_new_tasks(ydc_td* td, size_t n, const char* const* requestor_ips,
                                       const uint32_t* min_versions,
                                       const char* const* compiler_digests, int64_t expires_in_ns,
                                       const uint8_t* prefetching, int32_t* out_status,
                                       uint64_t* out_task_ids, char* out_locations,
                                       size_t location_stride) {
  if (!td || (n && (!requestor_ips || !min_versions || !compiler_digests || !out_status)))
    return YDC_ERR_INVALID_ARGUMENT;
  // Views of the caller's strings: nothing is copied on the way in, and the results go
  // straight into the caller's arrays (YDC_TD_* == the class's 0 / 1 / 2).
  static thread_local std::vector<ydc::RequestView> views;
  views.resize(n);
  // (one RPC's requests usually share the two strings, and a batch names a handful of digests:
  // the length of a string object seen a moment ago is not measured again)
  const char* last_ip = nullptr;
  size_t last_ip_len = 0;
  struct {
    const char* p = nullptr;
    size_t n = 0;
  } digests[4];
  unsigned next_digest = 0, cur = 0;
  for (size_t i = 0; i != n; ++i) {
    if (requestor_ips[i] != last_ip) {
      last_ip = requestor_ips[i];
      last_ip_len = std::strlen(last_ip);
    }
    const char* d = compiler_digests[i];
    if (digests[cur].p != d) {
      unsigned k = 0;
      while (k != 4 && digests[k].p != d) ++k;
      if (k == 4) {
        k = next_digest++ & 3;
        digests[k].p = d;
        digests[k].n = std::strlen(d);
      }
      cur = k;
    }
    views[i].requestor_ip = std::string_view(last_ip, last_ip_len);
    views[i].compiler_digest = std::string_view(d, digests[cur].n);
    views[i].min_version = min_versions[i];
    views[i].prefetching = prefetching && prefetching[i] != 0;
  }
  return td->impl->WaitForStartingNewTasksInto(n, views.data(), std::chrono::nanoseconds(expires_in_ns),
                                               out_status, out_task_ids, out_locations, location_stride);
}

int ydc_td_keep_task_alive(ydc_td* td, uint64_t task_id, int64_t new_expires_in_ns) {
  if (!td) return YDC_ERR_INVALID_ARGUMENT;
  return td->impl->KeepTaskAlive(task_id, std::chrono::nanoseconds(new_expires_in_ns)) ? 1 : 0;
}

int ydc_td_free_task(ydc_td* td, uint64_t task_id) {
  if (!td) return YDC_ERR_INVALID_ARGUMENT;
  td->impl->FreeTask(task_id);
  return YDC_OK;
}

int ydc_td_free_tasks(ydc_td* td, const uint64_t* task_ids, size_t n) {
  if (!td || (n && !task_ids)) return YDC_ERR_INVALID_ARGUMENT;
  td->impl->FreeTasks(task_ids, n);
  return YDC_OK;
}

int64_t ydc_td_notify_servant_running_tasks(ydc_td* td, const char* servant_location,
                                            const ydc_td_running_task* tasks, size_t n,
                                            uint64_t* out_unknown, size_t unknown_cap) {
  if (!td || !servant_location || (n && !tasks)) return YDC_ERR_INVALID_ARGUMENT;
  static thread_local std::vector<ydc::RunningTaskView> v;
  v.resize(n);
  const std::string_view where(servant_location);
  for (size_t i = 0; i != n; ++i) {
    v[i].servant_task_id = tasks[i].servant_task_id;
    v[i].task_grant_id = tasks[i].task_grant_id;
    v[i].servant_location = tasks[i].servant_location && tasks[i].servant_location != servant_location
                                ? std::string_view(tasks[i].servant_location)
                                : where;
    v[i].task_digest = tasks[i].task_digest ? std::string_view(tasks[i].task_digest) : std::string_view();
  }
  auto unknown = td->impl->NotifyServantRunningTasks(where, v.data(), n);
  for (size_t i = 0; i != unknown.size() && i < unknown_cap; ++i) out_unknown[i] = unknown[i];
  return (int64_t)unknown.size();
}

int64_t ydc_td_get_running_tasks(ydc_td* td, uint64_t* out_servant_task_ids,
                                 uint64_t* out_grant_ids, char* out_locations,
                                 size_t location_stride, char* out_digests, size_t digest_stride,
                                 size_t cap) {
  if (!td) return YDC_ERR_INVALID_ARGUMENT;
  const auto snapshot = td->impl->GetRunningTasksShared();  // (no copy of the list itself)
  const auto& tasks = *snapshot;
  for (size_t i = 0; i != tasks.size() && i < cap; ++i) {
    if (out_servant_task_ids) out_servant_task_ids[i] = tasks[i].servant_task_id;
    if (out_grant_ids) out_grant_ids[i] = tasks[i].task_grant_id;
    if (out_locations && location_stride)
      CopyString(tasks[i].servant_location, out_locations + i * location_stride, location_stride);
    if (out_digests && digest_stride)
      CopyString(tasks[i].task_digest, out_digests + i * digest_stride, digest_stride);
  }
  return (int64_t)tasks.size();
}

int ydc_td_running_tasks_acquire(ydc_td* td, void** out_handle, ydc_td_running_view* out_view) {
  if (!td || !out_handle || !out_view) return YDC_ERR_INVALID_ARGUMENT;
  ydc::RunningTaskBookkeeper::ColumnsSnapshot* held = nullptr;
  try {  // (the snapshot may have to be built: no exception crosses the C boundary)
    held = new ydc::RunningTaskBookkeeper::ColumnsSnapshot(td->impl->GetRunningTasksColumns());
  } catch (const std::exception&) {
    delete held;
    return YDC_ERR_CAPACITY;
  }
  const ydc::RunningTaskColumns& c = **held;
  out_view->n = c.task_grant_ids.size();
  out_view->servant_task_ids = c.servant_task_ids.data();
  out_view->task_grant_ids = c.task_grant_ids.data();
  out_view->location_off = c.location_off.data();
  out_view->location_len = c.location_len.data();
  out_view->digest_off = c.digest_off.data();
  out_view->digest_len = c.digest_len.data();
  out_view->strings = c.strings.c_str();
  *out_handle = held;
  return YDC_OK;
}

int ydc_td_running_tasks_release(void* handle) {
  delete (ydc::RunningTaskBookkeeper::ColumnsSnapshot*)handle;
  return YDC_OK;
}

int ydc_td_host_stats(ydc_td* td, ydc_td_stats* out) {
  if (!td || !out) return YDC_ERR_INVALID_ARGUMENT;
  const auto s = td->impl->host_stats();
  out->requests = s.requests;
  out->batches = s.batches;
  out->device_ns = s.device_ns;
  out->host_ns = s.host_ns;
  out->heartbeats = s.heartbeats;
  out->heartbeats_unchanged = s.heartbeats_unchanged;
  out->bookkeeper_rebuilds = s.bookkeeper_rebuilds;
  out->lease_pages = s.lease_pages;
  out->timer_ticks = s.timer_ticks;
  out->timer_lease_entries_seen = s.timer_lease_entries_seen;
  out->timer_last_ns = s.timer_last_ns;
  out->timer_max_ns = s.timer_max_ns;
  out->lease_wheel_entries = s.lease_wheel_entries;
  return YDC_OK;
}

int ydc_td_on_expiration_timer(ydc_td* td) {
  if (!td) return YDC_ERR_INVALID_ARGUMENT;
  td->impl->OnExpirationTimer();
  return YDC_OK;
}

const char* ydc_td_dump_internals(ydc_td* td) {
  if (!td) return "{}";
  td->dump = td->impl->DumpInternals();
  return td->dump.c_str();
}

int ydc_td_oplog_enable(ydc_td* td, int on) {
  if (!td) return YDC_ERR_INVALID_ARGUMENT;
  td->impl->EnableOpLog(on != 0);
  return YDC_OK;
}

const char* ydc_td_oplog_take(ydc_td* td) {
  if (!td) return "[]";
  td->oplog = td->impl->TakeOpLog();
  return td->oplog.c_str();
}

}  // extern "C"
