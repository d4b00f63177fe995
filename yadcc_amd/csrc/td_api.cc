// td_api.cc — extern "C" wrapper (ydc_td_*, include/yadcc_dispatch.h) of
// GpuTaskDispatcher: one function per public method of the reference's
// TaskDispatcher (yadcc/scheduler/task_dispatcher.h:139-181).
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/yadcc_dispatch.h"
#include "gpu_task_dispatcher.h"

using namespace std::literals;

struct ydc_td {
  std::atomic<std::int64_t> fake_now_ns{0};
  std::unique_ptr<ydc::GpuTaskDispatcher> impl;
  std::string dump;
};

namespace {

// false: `s` does not fit (the buffer gets the truncated text; the caller reports it).
bool CopyString(const std::string& s, char* out, size_t cap) {
  if (!out || !cap) return true;
  std::snprintf(out, cap, "%s", s.c_str());
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
  if (!td || !s || !s->observed_location) return YDC_ERR_INVALID_ARGUMENT;
  ydc::ServantPersonality p;
  p.version = s->version;
  p.observed_location = s->observed_location;
  p.reported_location = s->reported_location ? s->reported_location : s->observed_location;
  for (size_t i = 0; i != s->n_envs; ++i) p.environments.emplace_back(s->env_digests[i]);
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
  std::vector<ydc::TaskPersonality> ps(n);
  std::vector<bool> pf(n, false);
  for (size_t i = 0; i != n; ++i) {
    ps[i].requestor_ip = requestor_ips[i];
    ps[i].min_version = min_versions[i];
    ps[i].compiler_digest = compiler_digests[i];
    if (prefetching) pf[i] = prefetching[i] != 0;
  }
  auto rs = td->impl->WaitForStartingNewTasks(ps, std::chrono::nanoseconds(expires_in_ns), pf);
  int worst = YDC_OK;
  for (size_t i = 0; i != n; ++i) {
    out_status[i] = StatusOf(rs[i]);
    if (out_status[i] < 0) worst = out_status[i];
    if (out_task_ids) out_task_ids[i] = rs[i].ok ? rs[i].allocation.task_id : ~0ull;
    if (out_locations && location_stride &&
        !CopyString(rs[i].ok ? rs[i].allocation.servant_location : std::string(),
                    out_locations + i * location_stride, location_stride)) {
      td->impl->FreeTask(rs[i].allocation.task_id);  // (see the single-request wrapper)
      if (out_task_ids) out_task_ids[i] = ~0ull;
      out_status[i] = worst = YDC_ERR_CAPACITY;
    }
  }
  return worst;
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

int64_t ydc_td_notify_servant_running_tasks(ydc_td* td, const char* servant_location,
                                            const ydc_td_running_task* tasks, size_t n,
                                            uint64_t* out_unknown, size_t unknown_cap) {
  if (!td || !servant_location || (n && !tasks)) return YDC_ERR_INVALID_ARGUMENT;
  std::vector<ydc::RunningTask> v(n);
  for (size_t i = 0; i != n; ++i) {
    v[i].servant_task_id = tasks[i].servant_task_id;
    v[i].task_grant_id = tasks[i].task_grant_id;
    v[i].servant_location = tasks[i].servant_location ? tasks[i].servant_location : servant_location;
    if (tasks[i].task_digest) v[i].task_digest = tasks[i].task_digest;
  }
  auto unknown = td->impl->NotifyServantRunningTasks(servant_location, std::move(v));
  for (size_t i = 0; i != unknown.size() && i < unknown_cap; ++i) out_unknown[i] = unknown[i];
  return (int64_t)unknown.size();
}

int64_t ydc_td_get_running_tasks(ydc_td* td, uint64_t* out_servant_task_ids,
                                 uint64_t* out_grant_ids, char* out_locations,
                                 size_t location_stride, char* out_digests, size_t digest_stride,
                                 size_t cap) {
  if (!td) return YDC_ERR_INVALID_ARGUMENT;
  auto tasks = td->impl->GetRunningTasks();
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

}  // extern "C"
