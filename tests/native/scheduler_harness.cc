// scheduler_harness.cc — see scheduler_harness.h.
#include "scheduler_harness.h"

#include <cctype>

using namespace std::literals;

namespace ydc {

namespace {

// "ip:port" (or "[v6]:port") -> (ip text, port text). Stands in for
// flare::TryParse<flare::Endpoint> (scheduler_service_impl.cc:88-93): anything without a
// numeric port is a misbehaving daemon.
bool SplitEndpoint(const std::string& s, std::string* ip, std::string* port) {
  auto colon = s.rfind(':');
  if (colon == std::string::npos || colon == 0 || colon + 1 == s.size()) return false;
  *ip = s.substr(0, colon);
  *port = s.substr(colon + 1);
  if (port->size() > 5) return false;
  for (char c : *port)
    if (!std::isdigit((unsigned char)c)) return false;
  if (ip->front() == '[' && ip->back() == ']') *ip = ip->substr(1, ip->size() - 2);
  return !ip->empty();
}

std::string FormatEndpoint(const std::string& ip, const std::string& port) {
  // IPv6 hosts are bracketed (:98-104).
  return ip.find(':') != std::string::npos ? "[" + ip + "]:" + port : ip + ":" + port;
}

}  // namespace

Status SchedulerHarness::Heartbeat(const std::string& peer_ip, const HeartbeatRequest& request,
                                   HeartbeatResponse* response) {
  if (!options_.is_user(request.token) && !options_.is_servant(request.token))
    return kStatusAccessDenied;  // :73-77
  if (request.version < options_.min_daemon_version) return kStatusVersionTooOld;  // :78-81
  std::string ip, port;
  if (!SplitEndpoint(request.location, &ip, &port)) return kStatusInvalidArgument;  // :88-93
  // The address observed by the scheduler is authoritative, the port is the servant's (:94-105).
  const std::string observed_location = FormatEndpoint(peer_ip, port);
  const std::string reported_location = FormatEndpoint(ip, port);
  const auto expires_in = request.next_heartbeat_in_ms * 1ms;
  if (expires_in > 30s) return kStatusInvalidArgument;  // :117-121

  ServantPersonality servant;
  servant.version = (int)request.version;
  servant.observed_location = observed_location;
  servant.reported_location = reported_location;
  servant.current_load = request.current_load;
  servant.num_processors = request.num_processors ? request.num_processors : request.capacity;  // :130-134
  servant.total_memory_in_bytes = request.total_memory_in_bytes;
  servant.memory_available_in_bytes = request.memory_available_in_bytes;
  servant.priority = request.servant_priority;
  if (servant.priority != kServantPriorityDedicated && servant.priority != kServantPriorityUser)
    servant.priority = kServantPriorityUser;  // older servants (:138-142)
  servant.max_tasks = request.capacity;
  servant.not_accepting_task_reason = (int)request.not_accepting_task_reason;
  if (observed_location != reported_location) {  // behind NAT: unreachable from outside (:146-153)
    servant.max_tasks = 0;
    servant.not_accepting_task_reason = 4;  // NOT_ACCEPTING_TASK_REASON_BEHIND_NAT
  }
  if (!options_.is_servant(request.token)) {  // :154-157
    servant.max_tasks = 0;
    servant.not_accepting_task_reason = 100;  // NOT_ACCEPTING_TASK_REASON_NOT_VERIFIED
  }
  servant.environments = request.env_descs;
  if (expires_in == 0ns) servant.max_tasks = 0;  // leaving: no further tasks (:168-170)
  dispatcher_->KeepServantAlive(servant, expires_in);
  // Reconciled under the REPORTED location, like the reference (:180-182).
  response->expired_tasks =
      dispatcher_->NotifyServantRunningTasks(request.location, request.running_tasks);
  return kStatusSuccess;
}

Status SchedulerHarness::WaitForStartingTask(const std::string& peer_ip,
                                             const WaitForStartingTaskRequest& request,
                                             WaitForStartingTaskResponse* response) {
  if (!options_.is_user(request.token)) return kStatusAccessDenied;
  const auto max_wait = request.milliseconds_to_wait * 1ms;
  const auto next_keep_alive = request.next_keep_alive_in_ms * 1ms;
  if (max_wait > 10s || next_keep_alive > 30s) return kStatusInvalidArgument;  // :221-226

  TaskPersonality task;
  task.requestor_ip = peer_ip;
  task.min_version = request.min_version;
  task.compiler_digest = request.compiler_digest;
  const auto now = dispatcher_->Now();
  auto grant = [&](const WaitResult& r) {
    response->grants.push_back({r->task_id, r->servant_location});
  };

  // Only the first grant of the RPC may wait (:236-240,257); all requests of one RPC share one
  // personality, so everything after the first grant is one batch with timeout == now.
  std::uint32_t immediate_left = request.immediate_reqs, prefetch_left = request.prefetch_reqs;
  if (immediate_left) {
    auto r = dispatcher_->WaitForStartingNewTask(task, next_keep_alive, now + max_wait, false);
    if (r.device_error) return kStatusNotImplemented;  // the GPU path is gone: fail the RPC
    if (!r && r.error() == WaitStatus::EnvironmentNotFound) return kStatusEnvironmentNotAvailable;
    if (r) {
      grant(r);
      --immediate_left;
    } else {
      immediate_left = 0;  // `break` (:247)
    }
  }
  if (response->grants.empty() && prefetch_left) {
    auto r = dispatcher_->WaitForStartingNewTask(task, next_keep_alive, now + max_wait, true);
    if (r.device_error) return kStatusNotImplemented;
    if (r) {
      grant(r);
      --prefetch_left;
    } else {
      prefetch_left = 0;
    }
  }
  if (!response->grants.empty() && immediate_left + prefetch_left) {
    std::vector<TaskPersonality> batch(immediate_left + prefetch_left, task);
    std::vector<bool> prefetching(batch.size(), false);
    for (std::size_t i = immediate_left; i < batch.size(); ++i) prefetching[i] = true;
    auto rs = dispatcher_->WaitForStartingNewTasks(batch, next_keep_alive, prefetching);
    for (auto&& r : rs)
      if (r.device_error) return kStatusNotImplemented;
    // Same personality throughout, so a request that finds nothing is followed by requests
    // that find nothing either: the first failure ends each of the two loops (:247,259).
    for (std::size_t i = 0; i < immediate_left && rs[i]; ++i) grant(rs[i]);
    for (std::size_t i = immediate_left; i < rs.size() && rs[i]; ++i) grant(rs[i]);
  }
  if (response->grants.empty()) return kStatusNoQuotaAvailable;  // :266-270
  return kStatusSuccess;
}

Status SchedulerHarness::KeepTaskAlive(const std::string& token,
                                       const std::vector<std::uint64_t>& task_grant_ids,
                                       std::uint32_t next_keep_alive_in_ms,
                                       std::vector<bool>* statuses) {
  if (!options_.is_user(token)) return kStatusAccessDenied;
  const auto next_keep_alive = next_keep_alive_in_ms * 1ms;
  if (next_keep_alive > 30s) return kStatusInvalidArgument;
  statuses->clear();
  for (auto id : task_grant_ids) statuses->push_back(dispatcher_->KeepTaskAlive(id, next_keep_alive));
  return kStatusSuccess;
}

Status SchedulerHarness::FreeTask(const std::string& token,
                                  const std::vector<std::uint64_t>& task_grant_ids) {
  if (!options_.is_user(token)) return kStatusAccessDenied;
  for (auto id : task_grant_ids) dispatcher_->FreeTask(id);
  return kStatusSuccess;
}

}  // namespace ydc
