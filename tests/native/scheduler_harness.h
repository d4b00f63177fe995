// scheduler_harness.h — TEST SCAFFOLDING (not in libydc.so): flare-free stand-in for the
// reference's RPC layer.
//
// SchedulerServiceImpl (reference yadcc/scheduler/scheduler_service_impl.cc) is the only
// caller of TaskDispatcher. It is NOT rebuilt here (flare RPC, protobuf and token roll-out
// are out of scope); this class restates its *call pattern* into the six dispatcher methods
// — argument sanity, heartbeat -> ServantPersonality translation (:124-171), the grant loops
// (:228-264) and the status mapping — on plain structs shaped like api/scheduler.proto, so
// that the MI355X dispatcher can be driven natively (C++) exactly the way the scheduler
// binary drives the reference class. The peer address, which the reference takes from the RPC
// controller, is an explicit argument.
#ifndef YADCC_AMD_SCHEDULER_HARNESS_H_
#define YADCC_AMD_SCHEDULER_HARNESS_H_

#include <chrono>
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "gpu_task_dispatcher.h"

namespace ydc {

// api/scheduler.proto:22-35.
enum Status {
  kStatusSuccess = 0,
  kStatusNoQuotaAvailable = 1001,
  kStatusNotImplemented = 1002,
  kStatusAccessDenied = 1003,
  kStatusInvalidArgument = 1004,
  kStatusVersionTooOld = 1005,
  kStatusEnvironmentNotAvailable = 1006,
};

// api/scheduler.proto:64-120.
struct HeartbeatRequest {
  std::string token;
  std::uint32_t next_heartbeat_in_ms = 0;
  std::uint32_t version = 0;
  std::string location;  // "ip:port" as the servant sees itself
  std::uint32_t num_processors = 0, current_load = 0;
  int servant_priority = kServantPriorityUnknown;
  std::uint32_t not_accepting_task_reason = 0;
  std::uint32_t capacity = 0;
  std::uint64_t total_memory_in_bytes = 0, memory_available_in_bytes = 0;
  std::vector<std::string> env_descs;  // compiler digests
  std::vector<RunningTask> running_tasks;
};
struct HeartbeatResponse {
  std::vector<std::uint64_t> expired_tasks;
};

// api/scheduler.proto:181-210.
struct WaitForStartingTaskRequest {
  std::string token;
  std::uint32_t milliseconds_to_wait = 0;
  std::string compiler_digest;  // env_desc
  std::uint32_t immediate_reqs = 0, prefetch_reqs = 0;
  std::uint32_t next_keep_alive_in_ms = 0;
  std::uint32_t min_version = 0;
};
struct StartingTaskGrant {
  std::uint64_t task_grant_id = 0;
  std::string servant_location;
};
struct WaitForStartingTaskResponse {
  std::vector<StartingTaskGrant> grants;
};

class SchedulerHarness {
 public:
  struct Options {
    std::uint32_t min_daemon_version = 0;  // --min_daemon_version
    // Token classes (common/token_verifier.cc); default: everybody is a verified servant.
    std::function<bool(const std::string&)> is_user = [](const std::string&) { return true; };
    std::function<bool(const std::string&)> is_servant = [](const std::string&) { return true; };
  };
  SchedulerHarness(GpuTaskDispatcher* dispatcher, const Options& options)
      : dispatcher_(dispatcher), options_(options) {}
  explicit SchedulerHarness(GpuTaskDispatcher* dispatcher) : SchedulerHarness(dispatcher, Options()) {}

  // scheduler_service_impl.cc:67-194. peer_ip: what the RPC controller reports.
  Status Heartbeat(const std::string& peer_ip, const HeartbeatRequest& request,
                   HeartbeatResponse* response);
  // :209-271. Grants after the first one are placed as ONE device batch.
  Status WaitForStartingTask(const std::string& peer_ip, const WaitForStartingTaskRequest& request,
                             WaitForStartingTaskResponse* response);
  // :273-294.
  Status KeepTaskAlive(const std::string& token, const std::vector<std::uint64_t>& task_grant_ids,
                       std::uint32_t next_keep_alive_in_ms, std::vector<bool>* statuses);
  // :296-310.
  Status FreeTask(const std::string& token, const std::vector<std::uint64_t>& task_grant_ids);
  // :312-318.
  std::vector<RunningTask> GetRunningTasks() { return dispatcher_->GetRunningTasks(); }

 private:
  GpuTaskDispatcher* dispatcher_;
  Options options_;
};

}  // namespace ydc
#endif  // YADCC_AMD_SCHEDULER_HARNESS_H_
