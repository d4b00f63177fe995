// Oracle shim (test infrastructure, NOT product code).
// Plain-C++ stand-in for the protoc output of yadcc/api/scheduler.proto:
// enums :39-62 (same numeric values), RunningTask :233-238.
#ifndef ORACLE_SHIM_SCHEDULER_PB_H_
#define ORACLE_SHIM_SCHEDULER_PB_H_
#include <cstdint>
#include <string>

#include "yadcc/api/env_desc.pb.h"

namespace yadcc::scheduler {

enum ServantPriority : int {
  SERVANT_PRIORITY_UNKNOWN = 0,
  SERVANT_PRIORITY_DEDICATED = 1,
  SERVANT_PRIORITY_USER = 2,
};

enum NotAcceptingTaskReason : int {
  NOT_ACCEPTING_TASK_REASON_UNKNOWN = 0,
  NOT_ACCEPTING_TASK_REASON_USER_INSTRUCTED = 1,
  NOT_ACCEPTING_TASK_REASON_POOR_MACHINE = 2,
  NOT_ACCEPTING_TASK_REASON_CGROUPS_PRESENT = 3,
  NOT_ACCEPTING_TASK_REASON_BEHIND_NAT = 4,
  NOT_ACCEPTING_TASK_REASON_NOT_VERIFIED = 100,
};

// Value names of api/scheduler.proto:39-62 (what protoc's *_Name() return; an unknown
// number yields the empty string there too).
inline const std::string& ServantPriority_Name(ServantPriority v) {
  static const std::string names[] = {"SERVANT_PRIORITY_UNKNOWN", "SERVANT_PRIORITY_DEDICATED",
                                      "SERVANT_PRIORITY_USER"},
                           none;
  return (int)v >= 0 && (int)v <= 2 ? names[(int)v] : none;
}
inline const std::string& NotAcceptingTaskReason_Name(NotAcceptingTaskReason v) {
  static const std::string names[] = {"NOT_ACCEPTING_TASK_REASON_UNKNOWN",
                                      "NOT_ACCEPTING_TASK_REASON_USER_INSTRUCTED",
                                      "NOT_ACCEPTING_TASK_REASON_POOR_MACHINE",
                                      "NOT_ACCEPTING_TASK_REASON_CGROUPS_PRESENT",
                                      "NOT_ACCEPTING_TASK_REASON_BEHIND_NAT"},
                           not_verified = "NOT_ACCEPTING_TASK_REASON_NOT_VERIFIED", none;
  if ((int)v == 100) return not_verified;
  return (int)v >= 0 && (int)v <= 4 ? names[(int)v] : none;
}
inline bool ServantPriority_IsValid(int v) { return v >= 0 && v <= 2; }

class RunningTask {
 public:
  std::uint64_t servant_task_id() const { return servant_task_id_; }
  void set_servant_task_id(std::uint64_t v) { servant_task_id_ = v; }
  std::uint64_t task_grant_id() const { return task_grant_id_; }
  void set_task_grant_id(std::uint64_t v) { task_grant_id_ = v; }
  const std::string& servant_location() const { return servant_location_; }
  void set_servant_location(std::string v) { servant_location_ = std::move(v); }
  const std::string& task_digest() const { return task_digest_; }
  void set_task_digest(std::string v) { task_digest_ = std::move(v); }

 private:
  std::uint64_t servant_task_id_ = 0;
  std::uint64_t task_grant_id_ = 0;
  std::string servant_location_;
  std::string task_digest_;
};

}  // namespace yadcc::scheduler
#endif
