// Oracle shim (test infrastructure, NOT product code).
// task_dispatcher.cc:46-53,558-559,596-597 (FormatTime for the JSON dump).
#ifndef ORACLE_SHIM_FLARE_TIME_VIEW_H_
#define ORACLE_SHIM_FLARE_TIME_VIEW_H_
#include <chrono>
namespace flare::internal {
class SystemClockView {
 public:
  SystemClockView(std::chrono::system_clock::time_point tp) : tp_(tp) {}
  SystemClockView(std::chrono::steady_clock::time_point tp)
      : tp_(std::chrono::system_clock::time_point(
            std::chrono::duration_cast<std::chrono::system_clock::duration>(
                tp.time_since_epoch()))) {}
  std::chrono::system_clock::time_point Get() const { return tp_; }

 private:
  std::chrono::system_clock::time_point tp_;
};
}  // namespace flare::internal
#endif
