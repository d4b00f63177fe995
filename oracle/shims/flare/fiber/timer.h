// Oracle shim (test infrastructure, NOT product code).
// task_dispatcher.cc:81-82,90. The periodic callback is captured instead of
// scheduled; the driver fires it by hand (ref_fire_timers) to replay
// OnExpirationTimer deterministically.
#ifndef ORACLE_SHIM_FLARE_FIBER_TIMER_H_
#define ORACLE_SHIM_FLARE_FIBER_TIMER_H_
#include <chrono>
#include <cstdint>
#include <functional>
#include <map>

#include "flare/base/clock_shim.h"

namespace flare::shim {
inline std::map<std::uint64_t, std::function<void()>>& Timers() {
  static std::map<std::uint64_t, std::function<void()>> timers;
  return timers;
}
inline std::uint64_t& NextTimerId() {
  static std::uint64_t id = 1;
  return id;
}
}  // namespace flare::shim
namespace flare::fiber {
template <class F>
std::uint64_t SetTimer(std::chrono::steady_clock::time_point,
                       std::chrono::nanoseconds, F&& cb) {
  auto id = shim::NextTimerId()++;
  shim::Timers()[id] = std::forward<F>(cb);
  return id;
}
inline void KillTimer(std::uint64_t id) { shim::Timers().erase(id); }
}  // namespace flare::fiber
#endif
