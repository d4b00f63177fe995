// Oracle shim (test infrastructure, NOT product code). task_dispatcher.h:290.
// The harness is single threaded: nobody can ever signal a waiter, so a wait
// is reported as an immediate timeout (that is how "would block" is observed,
// task_dispatcher.cc:116-118) instead of sleeping on a clock the fake time
// never advances.
#ifndef ORACLE_SHIM_FLARE_FIBER_CV_H_
#define ORACLE_SHIM_FLARE_FIBER_CV_H_
#include <chrono>
#include <condition_variable>
#include <mutex>
#ifdef ORACLE_SHIM_REAL_THREADS
// The multi-threaded build (oracle/ref_parked_bench.cc, the parked-waiters baseline): fibers are
// OS threads there, fiber::Mutex is std::mutex, and this is the real thing.
namespace flare::fiber {
using ConditionVariable = std::condition_variable;
}
#else
namespace flare::fiber {
class ConditionVariable {
 public:
  template <class Lock, class TimePoint>
  std::cv_status wait_until(Lock&, const TimePoint&) {
    return std::cv_status::timeout;
  }
  void notify_all() {}
  void notify_one() {}
};
}  // namespace flare::fiber
#endif
#endif
