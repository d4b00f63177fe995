// Oracle shim (test infrastructure, NOT product code).
// flare::ReadCoarseSteadyClock() backed by a manually advanced fake clock so the
// reference's lease/expiry logic (task_dispatcher.cc:81,133-134,163,199,208-209,499)
// can be driven deterministically, without the reference tests' real sleeps.
#ifndef ORACLE_SHIM_FLARE_CLOCK_H_
#define ORACLE_SHIM_FLARE_CLOCK_H_
#include <chrono>
#include <cstdint>
namespace flare::shim {
inline std::int64_t& FakeNowNs() {
  static std::int64_t now = 1'000'000'000'000LL;  // Arbitrary non-zero epoch.
  return now;
}
}  // namespace flare::shim
namespace flare {
inline std::chrono::steady_clock::time_point ReadCoarseSteadyClock() {
#ifdef ORACLE_SHIM_REAL_THREADS
  // The multi-threaded build (oracle/ref_parked_bench.cc): waiters really sleep until a deadline.
  return std::chrono::steady_clock::now();
#else
  return std::chrono::steady_clock::time_point(
      std::chrono::nanoseconds(shim::FakeNowNs()));
#endif
}
inline std::chrono::steady_clock::time_point ReadSteadyClock() {
  return ReadCoarseSteadyClock();
}
}  // namespace flare
#endif
