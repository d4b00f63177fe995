// Oracle shim (test infrastructure, NOT product code).
// CHECKs abort (the reference's invariants must hold in the oracle too);
// log statements compile away but still evaluate nothing.
#ifndef ORACLE_SHIM_FLARE_LOGGING_H_
#define ORACLE_SHIM_FLARE_LOGGING_H_
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "flare/base/clock_shim.h"

namespace flare::shim {
[[noreturn]] inline void CheckFailed(const char* expr, const char* file, int line) {
  std::fprintf(stderr, "FLARE_CHECK failed: %s at %s:%d\n", expr, file, line);
  std::abort();
}
template <class... Ts>
inline void Sink(const Ts&...) {}
}  // namespace flare::shim

#define FLARE_CHECK(c, ...) \
  ((c) ? (void)0 : ::flare::shim::CheckFailed(#c, __FILE__, __LINE__))
#define FLARE_CHECK_EQ(a, b, ...) FLARE_CHECK((a) == (b))
#define FLARE_CHECK_NE(a, b, ...) FLARE_CHECK((a) != (b))
#define FLARE_CHECK_GT(a, b, ...) FLARE_CHECK((a) > (b))
#define FLARE_CHECK_GE(a, b, ...) FLARE_CHECK((a) >= (b))
#define FLARE_CHECK_LT(a, b, ...) FLARE_CHECK((a) < (b))
#define FLARE_CHECK_LE(a, b, ...) FLARE_CHECK((a) <= (b))

#define FLARE_LOG_INFO(...) ((void)0)
#define FLARE_LOG_WARNING(...) ((void)0)
#define FLARE_LOG_ERROR(...) ((void)0)
#define FLARE_LOG_WARNING_EVERY_SECOND(...) ((void)0)
#define FLARE_LOG_ERROR_EVERY_SECOND(...) ((void)0)
#define FLARE_LOG_WARNING_IF(c, ...) ((void)(c))
#define FLARE_LOG_ERROR_IF(c, ...) ((void)(c))
#define FLARE_LOG_WARNING_IF_EVERY_SECOND(c, ...) ((void)(c))
#define FLARE_LOG_ERROR_IF_EVERY_SECOND(c, ...) ((void)(c))
#define FLARE_VLOG(n, ...) ((void)0)
#endif
