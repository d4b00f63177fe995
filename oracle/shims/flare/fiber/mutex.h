// Oracle shim (test infrastructure, NOT product code). task_dispatcher.h:289.
#ifndef ORACLE_SHIM_FLARE_FIBER_MUTEX_H_
#define ORACLE_SHIM_FLARE_FIBER_MUTEX_H_
#include <mutex>
namespace flare::fiber {
using Mutex = std::mutex;
}
#endif
