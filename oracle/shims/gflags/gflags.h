// Oracle shim (test infrastructure, NOT product code).
// Only DEFINE_string is needed (task_dispatcher.cc:35-38).
#ifndef ORACLE_SHIM_GFLAGS_H_
#define ORACLE_SHIM_GFLAGS_H_
#include <string>
#define DEFINE_string(name, value, help) std::string FLAGS_##name = (value)
#define DECLARE_string(name) extern std::string FLAGS_##name
#endif
