// Oracle shim (test infrastructure, NOT product code).
// A do-nothing Json::Value: the reference only uses it for the debug dump
// (task_dispatcher.cc:538-614), which the oracle never reads.
#ifndef ORACLE_SHIM_JSONCPP_VALUE_H_
#define ORACLE_SHIM_JSONCPP_VALUE_H_
#include <cstdint>
#include <string>
namespace Json {
using UInt64 = unsigned long long;
using Int64 = long long;
class Value {
 public:
  Value() = default;
  template <class T>
  Value(const T&) {}
  template <class T>
  Value& operator=(const T&) { return *this; }
  Value& operator[](int) { return *this; }
  Value& operator[](const char*) { return *this; }
  Value& operator[](const std::string&) { return *this; }
  template <class T>
  Value& append(const T&) { return *this; }
};
}  // namespace Json
#endif
