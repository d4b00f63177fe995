// Oracle shim (test infrastructure, NOT product code).
// Minimal flare::Expected<T, E> used by task_dispatcher.h:139 / .cc:107,117,137.
#ifndef ORACLE_SHIM_FLARE_EXPECTED_H_
#define ORACLE_SHIM_FLARE_EXPECTED_H_
#include <utility>
#include <variant>
namespace flare {
template <class T, class E>
class Expected {
 public:
  Expected(T value) : v_(std::in_place_index<0>, std::move(value)) {}
  Expected(E error) : v_(std::in_place_index<1>, std::move(error)) {}
  explicit operator bool() const { return v_.index() == 0; }
  T* operator->() { return &std::get<0>(v_); }
  const T* operator->() const { return &std::get<0>(v_); }
  T& operator*() { return std::get<0>(v_); }
  const T& operator*() const { return std::get<0>(v_); }
  const E& error() const { return std::get<1>(v_); }

 private:
  std::variant<T, E> v_;
};
}  // namespace flare
#endif
