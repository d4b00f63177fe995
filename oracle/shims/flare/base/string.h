// Oracle shim (test infrastructure, NOT product code).
// flare::StartsWith (task_dispatcher.cc:68) and flare::TryParse<size_t>
// (common/parse_size.cc:40, whole-string parse).
#ifndef ORACLE_SHIM_FLARE_STRING_H_
#define ORACLE_SHIM_FLARE_STRING_H_
#include <charconv>
#include <optional>
#include <string_view>
namespace flare {
inline bool StartsWith(std::string_view s, std::string_view prefix) {
  return s.size() >= prefix.size() && s.substr(0, prefix.size()) == prefix;
}
template <class T>
std::optional<T> TryParse(std::string_view s) {
  T value{};
  auto [end, ec] = std::from_chars(s.data(), s.data() + s.size(), value);
  if (ec != std::errc() || end != s.data() + s.size() || s.empty()) {
    return std::nullopt;
  }
  return value;
}
}  // namespace flare
#endif
