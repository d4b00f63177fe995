// Oracle shim (test infrastructure, NOT product code).
// task_dispatcher.h:299 / .cc:79-80 register a dump callback; keep it, never call it.
#ifndef ORACLE_SHIM_FLARE_EXPOSED_VAR_H_
#define ORACLE_SHIM_FLARE_EXPOSED_VAR_H_
#include <functional>
#include <string>
namespace flare {
template <class T>
class ExposedVarDynamic {
 public:
  ExposedVarDynamic(std::string path, std::function<T()> getter)
      : path_(std::move(path)), getter_(std::move(getter)) {}

 private:
  std::string path_;
  std::function<T()> getter_;
};
}  // namespace flare
#endif
